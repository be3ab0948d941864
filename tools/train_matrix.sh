#!/bin/bash
# ms/step of Trainer._train_epoch for the run.sh recipe (4 x 256 codes, Sinkhorn on the last level, 768-d) at batch 1024 / 2048,
# with and without BatchNorm, engine (one hipGraph per step) vs the autograd path.
for B in 1024 2048; do for BN in "" "--bn"; do for ENG in auto off; do
  python tools/train_probe.py --trainer --steps 96 --batch $B $BN --engine $ENG 2>&1 | tr "\r" "\n" | grep -a "Trainer._train_epoch\|Error\|error" | tail -3
done; done; done
