#!/bin/bash
# forward GEMM of every layer at batch 1024 / 2048 with 64x64, 64x128 and 128x64 workgroup tiles (LCREC_GEMM_SMALL=0/1/2)
for v in 0 1 2; do echo "== LCREC_GEMM_SMALL=$v"; LCREC_GEMM_SMALL=$v python tools/train_gemm_probe.py 2>&1 | grep -a "fwd" | sed 's/| dX.*//'; done
