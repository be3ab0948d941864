# A/B of builds of the library on the same box at training-step shapes (default: tools/diag/liblcrec_hip_prev.so against the
# in-tree build; LIBS="a.so b.so ..." for others)
for lib in ${LIBS:-tools/diag/liblcrec_hip_prev.so lc-rec_amd/csrc/liblcrec_hip.so}; do
  echo "== $lib"
  LCREC_LIB_PATH=$lib timeout -k 10 150 python tools/train_gemm_probe.py --rows ${ROWS:-1024} 2>&1 | grep -a "TF\|total\|batch" | tail -12
  LCREC_LIB_PATH=$lib timeout -k 10 150 python tools/train_probe.py --trainer --steps 96 --batch 1024 --bn 2>&1 | tr "\r" "\n" | grep -a "Trainer._train_epoch\|Error\|error" | tail -2
done
