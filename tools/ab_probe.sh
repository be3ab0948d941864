# A/B of two builds of the library on the same box: tools/diag/liblcrec_hip_prev.so against the in-tree build
for lib in tools/diag/liblcrec_hip_prev.so lc-rec_amd/csrc/liblcrec_hip.so tools/diag/liblcrec_hip_prev.so lc-rec_amd/csrc/liblcrec_hip.so; do
  echo "== $lib"
  LCREC_LIB_PATH=$lib timeout -k 10 100 python tools/gemm_probe.py --rows 131072 --layers 0 1 2 3 4 | grep layer
done
