for pp3 in 0 1 0 1; do
  echo "== PP3=$pp3"
  LCREC_GEMM_PP3=$pp3 timeout -k 10 100 python tools/gemm_probe.py --rows 131072 --layers 0 1 2 3 | grep layer
done
