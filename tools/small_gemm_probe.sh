# generic (non ping-pong) GEMM kernels at training / Games sizes, with and without the VALU-free staging
for fast in 0 1; do
  for rows in 2048 16896; do
    echo "== rows $rows FAST=$fast"
    LCREC_GEMM_PP=0 LCREC_GEMM_FAST=$fast timeout -k 10 100 python tools/gemm_probe.py --rows $rows --layers 0 1 2 3 4 5 6 --reps 30 | grep layer
  done
done
