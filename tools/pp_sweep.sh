for rows in 2048 8192 16896 32768 65536; do
  for pp in 0 1; do
    echo "== rows $rows PP=$pp"
    LCREC_GEMM_PP=$pp timeout -k 10 100 python tools/gemm_probe.py --rows $rows --layers 0 1 2 3 --reps 30 | grep layer
  done
done
