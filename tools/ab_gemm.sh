# forward/dX/dW time of one batch-sized layer under several builds of the library: LIBS="a.so b.so" ROWS=1024 bash tools/ab_gemm.sh
for lib in $LIBS; do
  echo "== $lib"
  LCREC_LIB_PATH=$lib timeout -k 10 100 python tools/train_gemm_probe.py --rows ${ROWS:-1024} 2>&1 | grep -a "2048 ->  1024\|768 ->  2048\|1024 ->   512"
done
