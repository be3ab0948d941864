#!/bin/bash
# diagnostic build of the library (in-kernel cycle stamps) -> tools/diag/liblcrec_hip_stamp.so
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/tools/diag/obj
cd $root/lc-rec_amd/csrc
for f in abi gemm_f32 rq_assign vq_train train_ops collide index_json; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-function -DLCREC_GEMM_STAMP -c $f.hip -o $root/tools/diag/obj/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tools/diag/liblcrec_hip_stamp.so $root/tools/diag/obj/*.o
ls -la $root/tools/diag/liblcrec_hip_stamp.so
