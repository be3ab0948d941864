#!/bin/bash
# a second build of the library with extra compiler flags, for A/B runs on one box:
#   tools/build_variant.sh ring6 -DLCREC_GEMM_RING=6     -> tools/diag/liblcrec_hip_ring6.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/tools/diag/obj_$name
mkdir -p $obj
cd $root/lc-rec_amd/csrc
for f in abi gemm_f32 rq_assign vq_train train_ops collide index_json; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-function "$@" -c $f.hip -o $obj/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tools/diag/liblcrec_hip_$name.so $obj/*.o
rm -rf $obj
ls -la $root/tools/diag/liblcrec_hip_$name.so
