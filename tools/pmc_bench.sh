#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters, one counter per pass (MI355X_MICROARCH.md: separate --pmc passes),
# kernel trace only.  Writes gpurun_out/pmc_r03/{fetch,write}/..._results.db and folds them into profiles/pmc_dominant_kernel.json
# via tools/rocpd_summary.py pmc.   tools/pmc_bench.sh [workload]
set -u
wl=${1:-c3}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$root/gpurun_out/pmc_r03/$wl/$c
  mkdir -p $d
  timeout -k 5 240 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python3 $root/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --pipelines 1 > $d/run.log 2>&1
  echo "$c exit $?" | tee -a $d/run.log
done
cd $root
python tools/rocpd_summary.py pmc --fetch gpurun_out/pmc_r03/$wl/FETCH_SIZE/pmc_results.db --write gpurun_out/pmc_r03/$wl/WRITE_SIZE/pmc_results.db --workload $wl --out gpurun_out/pmc_r03/pmc_dominant_kernel.json
