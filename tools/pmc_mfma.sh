#!/bin/bash
# MFMA-pipe utilisation of the path's three MFMA kernels from SQ counters (BASELINE north_star: "MFMA utilisation on the
# distance GEMM against gfx950 peak").  One small group of counters per pass, kernel trace only, the program itself after `--`.
#   tools/pmc_mfma.sh            -> gpurun_out/pmc_mfma/<pass>/pmc_results.db -> gpurun_out/pmc_mfma/pmc_mfma.txt (copy to profiles/)
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_mfma
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -a -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $out/mfma_counters_available.txt
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVES"; do
  i=$((i+1)); d=$out/pass$i; mkdir -p $d
  timeout -k 5 200 rocprofv3 --pmc $grp --kernel-trace -d $d -o pmc -- python3 $root/tools/pmc_target.py > $d/run.log 2>&1
  echo "pass $i ($grp) exit $?" | tee -a $d/run.log
done
cd $root
python tools/rocpd_summary.py mfma $out/pass*/pmc_results.db | tee $out/pmc_mfma.txt
