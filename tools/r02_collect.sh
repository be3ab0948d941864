#!/bin/bash
# One GPU-box session that regenerates the round-2 evidence under gpurun_out/r02/ (copied into profiles/ afterwards).
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r02; mkdir -p $o
python bench.py --steps 20 --warmup 5 > $o/bench_c3.json 2> $o/bench_c3.err; echo "bench c3 rc=$?"
python bench.py --steps 20 --warmup 5 --pipelines 1 --no-cpu-baseline > $o/bench_c3_p1.json 2>> $o/bench_c3.err
for w in c2 c4 c5; do python bench.py --workload $w --steps 10 --warmup 3 > $o/bench_$w.json 2> $o/bench_$w.err; echo "bench $w rc=$?"; done
tools/train_matrix.sh > $o/train_matrix.txt 2>&1
for extra in "--rccl1" "--in_dim 4096" "--in_dim 4096 --batch 2048"; do      # data-parallel step on a one-rank RCCL group; the Games recipe's own width
  python tools/train_probe.py --trainer --steps 96 --batch 1024 --bn $extra 2>&1 | tr "\r" "\n" | grep -a "Trainer._train_epoch\|Error\|error" | tail -1 >> $o/train_matrix.txt
done
cat $o/train_matrix.txt
{ python tools/train_gemm_probe.py --rows 1024 2048; python tools/train_gemm_probe.py --rows 1024 --in_dim 4096; } 2>&1 | grep -a "batch\|TF\|total" > $o/train_gemm_probe.txt
python tools/generate_probe.py > $o/generate_probe.txt 2>&1; grep -a "pass 1\|SLOW\|conflict\|index.json" $o/generate_probe.txt
tools/prof_train.sh r02/prof_train_b1024_bn1 --steps 48 --batch 1024 --bn > $o/prof_train_b1024_bn1.txt 2>&1; head -3 $o/prof_train_b1024_bn1.txt
tools/pmc_bench.sh c3 > $o/pmc_c3.txt 2>&1; tail -8 $o/pmc_c3.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$o/prof_bench_c3 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --pipelines 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$o/prof_bench_c3.log 2>&1; echo "rocprof bench rc=$?"
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py stats $o/prof_bench_c3/b_results.db > $o/bench_c3_kernel_stats.csv; head -5 $o/bench_c3_kernel_stats.csv | cut -c1-140
rm -f $o/prof_bench_c3/b_results.db $o/prof_train_b1024_bn1/tr_results.db $o/../pmc_r02/*/*/pmc_results.db
