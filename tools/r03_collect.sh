#!/bin/bash
# One GPU-box session that regenerates the round-3 evidence under gpurun_out/r03/ (copied into profiles/ afterwards).
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03; mkdir -p $o
python bench.py --steps 20 --warmup 5 > $o/bench_c3.json 2> $o/bench_c3.err; echo "bench c3 rc=$?"
# (c2: a pass is 2.7 ms -- 40 of them; its per-launch brackets are taken in an untimed pass, bench.py --trace-pass; the same command with
#  --trace-pass timed beside it shows what the marker packets cost)
python bench.py --workload c2 --steps 40 --warmup 10 > $o/bench_c2.json 2> $o/bench_c2.err; echo "bench c2 rc=$?"
python bench.py --workload c2 --steps 40 --warmup 10 --trace-pass timed --no-cpu-baseline --no-secondary > $o/bench_c2_timed_brackets.json 2> $o/bench_c2_timed.err; echo "bench c2 (brackets in the timed region) rc=$?"
for w in c4 c5; do python bench.py --workload $w --steps 10 --warmup 3 > $o/bench_$w.json 2> $o/bench_$w.err; echo "bench $w rc=$?"; done
# the training step: run.sh recipe (bn=True, batch 1024) at 768-d and 4096-d, batch 2048, bn=False; fused-BatchNorm opt-in; one-rank RCCL group
{
for args in "--batch 1024 --bn" "--batch 1024" "--batch 2048 --bn" "--batch 1024 --bn --in_dim 4096" "--batch 2048 --bn --in_dim 4096" "--batch 1024 --bn --rccl1"; do
  python tools/train_probe.py --trainer --steps 96 $args 2>&1 | tr "\r" "\n" | grep -a "Trainer._train_epoch\|Error\|error" | tail -1
done
echo "# LCREC_FUSE_BN=1 (BatchNorm folded into the GEMMs, opt-in):"
LCREC_FUSE_BN=1 python tools/train_probe.py --trainer --steps 96 --batch 1024 --bn 2>&1 | tr "\r" "\n" | grep -a "Trainer._train_epoch\|Error" | tail -1
echo "# LCREC_TICKETS=0 (reduction tails in launches of their own, as in round 2):"
LCREC_TICKETS=0 python tools/train_probe.py --trainer --steps 96 --batch 1024 --bn 2>&1 | tr "\r" "\n" | grep -a "Trainer._train_epoch\|Error" | tail -1
} > $o/train_matrix.txt 2>&1
cat $o/train_matrix.txt
python tools/fused_fwd_probe.py 2>&1 | grep -a "\->\|total" > $o/fused_bn_probe.txt; cat $o/fused_bn_probe.txt
tools/prof_train.sh r03/prof_train_b1024_bn1 --steps 48 --batch 1024 --bn > $o/prof_train_b1024_bn1.txt 2>&1; head -3 $o/prof_train_b1024_bn1.txt
LCREC_FUSE_BN=1 tools/prof_train.sh r03/prof_train_b1024_bn1_fused --steps 48 --batch 1024 --bn > $o/prof_train_b1024_bn1_fused.txt 2>&1; head -3 $o/prof_train_b1024_bn1_fused.txt
python tools/generate_probe.py > $o/generate_probe.txt 2>&1; grep -a "pass 1\|SLOW\|conflict\|index.json" $o/generate_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$o/prof_bench_c3 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$o/prof_bench_c3.log 2>&1; echo "rocprof bench rc=$?"
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py stats $o/prof_bench_c3/b_results.db > $o/bench_c3_kernel_stats.csv; head -5 $o/bench_c3_kernel_stats.csv | cut -c1-140
grep -a "^{\"metric" $o/prof_bench_c3.log | tail -1 > $o/bench_c3_profiled.json
rm -f $o/prof_bench_c3/b_results.db $o/prof_train_b1024_bn1/tr_results.db $o/prof_train_b1024_bn1_fused/tr_results.db
