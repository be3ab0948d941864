#!/bin/bash
# rocprofv3 kernel trace of the training step (Trainer._train_epoch through the captured hipGraph) -> gpurun_out/<name>/tr_results.db,
# then per-kernel statistics as CSV.  Round 2 saw the profiled python once fail to exit after rocprofv3 had written its database
# (7 minutes until the silence watchdog).  The library made GPU calls from an atexit handler then (lcrec_context_destroy:
# hipStreamSynchronize / hipStreamDestroy while the runtime and the profiler's tool library were tearing down); since round 3 it
# makes none at exit (ops._forget_contexts_at_exit) and Trainer.fit releases its graphs itself.  Not seen again in this round's
# six profiled runs; the `timeout` stays as a backstop.
#   tools/prof_train.sh NAME [train_probe args...]
set -u
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $out -o tr -- python3 $GRAFT_REPO_ROOT/tools/train_probe.py --trainer "$@" > $out/run.log 2>&1
echo "rocprofv3 exit $?" >> $out/run.log
cd $GRAFT_REPO_ROOT
grep -a "Trainer._train_epoch" $out/run.log | tr '\r' '\n' | grep -a "ms/step" | tail -1
python tools/rocpd_summary.py stats $out/tr_results.db > $out/kernel_stats.csv && head -30 $out/kernel_stats.csv | cut -c1-150
