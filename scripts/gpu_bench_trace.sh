#!/bin/bash
# kernel trace of a bench run with its configs leg (which kernels the list / count_lines calls of the 10 GiB sub-shard launch, and how long they take)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
rm -rf $OUT/bench_trace; mkdir -p $OUT/bench_trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $REPO/bench.py --e2e-gib 0 --no-cpu-baseline --no-regex > $OUT/bench_trace/run.log 2>&1 || { tail -5 $OUT/bench_trace/run.log; exit 1; }
f=$(ls $OUT/bench_trace/*/*_kernel_stats.csv | head -1)
head -30 $f | cut -c1-200
