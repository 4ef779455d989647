#!/bin/bash
# kernel stats of scripts/config_times.py for the given patterns (10 GiB)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
rm -rf $OUT/cfg_trace; mkdir -p $OUT/cfg_trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_trace -- python3 $REPO/scripts/config_times.py --gib 10 --patterns "$1" > $OUT/cfg_trace/run.log 2>&1 || { tail -5 $OUT/cfg_trace/run.log; exit 1; }
f=$(ls $OUT/cfg_trace/*/*_kernel_stats.csv | head -1)
head -16 $f | cut -c1-170
