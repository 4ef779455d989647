#!/bin/bash
# Kernel-level breakdown of the list tags on a device-resident 10 GiB shard (BASELINE configs 2 and 4).
# usage: gpu_list_profile.sh [tag]   -> gpurun_out/<tag>_config_times.log, <tag>_list_kernel_stats.csv, <tag>_list_kernel_trace.csv
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r03}"
OUT=$REPO/gpurun_out
rm -rf $OUT/prof_list
cd $REPO
timeout -k 10 300 python3 scripts/config_times.py --gib 10 --reps 7 > $OUT/${TAG}_config_times.log 2>&1 || { tail -5 $OUT/${TAG}_config_times.log; exit 1; }
grep '^{' $OUT/${TAG}_config_times.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_list -- python3 $REPO/scripts/config_times.py --gib 10 --reps 5 --patterns Sherlock > $OUT/${TAG}_list_profile.log 2>&1
rc=$?
cd $REPO
for f in $(find $OUT/prof_list -name '*kernel_stats.csv' | head -1); do cp $f $OUT/${TAG}_list_kernel_stats.csv; head -40 $f | cut -c1-200; done
for f in $(find $OUT/prof_list -name '*kernel_trace.csv' | head -1); do cp $f $OUT/${TAG}_list_kernel_trace.csv; done
exit $rc
