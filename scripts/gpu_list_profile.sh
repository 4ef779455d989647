#!/bin/bash
# Kernel-level breakdown of the list tags on a device-resident 10 GiB shard (BASELINE configs 2 and 4).
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
rm -rf $OUT/prof_list
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_list -- python3 $REPO/scripts/config_times.py --gib 10 --reps 5 > $OUT/list_profile.log 2>&1
rc=$?
cd $REPO
grep '^{' $OUT/list_profile.log | grep -E "Sherlock" | cut -c1-200
for f in $(find $OUT/prof_list -name '*kernel_stats.csv' | head -1); do head -30 $f | cut -c1-220; done
exit $rc
