#!/bin/bash
# one sparse xs::lines search (`Sherlock`, 10 GiB) kernel by kernel (rocprofv3 kernel trace of scripts/config_times.py)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
rm -rf $OUT/sparse_tl; mkdir -p $OUT/sparse_tl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/sparse_tl -- python3 $REPO/scripts/config_times.py --gib 10 --reps 3 --patterns Sherlock > $OUT/sparse_tl/run.log 2>&1 || { tail -5 $OUT/sparse_tl/run.log; exit 1; }
grep '^{' $OUT/sparse_tl/run.log | cut -c1-200
cd $REPO
python3 scripts/timeline.py $OUT/sparse_tl --last k_line_gather --window-ms 1.9 --min-us 0 > $OUT/sparse_timeline.txt
cat $OUT/sparse_timeline.txt
