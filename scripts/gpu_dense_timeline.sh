#!/bin/bash
# one time axis for a dense xs::lines search (`She`, 10 GiB): kernels and copies (rocprofv3 kernel + memory-copy trace)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
rm -rf $OUT/dense_tl; mkdir -p $OUT/dense_tl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/dense_tl -- python3 $REPO/scripts/config_times.py --gib 10 --reps 2 --patterns ${PAT:-She} > $OUT/dense_tl/run.log 2>&1 || { tail -5 $OUT/dense_tl/run.log; exit 1; }
grep '^{' $OUT/dense_tl/run.log | cut -c1-200
cd $REPO
python3 scripts/timeline.py $OUT/dense_tl --last k_line_gather --window-ms 130 > $OUT/dense_timeline.txt
cat $OUT/dense_timeline.txt
