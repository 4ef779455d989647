#!/bin/bash
# The bench command under rocprofv3: kernel stats, then FETCH_SIZE and WRITE_SIZE in their own passes (no trace flags
# next to --pmc) for BOTH hot-filter instantiations of the timed kernel at the stagger the timed run measured (XSG_HOT /
# XSG_TUNE pin them: what xsg_shard_tune picks between the two moves from box to box), then
# profiles/<tag>_pmc_traffic.json.  Usage: gpu_profiles.sh <tag>
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
TAG=${1:-r03}
mkdir -p $OUT
export XSG_BENCH_CLI=0  # the cli block starts fresh xsgrep / grep processes: not under the profiler, not in these runs
rm -rf $OUT/prof_${TAG}_stats $OUT/prof_${TAG}_fetch* $OUT/prof_${TAG}_write*
cd $REPO
timeout -k 10 600 python bench.py --e2e-gib 0 --no-cpu-baseline --configs-gib 0 --no-regex > $OUT/bench_${TAG}_plain.log 2>&1 || { tail -3 $OUT/bench_${TAG}_plain.log; exit 1; }
grep '^{' $OUT/bench_${TAG}_plain.log > $OUT/BENCH_${TAG}_plain.json
ST=$(python3 -c "import json;d=json.load(open('$OUT/BENCH_${TAG}_plain.json'));print(d['roofline']['stagger'] if d['roofline']['stagger'] is not None else 16)")
echo "timed: $(python3 -c "import json;d=json.load(open('$OUT/BENCH_${TAG}_plain.json'));print(d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])")"
cd /tmp && export TMPDIR=/tmp
SMALL="--no-cpu-baseline --e2e-gib 0 --configs-gib 0 --no-regex"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -- python3 $REPO/bench.py --steps 20 --warmup 3 --kernel-iters 10 $SMALL > $OUT/prof_${TAG}_stats.log 2>&1 || { echo stats failed; tail -3 $OUT/prof_${TAG}_stats.log; exit 1; }
ARGS=""
for hot in 0 1; do
  XSG_HOT=$hot XSG_TUNE=$ST timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch$hot -- python3 $REPO/bench.py --steps 2 --warmup 1 --kernel-iters 1 --no-tune $SMALL > $OUT/prof_${TAG}_fetch$hot.log 2>&1 || { echo fetch $hot failed; tail -3 $OUT/prof_${TAG}_fetch$hot.log; exit 1; }
  XSG_HOT=$hot XSG_TUNE=$ST timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write$hot -- python3 $REPO/bench.py --steps 2 --warmup 1 --kernel-iters 1 --no-tune $SMALL > $OUT/prof_${TAG}_write$hot.log 2>&1 || { echo write $hot failed; exit 1; }
  F=$(find $OUT/prof_${TAG}_fetch$hot -name '*counter_collection.csv' | head -1)
  W=$(find $OUT/prof_${TAG}_write$hot -name '*counter_collection.csv' | head -1)
  ARGS="$ARGS $F $W"
done
cd $REPO
python3 scripts/pmc_traffic.py $OUT/BENCH_${TAG}_plain.json $TAG $ARGS > $OUT/${TAG}_pmc_traffic.json || exit 1
python3 -c "import json;d=json.load(open('$OUT/${TAG}_pmc_traffic.json'));[print(k, v['ratio_traffic_over_algorithmic'], v['hbm_bytes_per_launch']) for k,v in d['by_kernel'].items()]"
for f in $(find $OUT/prof_${TAG}_stats -name '*kernel_stats.csv' | head -1); do cp $f $OUT/${TAG}_bench50g_kernel_stats.csv; head -8 $f | cut -c1-200; done
for f in $(find $OUT/prof_${TAG}_stats -name '*kernel_trace.csv' | head -1); do python3 scripts/trace_by_grid.py $f > $OUT/${TAG}_bench50g_kernel_trace_by_grid.txt; done
exit 0
