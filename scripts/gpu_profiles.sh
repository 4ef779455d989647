#!/bin/bash
# The bench command under rocprofv3: kernel stats, then FETCH_SIZE and WRITE_SIZE in their own passes (no trace flags
# next to --pmc), then profiles/<tag>_pmc_traffic.json.  Usage: gpu_profiles.sh <tag>
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
TAG=${1:-r02}
mkdir -p $OUT
rm -rf $OUT/prof_${TAG}_stats $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write
cd $REPO
timeout -k 10 600 python bench.py --e2e-gib 0 --no-cpu-baseline > $OUT/bench_${TAG}_plain.log 2>&1 || { tail -3 $OUT/bench_${TAG}_plain.log; exit 1; }
grep '^{' $OUT/bench_${TAG}_plain.log > $OUT/BENCH_${TAG}_plain.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --e2e-gib 0 --kernel-iters 10 --no-regex > $OUT/prof_${TAG}_stats.log 2>&1 || { echo stats failed; tail -3 $OUT/prof_${TAG}_stats.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-gib 0 --kernel-iters 1 --no-tune --no-regex > $OUT/prof_${TAG}_fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-gib 0 --kernel-iters 1 --no-tune --no-regex > $OUT/prof_${TAG}_write.log 2>&1 || { echo write failed; exit 1; }
cd $REPO
F=$(find $OUT/prof_${TAG}_fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/prof_${TAG}_write -name '*counter_collection.csv' | head -1)
python3 scripts/pmc_traffic.py $F $W $OUT/BENCH_${TAG}_plain.json $TAG | tee $OUT/${TAG}_pmc_traffic.json | cut -c1-600
for f in $(find $OUT/prof_${TAG}_stats -name '*kernel_stats.csv' | head -1); do cp $f $OUT/${TAG}_bench50g_kernel_stats.csv; head -8 $f | cut -c1-200; done
exit 0
