#!/bin/bash
# Bench + profiles on the GPU box.  Usage: gpu_bench.sh [tag]
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
TAG=${1:-r01}
OUT=$REPO/gpurun_out
mkdir -p $OUT
step() {
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a $OUT/bench_round.log
  timeout -k 10 "$t" "$@" > $OUT/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $OUT/bench_round.log
  tail -n 4 $OUT/$name.log | cut -c1-1500 | tee -a $OUT/bench_round.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED in $name: stopping" | tee -a $OUT/bench_round.log; exit 1; fi
  return $rc
}
: > $OUT/bench_round.log
rm -rf $OUT/prof_${TAG}_stats $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write  # gpurun_out/ is merged across calls: no stale CSVs
step bench_4g 600 python bench.py --gib-per-gpu 4 --steps 10 --warmup 2 --cpu-seconds 6 || exit 1
step bench_50g 900 python bench.py || exit 1
grep '^{' $OUT/bench_50g.log > $OUT/BENCH_${TAG}_local.json
cd /tmp && export TMPDIR=/tmp
step prof_stats 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel-iters 10 --no-regex
step prof_fetch 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-iters 1 --no-regex
step prof_write 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-iters 1 --no-regex
cd $REPO
find $OUT/prof_${TAG}_stats -name '*stats*.csv' | head | tee -a $OUT/bench_round.log
for f in $(find $OUT/prof_${TAG}_stats -name '*kernel_stats.csv' | head -1); do head -12 $f | cut -c1-300 | tee -a $OUT/bench_round.log; done
exit 0
