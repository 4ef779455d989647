#!/bin/bash
# Round-2 GPU visit: host facts, parity tests, fixed-cost probe, bench (N=1 and the N=2 self-launch rehearsal).
# Every GPU step runs under its own timeout; a step that times out or is killed ends the visit.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
WHAT=${1:-all}
step() {  # name, timeout, command...
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a $OUT/r2.log
  timeout -k 10 "$t" "$@" > $OUT/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $OUT/r2.log
  tail -n 5 $OUT/$name.log | cut -c1-3000 | tee -a $OUT/r2.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED in $name: stopping" | tee -a $OUT/r2.log; exit 1; fi
  return $rc
}
: > $OUT/r2.log
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA"; free -g | head -2; df -h /dev/shm | tail -1; } 2>&1 | tee $OUT/host_facts.txt | tee -a $OUT/r2.log
case $WHAT in
  all|tests) step pytest_gpu 1100 python -m pytest tests -m gpu -x -q || exit 1 ;;
esac
case $WHAT in
  all|fixed) step fixed_cost 500 python scripts/fixed_cost.py --modes count,count_lines,count+nl || exit 1 ;;
esac
case $WHAT in
  all|bench)
    step bench_50g 900 python bench.py || exit 1
    grep '^{' $OUT/bench_50g.log > $OUT/BENCH_r02_local.json
    XSG_BENCH_BACKEND=gloo step bench_n2_gloo 900 python bench.py --gpus 2 --gib-per-gpu 8 --steps 10 --warmup 2 --e2e-gib 1 || exit 1
    ;;
esac
exit 0
