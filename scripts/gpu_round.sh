#!/bin/bash
# One GPU-box visit: parity tests (both neighbour-fetch variants), a short and a
# full bench.  Every GPU step runs under its own timeout; a step that times out
# or is killed ends the visit (no further GPU step is started).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
step() {  # name, timeout, command...
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a $OUT/round.log
  timeout -k 10 "$t" "$@" > $OUT/$name.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $OUT/round.log
  tail -n 6 $OUT/$name.log | tee -a $OUT/round.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED in $name: stopping" | tee -a $OUT/round.log; exit 1; fi
  return $rc
}
: > $OUT/round.log
rocminfo | grep -E "Marketing Name|gfx9" | head -4 | tee -a $OUT/round.log
nproc | tee -a $OUT/round.log
free -g | head -2 | tee -a $OUT/round.log
step pytest_dpp 900 python -m pytest tests -m gpu -x -q
DPP_RC=$?
XSG_LIB=$PWD/x-search_amd/lib/libxsg_shfl.so step pytest_shfl 900 python -m pytest tests -m gpu -x -q
SHFL_RC=$?
if [ $DPP_RC -ne 0 ] && [ $SHFL_RC -ne 0 ]; then echo "both variants fail parity: no bench" | tee -a $OUT/round.log; exit 1; fi
if [ $DPP_RC -ne 0 ]; then export XSG_LIB=$PWD/x-search_amd/lib/libxsg_shfl.so; echo "using shfl variant" | tee -a $OUT/round.log; fi
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step bench_4g 600 python bench.py --gib-per-gpu 4 --steps 10 --warmup 2 --cpu-seconds 6
step bench_50g 900 python bench.py
exit 0
