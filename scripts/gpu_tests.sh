#!/bin/bash
# GPU-box visit: full -m gpu test suite (+ optional extra command)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"
tail -n 25 $OUT/pytest_gpu.log
exit $rc
