#!/bin/bash
# One GPU-box visit made of steps; a step that times out or is killed ends the visit (no further GPU step is started).
# usage: gpu_step.sh name timeout command... [-- name timeout command...]...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
while [ $# -gt 0 ]; do
  name=$1; t=$2; shift 2
  cmd=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do cmd+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  echo "=== $name"
  timeout -k 10 "$t" "${cmd[@]}" > $OUT/$name.log 2>&1
  rc=$?
  echo "rc=$rc"
  tail -n 12 $OUT/$name.log | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED in $name: stopping"; exit 1; fi
  if [ $rc -ne 0 ]; then echo "FAILED $name: stopping"; exit $rc; fi
done
exit 0
