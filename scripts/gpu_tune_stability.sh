#!/bin/bash
# does xsg_shard_tune come back with the same (filter, stagger) every time?  N bench runs without side legs, one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export XSG_BENCH_CLI=0
for i in $(seq 1 ${1:-4}); do
  python bench.py --e2e-gib 0 --no-cpu-baseline --configs-gib 0 --no-regex 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'], 'setup_s', d['config'].get('setup_s'))"
done
