#!/bin/bash
# config 5 (LZ4 metafile): how many decoder threads and device workers inside the 16-CPU quota?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export XSG_BENCH_CLI=0
for rep in 1 2; do for w in 4; do for dcd in 12 14 16; do
  XSG_E2E_WORKERS=$w XSG_E2E_DECODERS=$dcd python bench.py --gib-per-gpu 4 --steps 3 --warmup 1 --e2e-gib 8 --no-cpu-baseline --configs-gib 0 --no-regex 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())['e2e']; print('workers', $w, 'decoders', $dcd, 'lz4', d['lz4_metafile_count_gib_s'], 'per thread', d['lz4']['decode_gib_s_per_thread'], 'count', d['count_gib_s'])"
done; done; done
