#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
rm -rf $OUT/lt; mkdir -p $OUT/lt
cd $REPO && python scripts/lines_transient.py 2>/dev/null | cut -c1-400
python scripts/lines_transient.py --buffer-gib 50 --list-first 2>/dev/null | cut -c1-400
