#!/bin/bash
# scripts/lines_transient.py on a 10 GiB buffer and on the head of a 50 GiB one (with a list search first)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
rm -rf $OUT/lt; mkdir -p $OUT/lt
cd $REPO && python scripts/lines_transient.py 2>/dev/null | cut -c1-400
python scripts/lines_transient.py --buffer-gib 50 --list-first 2>/dev/null | cut -c1-400
