#!/bin/bash
# 4..8-byte needles of the bench corpus with the byte-parallel route pinned off / on (XSG_DENSE_PER; note_density, xsg_api.cpp)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
CASES=one_that,lines_that,mask2_Holmes,lines_Holmes,mask2_Sherl,lines_Sherl,mask2_detecti,lines_detecti,two_detectiv,lines_detectiv,count_Sherlock,lines_Sherlock
for per in 1 1099511627776; do
  echo "# XSG_DENSE_PER=$per" | tee -a gpurun_out/density_bench.jsonl
  XSG_DENSE_PER=$per python scripts/variant_profile.py --gib ${GIB:-20} --case $CASES 2>/dev/null | grep '^{' | tee -a gpurun_out/density_bench.jsonl
done
