#!/bin/bash
# the dense literal variants on the 50 GiB shard, one process (profiles/r04_dense_variants.txt)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python scripts/variant_profile.py --gib ${GIB:-50} --case count_Sherlock,count_nl_Sherlock,lines_Sherlock,mask1_e,mask1_the,lines_e,lines_the,lines_She,icase_the,icase_lines_the,one_that,mask2_Holmes,icase_that 2>/dev/null | grep '^{' | tee gpurun_out/dense_r04.jsonl
