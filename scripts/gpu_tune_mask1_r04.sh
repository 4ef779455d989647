#!/bin/bash
# which wave stagger suits the byte-parallel kinds now that their loop is shorter (xsg_shard_tune: all staggers, full size)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python scripts/variant_profile.py --gib ${GIB:-50} --tune --case mask1_e,mask1_the,lines_e,lines_the,one_that,mask2_Holmes,icase_the 2>/dev/null | grep '^{' | tee gpurun_out/tune_mask1_r04.jsonl
