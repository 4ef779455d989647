#!/bin/bash
# the dense / long literal variants and the class kinds on the 50 GiB shard, default path (profiles/r03_dense_variants.txt)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in count_Sherlock count_nl_Sherlock lines_Sherlock mask1_e mask1_the one_that icase_that lines_e lines_the class_The_az3 class_She_r_lock long_detective_street; do python scripts/variant_profile.py --case $c --gib 50 2>/dev/null | grep '^{' ; done
