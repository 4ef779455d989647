#!/bin/bash
# what xsg_shard_tune picks for the short-needle kernels (stagger), 50 GiB
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in mask1_e mask1_the lines_e lines_the one_that; do python scripts/variant_profile.py --case $c --gib 50 --tune 2>/dev/null | grep '^{' | cut -c1-600; done
