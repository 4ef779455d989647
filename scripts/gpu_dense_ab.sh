#!/bin/bash
# dense 4..8-byte needles: the byte-parallel route against the hot filter + slow path (XSG_DENSE_BYTES=0), same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in one_that mask2_Holmes; do
  for v in 1 0; do
    echo "XSG_DENSE_BYTES=$v"
    XSG_DENSE_BYTES=$v python scripts/variant_profile.py --case $c --gib 50 2>/dev/null | grep '^{' | cut -c1-700
    XSG_DENSE_BYTES=$v python scripts/variant_profile.py --case $c --gib 50 --tune 2>/dev/null | grep '^{' | cut -c1-700
  done
done
for c in mask1_e lines_e mask1_the; do python scripts/variant_profile.py --case $c --gib 50 2>/dev/null | grep '^{' | cut -c1-700; done
