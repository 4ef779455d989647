#!/bin/bash
# ignore_case needles that are dense in the text (letters only: decided on data | 0x20), 50 GiB, default path
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in icase_the icase_lines_the icase_that icase_Holmes icase_Sherlock; do python scripts/variant_profile.py --case $c --gib 50 2>/dev/null | grep '^{' | cut -c1-700; done
