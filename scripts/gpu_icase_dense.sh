#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in icase_the icase_lines_the icase_that icase_Holmes icase_Sherlock; do python scripts/variant_profile.py --case $c --gib 50 2>/dev/null | grep '^{' | cut -c1-700; done
