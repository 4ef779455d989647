cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in long_detective_street long_Sherlock_Holmes lines_e lines_the icase_lines_the lines_Sherlock count_Sherlock mask1_e mask1_the one_that; do python scripts/variant_profile.py --case $c --gib 50 2>/dev/null | grep '^{' ; done
