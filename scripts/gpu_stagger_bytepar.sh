#!/bin/bash
# the wave stagger of the byte-parallel kinds after the per-tile epilogue (XSG_TUNE pins it for every variant), warm timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
fmt() { grep '^{' | python -c "
import sys,json
print(' '.join(f\"{json.loads(l)['case']}={json.loads(l)['frac_of_8tbs']}\" for l in sys.stdin))"; }
for round in 1 2; do for st in 0 2 4 6 8; do
  echo -n "XSG_TUNE=$st r$round: "; XSG_TUNE=$st python scripts/variant_profile.py --gib 20 --case mask1_e,mask1_the,icase_the,one_that,icase_that,lines_e,lines_the,icase_lines_the 2>/dev/null | fmt
done; done
