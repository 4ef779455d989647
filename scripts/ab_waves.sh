#!/bin/bash
# A/B of an occupancy request on k_scan (amdgpu_waves_per_eu(8, 8) on the class-sequence kinds / on all kinds / none): three
# builds of the library (the variants were macros in xsg_kernels.hip at commit "Class-sequence kinds get their own kernel
# entry ..."; the request was not adopted), the class-sequence and a few other variants on one 50 GiB shard each, two
# interleaved rounds on the same box.  Kept as the record of how profiles/r02_ab_waves.txt was made.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out/ab_waves
mkdir -p $OUT
cd $REPO
for round in 1 2; do
  for lib in libxsg.so libxsg_ab_all.so libxsg_ab_none.so; do
    XSG_LIB=$REPO/x-search_amd/lib/$lib timeout -k 10 150 python scripts/variant_profile.py --case all --gib 50 > $OUT/${lib%.so}_$round.jsonl 2> $OUT/err.log || { echo "$lib failed"; tail -3 $OUT/err.log; exit 1; }
  done
done
python3 - <<PY
import json, glob
cases = ["count_Sherlock", "count_nl_Sherlock", "mask1_e", "one_that", "class_She_r_lock", "class_Ss_herlock", "class_digits", "class_The_az3"]
print("case | default (cls only) r1 r2 | all kernels r1 r2 | none r1 r2   [TB/s, 50 GiB]")
d = {}
for f in glob.glob("$OUT/*.jsonl"):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); d[(f.split("/")[-1][:-6], j["case"])] = j["tb_s"]
for c in cases:
    print(c, "|", *[f"{d.get((lib + '_' + r, c), 0):.3f}" for lib in ("libxsg", "libxsg_ab_all", "libxsg_ab_none") for r in "12"])
PY
exit 0
