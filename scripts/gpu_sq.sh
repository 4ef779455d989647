#!/bin/bash
# SQ counters per wave (= per 4 KiB of text) of the bulk kernel of the given variant_profile cases on a 20 GiB shard.
# usage: gpu_sq.sh <tag> case...
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
TAG=$1; shift
D=$OUT/sq_$TAG
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  d=$D/$c
  mkdir -p $d
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $d/sq -- python3 $REPO/scripts/variant_profile.py --case $c --gib 20 --iters 2 > $d/sq.log 2>&1 || { echo "sq pass of $c failed"; tail -3 $d/sq.log; exit 1; }
  python3 - "$d" "$c" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
d, case = sys.argv[1:3]
want = None
ms = None
for l in open(d + "/sq.log"):
    if l.startswith("{"):
        j = json.loads(l)
        want = "void " + j["kernel"].split(" stagger")[0] + "("
acc = defaultdict(list)
for f in glob.glob(d + "/sq/*/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(want)]
    if rows:
        g = max(int(r["Grid_Size"]) for r in rows)
        for r in rows:
            if int(r["Grid_Size"]) == g:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc.get("SQ_WAVES"):
    w = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
    out = {"case": case, "kernel": want}
    for k, v in acc.items():
        if k != "SQ_WAVES":
            out[k + "_per_wave"] = round(sum(v) / len(v) / w, 1)
    out["wait_share"] = round(sum(acc["SQ_WAIT_ANY"]) / sum(acc["SQ_WAVE_CYCLES"]), 3)
    print(json.dumps(out))
PY
done | tee $D/sq_summary.jsonl
exit 0
