#!/bin/bash
# The class-sequence kinds on two corpora: (a) the bench lexicon, in which `She`, `lock`, `locked`,
# `Sher` are words -- `She lock` is a TRUE match of `She[r ]lock` every ~11 KB -- and (b) the same lexicon without them
# (x-search_amd/corpus.py: LEXICON_PLAIN).  HIP-event rates, then SQ counters (their own rocprofv3 runs).
# usage: gpu_class_ab.sh <tag> [cases...]
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
TAG=${1:-r03}; shift
CASES=${*:-count_Sherlock class_She_r_lock class_Ss_herlock class_digits class_The_az3}
D=$OUT/class_ab_$TAG
rm -rf $D; mkdir -p $D
cd $REPO
for lex in bench plain; do
  : > $D/sweep_$lex.jsonl
  for c in $CASES; do
    timeout -k 10 200 python scripts/variant_profile.py --case $c --gib 20 --lexicon $lex 2>>$D/err.log | grep '^{' >> $D/sweep_$lex.jsonl || { echo "sweep $lex $c failed"; tail -3 $D/err.log; exit 1; }
  done
  cat $D/sweep_$lex.jsonl | cut -c1-330
done
cd /tmp && export TMPDIR=/tmp
for lex in bench plain; do
  for c in class_She_r_lock class_The_az3; do
    d=$D/${c}_$lex
    mkdir -p $d
    timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $d/sq -- python3 $REPO/scripts/variant_profile.py --case $c --gib 20 --iters 2 --lexicon $lex > $d/sq.log 2>&1 || { echo "sq pass of $c/$lex failed"; tail -3 $d/sq.log; exit 1; }
    python3 - "$d" "$c" "$lex" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
d, case, lex = sys.argv[1:4]
want = None
for l in open(d + "/sq.log"):
    if l.startswith("{"):
        want = "void " + json.loads(l)["kernel"].split(" stagger")[0] + "("
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
    out = {"case": case, "lexicon": lex, "kernel": want}
    for k, v in acc.items():
        if k != "SQ_WAVES":
            out[k + "_per_wave"] = round(sum(v) / len(v) / w, 1)
    out["wait_share"] = round(sum(acc["SQ_WAIT_ANY"]) / sum(acc["SQ_WAVE_CYCLES"]), 3)
    print(json.dumps(out))
PY
  done
done | tee $D/sq_summary.jsonl
exit 0
