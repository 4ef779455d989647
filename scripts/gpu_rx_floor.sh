#!/bin/bash
# k_rx_scan's staging floor: expressions with few trigger bytes on the bench corpus and on the
# corpus without capital S / H words (x-search_amd/corpus.py: LEXICON_NOSH), HIP-event rates, then SQ counters.
# usage: gpu_rx_floor.sh <tag>
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
TAG=${1:-r03}
D=$OUT/rx_floor_$TAG
rm -rf $D; mkdir -p $D
cd $REPO
for lex in bench nosh; do
  : > $D/sweep_$lex.jsonl
  for c in rx_none rx_alt rx_dotstar rx_optional rx_lines_alt rx_word; do
    timeout -k 10 200 python scripts/variant_profile.py --case $c --gib 20 --lexicon $lex 2>>$D/err.log | grep '^{' >> $D/sweep_$lex.jsonl || { echo "sweep $lex $c failed"; tail -3 $D/err.log; exit 1; }
  done
  cut -c1-300 $D/sweep_$lex.jsonl
done
cd /tmp && export TMPDIR=/tmp
for c in rx_none rx_alt; do
  d=$D/${c}_nosh; mkdir -p $d
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $d/sq -- python3 $REPO/scripts/variant_profile.py --case $c --gib 20 --iters 2 --lexicon nosh > $d/sq.log 2>&1 || { echo "sq pass failed"; tail -3 $d/sq.log; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $REPO/scripts/variant_profile.py --case $c --gib 20 --iters 2 --lexicon nosh > $d/fetch.log 2>&1 || { echo "fetch pass failed"; exit 1; }
done
cd $REPO
python3 - "$D" <<'PY' | tee $D/pmc_summary.jsonl
import csv, glob, json, sys
from collections import defaultdict
D = sys.argv[1]
for case in ("rx_none", "rx_alt"):
    d = f"{D}/{case}_nosh"
    nbytes = None
    for l in open(d + "/sq.log"):
        if l.startswith("{"):
            nbytes = json.loads(l)["bytes"]
    acc = defaultdict(list)
    for f in glob.glob(d + "/sq/*/*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if ("k_rx_scan" in r["Kernel_Name"] or "k_rx_count" in r["Kernel_Name"])]
        if rows:
            g = max(int(r["Grid_Size"]) for r in rows)
            for r in rows:
                if int(r["Grid_Size"]) == g:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"case": case, "lexicon": "nosh"}
    if acc.get("SQ_WAVES"):
        w = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
        for k, v in acc.items():
            if k != "SQ_WAVES":
                out[k + "_per_wave"] = round(sum(v) / len(v) / w, 1)
        out["wait_share"] = round(sum(acc["SQ_WAIT_ANY"]) / sum(acc["SQ_WAVE_CYCLES"]), 3)
    fs = []
    for f in glob.glob(d + "/fetch/*/*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if ("k_rx_scan" in r["Kernel_Name"] or "k_rx_count" in r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE"]
        if rows:
            g = max(int(r["Grid_Size"]) for r in rows)
            fs += [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == g]
    if fs and nbytes:
        out["read_traffic_over_algorithmic"] = round(2.0 * 1024.0 * sum(fs) / len(fs) / nbytes, 4)
    print(json.dumps(out))
PY
exit 0
