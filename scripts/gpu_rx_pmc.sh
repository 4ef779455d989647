#!/bin/bash
# SQ counters of k_rx_scan (the automaton route of the regex row) for a few expressions: instruction mix per wave,
# wait share, LDS bank conflicts.  Counters in their own passes (no trace flags next to --pmc).
# Usage: gpu_rx_pmc.sh [case ...]
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out/rx_pmc
CASES=${*:-rx_none rx_alt rx_dotstar rx_word}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in $CASES; do
  d=$OUT/$c
  rm -rf $d; mkdir -p $d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d/stats -- python3 $REPO/scripts/variant_profile.py --case $c --gib 8 > $d/stats.log 2>&1 || { echo "stats pass of $c failed"; tail -3 $d/stats.log; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $d/sq -- python3 $REPO/scripts/variant_profile.py --case $c --gib 8 --iters 2 > $d/sq.log 2>&1 || { echo "sq pass of $c failed"; tail -3 $d/sq.log; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $d/lds -- python3 $REPO/scripts/variant_profile.py --case $c --gib 8 --iters 2 > $d/lds.log 2>&1 || { echo "lds pass of $c failed (counter names?)"; tail -3 $d/lds.log; }
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $REPO/scripts/variant_profile.py --case $c --gib 8 --iters 2 > $d/fetch.log 2>&1 || { echo "fetch pass of $c failed"; tail -3 $d/fetch.log; exit 1; }
  echo "profiled $c"
done
cd $REPO
python3 scripts/variant_summary.py $OUT > $OUT/summary.txt 2>&1
python3 - <<PY >> $OUT/summary.txt
import csv, glob, os, collections
for d in sorted(glob.glob("$OUT/*/lds")):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs: continue
    f = max(fs, key=os.path.getmtime)
    acc = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if ("k_rx_scan" in r["Kernel_Name"] or "k_rx_count" in r["Kernel_Name"])]
    if not rows: continue
    g = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if int(r["Grid_Size"]) == g: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    w = sum(acc["SQ_WAVES"]) / max(len(acc["SQ_WAVES"]), 1)
    print(d.split("/")[-2], "LDS pass per wave:", {k: round(sum(v) / len(v) / w, 1) for k, v in acc.items() if k != "SQ_WAVES"})
PY
cat $OUT/summary.txt | cut -c1-400
exit 0
