#!/bin/bash
# HBM read traffic (FETCH_SIZE, its own rocprofv3 pass; x 2 x 1024 bytes on gfx950) of the bulk kernel of the given
# variant_profile cases on a 20 GiB shard, over the bytes scanned.  usage: gpu_fetch.sh <tag> case...
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
TAG=$1; shift
D=$OUT/fetch_$TAG
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  d=$D/$c; mkdir -p $d
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $REPO/scripts/variant_profile.py --case $c --gib 20 --iters 2 > $d/run.log 2>&1 || { echo "fetch pass of $c failed"; tail -3 $d/run.log; exit 1; }
  python3 - "$d" "$c" <<'PY'
import csv, glob, json, sys
d, case = sys.argv[1:3]
want, nbytes = None, None
for l in open(d + "/run.log"):
    if l.startswith("{"):
        j = json.loads(l)
        want = "void " + j["kernel"].split(" stagger")[0] + "("
        nbytes = j["bytes"]
fs = []
for f in glob.glob(d + "/fetch/*/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(want) and r["Counter_Name"] == "FETCH_SIZE"]
    if rows:
        g = max(int(r["Grid_Size"]) for r in rows)
        fs += [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == g]
if fs and nbytes:
    print(json.dumps({"case": case, "kernel": want, "launches": len(fs), "read_traffic_over_algorithmic": round(2.0 * 1024.0 * sum(fs) / len(fs) / nbytes, 4)}))
PY
done | tee $D/fetch_summary.jsonl
exit 0
