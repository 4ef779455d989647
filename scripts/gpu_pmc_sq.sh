#!/bin/bash
# One extra PMC pass over the bench: instruction mix and wait share of k_scan.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/prof_sq -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-iters 1 --no-regex > $OUT/prof_sq.log 2>&1
rc=$?
echo "rc=$rc"
tail -3 $OUT/prof_sq.log | cut -c1-300
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/prof_sq/*/*counter_collection.csv")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_scan" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, len(v), sum(v) / len(v))
PY
exit $rc
