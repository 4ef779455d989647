#!/bin/bash
# Per-variant evidence: HIP-event sweep of every k_scan variant, then rocprofv3 passes (kernel stats, SQ counters,
# FETCH_SIZE -- counters in their own runs, no trace flags next to --pmc) for the cases named on the command line.
# Usage: gpu_variants.sh <tag> [case ...]
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
TAG=${1:-r02}; shift
CASES=${*:-count_nl_Sherlock lines_e icase_that long_detective_street class_She_r_lock class_The_az3 mask1_e rx_none rx_alt rx_dotstar rx_word}
mkdir -p $OUT/variants_$TAG
cd $REPO
timeout -k 10 600 python scripts/variant_profile.py --case all --gib 50 > $OUT/variants_$TAG/sweep.jsonl 2> $OUT/variants_$TAG/sweep.err || { echo "sweep failed"; tail -5 $OUT/variants_$TAG/sweep.err; exit 1; }
cat $OUT/variants_$TAG/sweep.jsonl | cut -c1-260
timeout -k 10 600 python scripts/variant_profile.py --case all --gib 50 --tune > $OUT/variants_$TAG/sweep_tuned.jsonl 2>> $OUT/variants_$TAG/sweep.err || { echo "tuned sweep failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
for c in $CASES; do
  d=$OUT/variants_$TAG/$c
  rm -rf $d; mkdir -p $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d/stats -- python3 $REPO/scripts/variant_profile.py --case $c --gib 20 > $d/stats.log 2>&1 || { echo "stats pass of $c failed"; tail -3 $d/stats.log; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $d/sq -- python3 $REPO/scripts/variant_profile.py --case $c --gib 20 --iters 2 > $d/sq.log 2>&1 || { echo "sq pass of $c failed"; tail -3 $d/sq.log; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $REPO/scripts/variant_profile.py --case $c --gib 20 --iters 2 > $d/fetch.log 2>&1 || { echo "fetch pass of $c failed"; tail -3 $d/fetch.log; exit 1; }
  echo "profiled $c"
done
cd $REPO
python3 scripts/variant_summary.py $OUT/variants_$TAG > $OUT/variants_$TAG/summary.txt 2>&1
cat $OUT/variants_$TAG/summary.txt | cut -c1-220
exit 0
