set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for g in 256 1024 8192 65536; do
rm -rf $OUT/prof_eg
XSG_EMIT_GRID=$g timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eg -- python3 $REPO/scripts/config_times.py --gib 10 --reps 3 --patterns Sherlock > $OUT/eg.log 2>&1
for f in $(find $OUT/prof_eg -name '*kernel_stats.csv' | head -1); do echo "grid $g: $(grep 'false, false, true, 4' $f | cut -c1-140)"; done
done
