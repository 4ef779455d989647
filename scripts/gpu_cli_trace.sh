#!/bin/bash
# XSG_TRACE marks of fresh xsgrep / my_grep processes on BASELINE config 1's file (and a 2 GiB one), bracketed by the
# launcher's clock (scripts/cli_trace.py); then the `cli` block and config 1's warm job times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/cli_trace.txt
python - > $O 2>&1 <<'PY'
import sys
sys.path.insert(0, "scripts")
import cli_clock
cli_clock.make_file("/dev/shm/xsg_c1.txt", 100_000_000, b"Sherlock")
cli_clock.make_file("/dev/shm/xsg_c2.txt", 2 << 30, b"Sherlock")
print("made files")
PY
for f in /dev/shm/xsg_c1.txt /dev/shm/xsg_c2.txt; do
  python scripts/cli_trace.py --reps 2 -- tools/build/xsgrep -c Sherlock $f >> $O 2>&1
  python scripts/cli_trace.py --reps 2 -- tools/build/xsgrep Sherlock $f >> $O 2>&1
  python scripts/cli_trace.py --reps 2 -- tools/build/my_grep Sherlock $f >> $O 2>&1
done
python scripts/cli_trace.py --reps 2 -- scripts/microbench/build/hip_start >> $O 2>&1
rm -f /dev/shm/xsg_c1.* /dev/shm/xsg_c2.*
timeout -k 10 600 python scripts/cli_clock.py --gib ${CLI_GIB:-10} --reps 3 > gpurun_out/cli_clock.jsonl 2>&1
echo "cli_clock rc=$?" >> $O
timeout -k 10 300 python scripts/config1_e2e.py > gpurun_out/config1_e2e.log 2>&1
echo "config1 rc=$?" >> $O
tail -3 gpurun_out/cli_clock.jsonl | cut -c1-2500
cat gpurun_out/config1_e2e.log | cut -c1-300
