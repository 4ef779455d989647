#!/bin/bash
# Where does a fresh process's first search go?    HIP start-up floor, XSG_TRACE marks of xsgrep /
# my_grep on BASELINE config 1's file, then the whole `cli` block (scripts/cli_clock.py) and config 1's warm job times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/cli_start.txt
{
  echo "== host"; nproc; cat /sys/fs/cgroup/cpu.max; grep --version | head -1; df -h /dev/shm | tail -1
  echo "== HIP start-up floor (scripts/microbench/hip_start.hip), three fresh processes"
  for i in 1 2 3; do scripts/microbench/build/hip_start; echo; done
} > $O 2>&1
python - >> $O 2>&1 <<'PY'
import sys
sys.path.insert(0, "scripts")
import cli_clock
cli_clock.make_file("/dev/shm/xsg_c1.txt", 100_000_000, b"Sherlock")
print("made /dev/shm/xsg_c1.txt")
PY
{
  for prog in "tools/build/xsgrep -c" "tools/build/xsgrep" "tools/build/my_grep"; do
    for i in 1 2; do
      echo "== XSG_TRACE=1 $prog Sherlock /dev/shm/xsg_c1.txt  (run $i)"
      /usr/bin/time -f "wall %e s user %U s sys %S s" env XSG_TRACE=1 $prog Sherlock /dev/shm/xsg_c1.txt 2>&1 >/dev/shm/xsg_c1.out | grep -v amdgpu.ids
    done
  done
  echo "== grep"; /usr/bin/time -f "wall %e s user %U s sys %S s" grep Sherlock /dev/shm/xsg_c1.txt > /dev/shm/xsg_c1.g 2>&1; tail -1 /dev/shm/xsg_c1.g
  rm -f /dev/shm/xsg_c1.*
} >> $O 2>&1
echo "== cli_clock" >> $O
timeout -k 10 600 python scripts/cli_clock.py --gib 10 --reps 3 > gpurun_out/cli_clock.jsonl 2>&1
echo "cli_clock rc=$?" >> $O
timeout -k 10 300 python scripts/config1_e2e.py > gpurun_out/config1_e2e.log 2>&1
echo "config1 rc=$?" >> $O
tail -5 gpurun_out/cli_clock.jsonl | cut -c1-1500
cat gpurun_out/config1_e2e.log | cut -c1-300
