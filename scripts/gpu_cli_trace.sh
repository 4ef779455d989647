#!/bin/bash
# XSG_TRACE marks of fresh xsgrep / my_grep processes on BASELINE config 1's file, and the mmap / hipHostRegister
# alternatives to the pread ring (scripts/microbench/mmap_h2d.hip).
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
{
  for f in /dev/shm/xsg_c1.txt /dev/shm/xsg_c2.txt; do
  for prog in "tools/build/xsgrep -c" "tools/build/xsgrep" "tools/build/my_grep"; do
    for i in 1 2; do
      echo "== XSG_TRACE=1 $prog Sherlock $f  (run $i)"
      time (XSG_TRACE=1 $prog Sherlock $f 2>&1 >/dev/shm/xsg_c1.out | grep -v amdgpu.ids)
    done
  done
  done
  echo "== grep"; time (grep Sherlock /dev/shm/xsg_c1.txt > /dev/shm/xsg_c1.g)
  rm -f /dev/shm/xsg_c1.* /dev/shm/xsg_c2.*
} >> $O 2>&1
if [ -x scripts/microbench/build/mmap_h2d ]; then
  timeout -k 10 300 scripts/microbench/build/mmap_h2d /dev/shm/xsg_mm.bin 2 > gpurun_out/mmap_h2d.txt 2>&1
  echo "mmap_h2d rc=$?" >> $O
fi
tail -60 $O
