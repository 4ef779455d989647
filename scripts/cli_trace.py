#!/usr/bin/env python3
"""Where a fresh process's search goes: runs a command with XSG_TRACE=1 and brackets the library's marks with the same
clock (CLOCK_MONOTONIC) taken just before the process is started and just after it has exited -- so the time before
the library is loaded (exec, dynamic linking) and behind its last mark (result output, HIP teardown, exit) shows too.
usage: cli_trace.py [--reps N] -- command args..."""
import os
import re
import subprocess
import sys
import time

args = sys.argv[1:]
reps = 1
if args[:1] == ["--reps"]:
    reps = int(args[1])
    args = args[2:]
if args[:1] == ["--"]:
    args = args[1:]
for rep in range(reps):
    env = dict(os.environ, XSG_TRACE="1")
    t0 = time.monotonic()
    p = subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
    t1 = time.monotonic()
    marks = []
    for line in p.stderr.decode(errors="replace").splitlines():
        m = re.match(r"\[xsg \+\s*([\d.]+) ms \| ([\d.]+)\] (.*)", line)
        if m:
            marks.append((float(m.group(2)), m.group(3)))
    print(f"== {' '.join(args)}   (run {rep + 1}, rc {p.returncode}): {1e3 * (t1 - t0):.1f} ms from spawn to exit")
    prev = t0
    for t, what in marks:
        print(f"  +{1e3 * (t - t0):9.3f} ms  (+{1e3 * (t - prev):8.3f})  {what}")
        prev = t
    print(f"  +{1e3 * (t1 - t0):9.3f} ms  (+{1e3 * (t1 - prev):8.3f})  process has exited")
