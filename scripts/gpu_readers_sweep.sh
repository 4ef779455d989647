#!/bin/bash
# how many reader threads does a fresh xsgrep want on a 10 GiB page-cache file?  (XSG_MIN_READERS; the default is clamp(cpus / 2, 2, 8))
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import os, subprocess, sys, time
sys.path.insert(0, "scripts")
import cli_clock
f = "/dev/shm/xsg_rs.txt"
cli_clock.make_file(f, 10 << 30, b"Sherlock")
for r in (8, 10, 12, 14, 16):
    env = dict(os.environ, XSG_MIN_READERS=str(r))
    for args, label in ((["-c"], "count"), ([], "lines")):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            subprocess.run(["tools/build/xsgrep"] + args + ["Sherlock", f], env=env, stdout=subprocess.DEVNULL, check=True)
            ts.append(time.perf_counter() - t0)
        print(f"readers={r} {label}: min {min(ts):.3f} s  all {[round(t, 3) for t in ts]}", flush=True)
os.unlink(f)
PY
