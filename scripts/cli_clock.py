#!/usr/bin/env python3
"""The reference's one published measurement, on this box: `time grep Sherlock FILE` against `time my_grep Sherlock FILE`
(README.md:44-62) -- wall time of a FRESH PROCESS each, output redirected to a file, process start-up included.

Programs (every one prints the matching lines; outputs are compared byte for byte):
  xsgrep     tools/build/xsgrep PATTERN FILE            (example/grep.cpp on this engine; 2 threads like grep.cpp:21)
  my_grep    tools/build/my_grep PATTERN FILE           (README.md:31-41 compiled unchanged: extern_search<lines>(p, f, false, 1))
  grep       GNU grep PATTERN FILE                      (the caller's locale, and LC_ALL=C)
  ref_T1/_Tq oracle/_ref/xsref_grep PATTERN FILE T      (the reference's compiled simd_search.cpp under the restated walk;
                                                         T = 1 and T = the cgroup's CPU quota)
Files: BASELINE config 1's shape (100 000 000 bytes, six newline-aligned 16 MiB chunks) and 10 GiB, synthetic corpus
(x-search_amd/corpus.py), on tmpfs.  bench.py imports cli_block() for its `cli` object (N = 1, before the bench
process itself touches the GPU); run directly it prints the block as JSON lines.
"""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def make_file(path: str, nbytes: int, pattern: bytes, seed: int = 0x5EED, templates: int = 8):
    """nbytes of corpus text: seeded template chunks of 16 MiB(+) in a seeded order, cut at nbytes, last byte '\\n'"""
    import bench
    import corpus
    args = argparse.Namespace(chunk_mib=16, templates=min(templates, max(1, -(-nbytes // (16 << 20)))), seed=seed)
    blocks = bench.template_blocks(args, pattern)
    left, k = nbytes, 0
    with open(path, "wb") as f:
        while left > 0:
            b = blocks[corpus._mix(seed ^ 0xC11, k) % len(blocks)]
            k += 1
            if b.size >= left:
                b = b[:left].copy()
                b[-1] = 10
            f.write(memoryview(b))
            left -= b.size
    assert os.path.getsize(path) == nbytes


def run_timed(cmd, out_path, reps, env=None):
    """wall seconds of `reps` fresh processes of cmd with stdout > out_path; returns (times, sha256 of the output)"""
    ts = []
    for _ in range(reps):
        with open(out_path, "wb") as f:
            t0 = time.perf_counter()
            r = subprocess.run(cmd, stdout=f, stderr=subprocess.PIPE, env=env)
            ts.append(time.perf_counter() - t0)
        if r.returncode != 0:
            raise RuntimeError(f"{cmd}: rc {r.returncode}: {r.stderr.decode(errors='replace')[-400:]}")
    h = hashlib.sha256()
    with open(out_path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return ts, h.hexdigest(), os.path.getsize(out_path)


def cpu_quota() -> int:
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    return len(os.sched_getaffinity(0))


def cli_block(sizes=(("config1_100MB", 100_000_000, 5), ("10GiB", 10 << 30, 3)), pattern=b"Sherlock", tmpdir="/dev/shm",
              log=None):
    xsgrep = ROOT / "tools" / "build" / "xsgrep"
    my_grep = ROOT / "tools" / "build" / "my_grep"
    ref = ROOT / "oracle" / "_ref" / "xsref_grep"
    tq = cpu_quota()
    pat = pattern.decode()
    out = {"pattern": pat, "cpu_quota": tq, "what": "wall seconds of a fresh process, stdout to a tmpfs file (README.md:44-62)",
           "files": {}}
    env_c = dict(os.environ, LC_ALL="C")
    for name, nbytes, reps in sizes:
        path = f"{tmpdir}/xsg_cli_{os.getpid()}_{name}.txt"
        res = f"{tmpdir}/xsg_cli_{os.getpid()}_{name}.out"
        try:
            t0 = time.perf_counter()
            make_file(path, nbytes, pattern)
            rec = {"bytes": nbytes, "make_file_s": round(time.perf_counter() - t0, 2), "programs": {}}
            progs = [("xsgrep", [str(xsgrep), pat, path], None), ("my_grep", [str(my_grep), pat, path], None),
                     ("grep", ["grep", pat, path], None), ("grep_LC_ALL_C", ["grep", pat, path], env_c)]
            if ref.exists():
                progs += [("ref_T1", [str(ref), pat, path, "1"], None), (f"ref_T{tq}", [str(ref), pat, path, str(tq)], None)]
            digests = {}
            for pname, cmd, env in progs:
                ts, dg, nout = run_timed(cmd, res, reps, env)
                digests[pname] = dg
                rec["programs"][pname] = {"s_min": round(min(ts), 4), "s_median": round(statistics.median(ts), 4),
                                          "s_first": round(ts[0], 4), "gib_s": round(nbytes / min(ts) / 2**30, 2)}
                rec["output_bytes"] = nout
                if log:
                    log(json.dumps({"file": name, "program": pname, **rec["programs"][pname]}))
            rec["outputs_identical"] = len(set(digests.values())) == 1
            if not rec["outputs_identical"]:
                rec["digests"] = digests
            g = rec["programs"]["grep"]["s_min"]
            rec["speedup_vs_grep"] = {k: round(g / v["s_min"], 2) for k, v in rec["programs"].items() if k != "grep"}
            out["files"][name] = rec
        finally:
            for p in (path, res):
                if os.path.exists(p):
                    os.unlink(p)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=10.0)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    sizes = [("config1_100MB", 100_000_000, max(a.reps, 5))]
    if a.gib > 0:
        sizes.append((f"{a.gib:g}GiB", int(a.gib * 2**30), a.reps))
    blk = cli_block(tuple(sizes), log=lambda s: print(s, flush=True))
    print(json.dumps(blk), flush=True)
