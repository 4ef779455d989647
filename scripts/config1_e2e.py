#!/usr/bin/env python3
"""BASELINE config 1's shape end to end on the GPU: xs::count 'Sherlock' on a 100 MB file (6 newline-aligned 16 MiB
chunks, tmpfs), wall time of the whole job (xsg_job_start -> join) once the process is warm, for a few thread
settings; the CPU reference on the same bytes beside it (one thread, the reference's compiled simd_search.cpp)."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import argparse  # noqa: E402
import bench  # noqa: E402
import xsg  # noqa: E402
from xs_oracle import Oracle, Reference  # noqa: E402

args = argparse.Namespace(chunk_mib=16, templates=6, seed=0x5EED)
blocks = bench.template_blocks(args, b"Sherlock")
blocks[-1] = blocks[-1][:100_000_000 - sum(b.size for b in blocks[:-1])].copy()
blocks[-1][-1] = 10
data = np.concatenate(blocks)
assert data.size == 100_000_000
path = f"/dev/shm/xsg_config1_{os.getpid()}.txt"
data.tofile(path)
orc = Oracle()
if Reference.available():
    orc.use_reference_primitives(Reference())
want = sum(orc.count(b, b"Sherlock", False) for b in blocks)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    got = sum(orc.count(b, b"Sherlock", False) for b in blocks)
    ts.append(time.perf_counter() - t0)
print(json.dumps({"what": "CPU reference, 1 thread, 6 chunks in RAM", "ms": round(min(ts) * 1e3, 2),
                  "gib_s": round(data.size / min(ts) / 2**30, 2), "count": got}), flush=True)
try:
    for th, rd, chunk in ((1, 1, 16 << 20), (1, 2, 16 << 20), (2, 2, 16 << 20), (2, 4, 16 << 20), (2, 6, 16 << 20), (2, 6, 4 << 20), (3, 8, 4 << 20),
                          (4, 8, 2 << 20)):
        ms = []
        for rep in range(8):
            t0 = time.perf_counter()
            j = xsg.Job(b"Sherlock", path, xsg.COUNT_MATCHES, num_threads=th, num_max_readers=rd, chunk_bytes=chunk)
            r = j.result()
            ms.append((time.perf_counter() - t0) * 1e3)
            j.close()
            assert r == want, (r, want)
        print(json.dumps({"workers": th, "readers": rd, "chunk_mib": chunk >> 20, "first_ms": round(ms[0], 2),
                          "warm_ms_min": round(min(ms[2:]), 2), "warm_ms_median": round(float(np.median(ms[2:])), 2),
                          "gib_s_warm": round(data.size / min(ms[2:]) * 1e3 / 2**30, 2)}), flush=True)
finally:
    os.unlink(path)
