#!/usr/bin/env python3
"""Wall time of the list tags on device-resident shards (BASELINE configs 2 and 4):
the whole xsg_search call (bulk scan + ranks + emission + post-processing + D2H of
the results), median of several runs."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "x-search_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
import xsg  # noqa: E402
from test_gpu_fullsize import build_shard  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=10.0)
ap.add_argument("--reps", type=int, default=7)
a = ap.parse_args()
t, blocks, plan, chunks, goffs, cap = build_shard(a.gib)
nbytes = int(chunks["length"].sum())
ctx = xsg.Context(0)
sh = xsg.Shard(ctx, t.data_ptr(), cap, chunks)
for pat in (b"Sherlock", b"She", b"the"):
    ctx.set_pattern(pat)
    for name, fn in (("count", lambda: sh.count(xsg.COUNT_MATCHES)), ("count_lines", lambda: sh.count(xsg.COUNT_LINES)),
                     ("match_byte_offsets", lambda: sh.search_u64(xsg.MATCH_BYTE_OFFSETS)),
                     ("line_byte_offsets", lambda: sh.search_u64(xsg.LINE_BYTE_OFFSETS)),
                     ("line_indices", lambda: sh.search_u64(xsg.LINE_INDICES)), ("lines", lambda: sh.search_lines())):
        if pat == b"the" and name == "lines":
            continue  # hundreds of MB of Python string objects: not a kernel measurement
        ts = []
        n = None
        for _ in range(a.reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
            n = int(r[0]) if name == "count" else int(r[1]) if name == "count_lines" else len(r[0]) if name == "lines" else len(r)
        ms = float(np.median(ts)) * 1e3
        print(json.dumps({"gib": a.gib, "pattern": pat.decode(), "tag": name, "results": n, "ms_median": round(ms, 3),
                          "gib_per_s": round(nbytes / 2**30 / (ms / 1e3), 1)}), flush=True)
