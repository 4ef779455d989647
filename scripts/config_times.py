#!/usr/bin/env python3
"""Whole-call time of every xs:: tag on a device-resident shard (BASELINE configs 2 and 4), measured at the C ABI:
xsg_count for the count tags; xsg_search + the result in host memory for the list tags (xsg_result_u64_view -- a
pointer into the shard's pinned buffer -- for the uint64 tags; xsg_result_lines into preallocated host arrays for
xs::lines).  No Python object is built per result.  Median of --reps calls after one warm call."""
import argparse
import ctypes as C
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

_u64p = C.POINTER(C.c_uint64)


def timed_calls(lib, sh, reps, dense_ok=True):
    """-> {tag: (results, median ms)} for the shard's current pattern"""
    out = {}
    ctr = np.zeros(xsg.NUM_COUNTERS, dtype=np.uint64)
    n = C.c_uint64(0)
    ptr, cnt = _u64p(), C.c_uint64(0)
    nl, nb = C.c_uint64(0), C.c_uint64(0)
    bufs = {}

    def count(mode):
        xsg._check(lib.xsg_count(sh.h, mode, ctr.ctypes.data_as(_u64p)))
        return int(ctr[xsg.CTR_MATCHES if mode == xsg.COUNT_MATCHES else xsg.CTR_LINES])

    def u64(mode):
        xsg._check(lib.xsg_search(sh.h, mode, C.byref(n)))
        xsg._check(lib.xsg_result_u64_view(sh.h, C.byref(ptr), C.byref(cnt)))
        return int(cnt.value)

    def lines():
        xsg._check(lib.xsg_search(sh.h, xsg.LINES, C.byref(n)))
        xsg._check(lib.xsg_result_lines_size(sh.h, C.byref(nl), C.byref(nb)))
        if "lens" not in bufs or bufs["lens"].size < nl.value or bufs["bytes"].size < nb.value:
            bufs["lens"] = np.empty(max(nl.value, 1), dtype=np.uint64)
            bufs["offs"] = np.empty(max(nl.value, 1), dtype=np.uint64)
            bufs["bytes"] = np.empty(max(nb.value, 1), dtype=np.uint8)
        xsg._check(lib.xsg_result_lines(sh.h, bufs["lens"].ctypes.data_as(_u64p), bufs["bytes"].ctypes.data, nb.value,
                                        bufs["offs"].ctypes.data_as(_u64p)))
        return int(nl.value)

    calls = [("count", lambda: count(xsg.COUNT_MATCHES)), ("count_lines", lambda: count(xsg.COUNT_LINES)),
             ("match_byte_offsets", lambda: u64(xsg.MATCH_BYTE_OFFSETS)), ("line_byte_offsets", lambda: u64(xsg.LINE_BYTE_OFFSETS)),
             ("line_indices", lambda: u64(xsg.LINE_INDICES)), ("lines", lines)]
    for name, fn in calls:
        res = fn()  # warm (buffers, probes, the cached newline counts)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            res = fn()
            ts.append(time.perf_counter() - t0)
        out[name] = (res, float(np.median(ts)) * 1e3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=10.0)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--patterns", type=str, default="Sherlock,She,the")
    a = ap.parse_args()
    from test_gpu_fullsize import build_shard
    t, blocks, plan, chunks, goffs, cap = build_shard(a.gib)
    nbytes = int(chunks["length"].sum())
    lib = xsg.load()
    ctx = xsg.Context(0)
    sh = xsg.Shard(ctx, t.data_ptr(), cap, chunks)
    for pat in a.patterns.split(","):
        ctx.set_pattern(pat.encode())
        r = timed_calls(lib, sh, a.reps)
        base = r["count"][1]
        for tag, (n, ms) in r.items():
            print(json.dumps({"gib": a.gib, "pattern": pat, "tag": tag, "results": n, "ms_median": round(ms, 3),
                              "gb_per_s": round(nbytes / ms / 1e6, 1), "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000, 3),
                              "vs_count": round(ms / base, 3)}), flush=True)


if __name__ == "__main__":
    main()
