#!/usr/bin/env python3
"""Whole-call time of every xs:: tag on a device-resident shard (BASELINE configs 2 and 4), measured at the C ABI:
xsg_count for the count tags; xsg_search + the result in host memory for the list tags (xsg_result_u64_view /
xsg_result_lines_view: pointers into the shard's pinned buffers).  No Python object is built per result.  Median of --reps calls after one warm call."""
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


def timed_calls(lib, sh, reps, dense_ok=True, spread=None):
    """-> {tag: (results, median ms)} for the shard's current pattern; spread (a dict) receives {tag: (min ms, max ms)}"""
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

    lp, op, dp = _u64p(), _u64p(), C.c_char_p()

    def lines():  # lengths, offsets and packed bytes in the shard's pinned buffers (xsg_result_lines_view)
        xsg._check(lib.xsg_search(sh.h, xsg.LINES, C.byref(n)))
        xsg._check(lib.xsg_result_lines_view(sh.h, C.byref(lp), C.cast(C.byref(dp), C.POINTER(C.c_char_p)), C.byref(op), C.byref(nl), C.byref(nb)))
        return int(nl.value)

    calls = [("count", lambda: count(xsg.COUNT_MATCHES)), ("count_lines", lambda: count(xsg.COUNT_LINES)),
             ("match_byte_offsets", lambda: u64(xsg.MATCH_BYTE_OFFSETS)), ("line_byte_offsets", lambda: u64(xsg.LINE_BYTE_OFFSETS)),
             ("line_indices", lambda: u64(xsg.LINE_INDICES)), ("lines", lines)]
    for name, fn in calls:
        res = fn()  # warm (buffers, probes, the cached newline counts)
        for _ in range(3):  # ... and the clocks: the first launches after a quiet spell run up to 30 % slower (GPU timestamps)
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            res = fn()
            ts.append(time.perf_counter() - t0)
        out[name] = (res, float(np.median(ts)) * 1e3)
        if spread is not None:
            spread[name] = (min(ts) * 1e3, max(ts) * 1e3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=10.0)
    ap.add_argument("--reps", type=int, default=15)
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
