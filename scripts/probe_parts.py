#!/usr/bin/env python3
"""k_read_probe: which part of k_scan costs what against the pure nt read?"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "x-search_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import bench  # noqa: E402
import torch  # noqa: E402
import corpus  # noqa: E402
import xsg  # noqa: E402

args = argparse.Namespace(chunk_mib=16, templates=8, seed=0x5EED)
blocks = bench.template_blocks(args, b"Sherlock")
n = 3200
plan = bench.chunk_plan(args, 0, n)
tbytes = np.array([b.size for b in blocks], dtype=np.int64)
off, ln, cap = corpus.chunk_table(tbytes[plan])
dev = torch.device("cuda", 0)
t = torch.empty(cap, dtype=torch.uint8, device=dev)
dts = [torch.from_numpy(b).to(dev) for b in blocks]
for c in range(n):
    o = int(off[c])
    t[o:o + dts[int(plan[c])].numel()].copy_(dts[int(plan[c])])
torch.cuda.synchronize()
ctx = xsg.Context(0)
ctx.set_pattern(b"Sherlock")
sh = xsg.Shard(ctx, t.data_ptr(), cap, xsg.make_chunks(off, ln))
lib = xsg.load()
lib.xsg_diag_read_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
total = int(ln.sum())
names = {0: "read only", 1: "+prologue", 2: "+epilogue", 3: "+pro+epi", 4: "+alu", 5: "+pro+alu", 6: "+epi+alu", 7: "+all"}
for rnd in range(3):
    for parts in range(8):
        ms = C.c_float(0)
        assert lib.xsg_diag_read_probe(sh.h, parts, 5, C.byref(ms)) == 0, lib.xsg_last_error()
        print(json.dumps({"round": rnd, "parts": names[parts], "ms": round(ms.value, 3), "gbs": round(total / ms.value / 1e6, 1)}), flush=True)
    ms = sh.time_scan_kernel(xsg.COUNT_MATCHES, 5)
    print(json.dumps({"round": rnd, "parts": "k_scan", "ms": round(ms, 3), "gbs": round(total / ms / 1e6, 1)}), flush=True)
