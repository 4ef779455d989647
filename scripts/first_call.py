#!/usr/bin/env python3
"""What the FIRST xsg_count of a binding costs (the probe) on a 50 GiB shard: first process-wide call, a
second pattern on a fresh binding (code loaded, allocations warm: the probe alone), later calls.  XSG_HOT=0 pins the
filter (no probe) for the comparison."""
import argparse, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import bench
import torch, corpus, xsg
ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=50.0)
a = ap.parse_args()
args = argparse.Namespace(chunk_mib=16, templates=32, seed=0x5EED, lexicon=None)
blocks = bench.template_blocks(args, b"Sherlock")
tbytes = np.array([b.size for b in blocks], dtype=np.int64)
n = int(a.gib * 64)
plan = bench.chunk_plan(args, 0, n)
off, ln, cap = corpus.chunk_table(tbytes[plan])
t = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
dts = [torch.from_numpy(b).to("cuda:0") for b in blocks]
for c in range(n):
    o = int(off[c]); t[o:o + dts[int(plan[c])].numel()].copy_(dts[int(plan[c])])
torch.cuda.synchronize()
ctx = xsg.Context(0)
chunks = xsg.make_chunks(off, ln)
for pat in (b"Sherlock", b"Sherlocx", b"Watson said", b"Sherlock"):
    ctx.set_pattern(pat)
    t0 = time.perf_counter()
    sh = xsg.Shard(ctx, t.data_ptr(), cap, chunks)
    t1 = time.perf_counter()
    c = sh.count(xsg.COUNT_MATCHES)
    t2 = time.perf_counter()
    for _ in range(5):
        c = sh.count(xsg.COUNT_MATCHES)
    t3 = time.perf_counter()
    print({"pattern": pat.decode(), "hot_env": os.environ.get("XSG_HOT"), "create_ms": round((t1 - t0) * 1e3, 3),
           "first_count_ms": round((t2 - t1) * 1e3, 3), "later_ms": round((t3 - t2) / 5 * 1e3, 3),
           "kernel": sh.scan_kernel_name(xsg.COUNT_MATCHES)}, flush=True)
    sh.close()
