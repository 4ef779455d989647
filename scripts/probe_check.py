#!/usr/bin/env python3
"""How stable is the library's hot-filter probe?  Fresh bindings of one 50 GiB shard, the probe's own timings (XSG_PROBE_LOG)."""
import argparse, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import bench
os.environ["XSG_PROBE_LOG"] = "1"
import torch, corpus, xsg
args = argparse.Namespace(chunk_mib=16, templates=32, seed=0x5EED, lexicon=None)
blocks = bench.template_blocks(args, b"Sherlock")
tbytes = np.array([b.size for b in blocks], dtype=np.int64)
n = 3200
plan = bench.chunk_plan(args, 0, n)
off, ln, cap = corpus.chunk_table(tbytes[plan])
t = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
dts = [torch.from_numpy(b).to("cuda:0") for b in blocks]
for c in range(n):
    o = int(off[c]); t[o:o + dts[int(plan[c])].numel()].copy_(dts[int(plan[c])])
torch.cuda.synchronize()
ctx = xsg.Context(0)
for pat in (b"Sherlock", b"Sherlock Holmes", b"detective street"):
    ctx.set_pattern(pat)
    for k in range(8):
        sh = xsg.Shard(ctx, t.data_ptr(), cap, xsg.make_chunks(off, ln))
        c = sh.count(xsg.COUNT_MATCHES)
        print(pat, k, sh.scan_kernel_name(xsg.COUNT_MATCHES), flush=True)
        sh.close()
