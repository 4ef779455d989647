#!/usr/bin/env python3
"""XSG_TUNE sweep: does staggering / holding the waves' load bursts change k_scan's rate?"""
import argparse, json, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import bench, torch, corpus, xsg  # noqa: E402
args = argparse.Namespace(chunk_mib=16, templates=32, seed=0x5EED)
blocks = bench.template_blocks(args, b"Sherlock")
n = 3200
plan = bench.chunk_plan(args, 0, n)
tbytes = np.array([b.size for b in blocks], dtype=np.int64)
off, ln, cap = corpus.chunk_table(tbytes[plan])
t = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
dts = [torch.from_numpy(b).to("cuda:0") for b in blocks]
for c in range(n):
    o = int(off[c]); t[o:o + dts[int(plan[c])].numel()].copy_(dts[int(plan[c])])
torch.cuda.synchronize(); del dts
chunks = xsg.make_chunks(off, ln); total = int(ln.sum())
tunes = [0, 6, 12]
cases = [("e", xsg.COUNT_MATCHES, "count"), ("the", xsg.COUNT_MATCHES, "count"), ("~", xsg.COUNT_MATCHES, "count"),
         ("q~", xsg.COUNT_MATCHES, "count")]
shards = {}
for tu in tunes:
    os.environ["XSG_TUNE"] = hex(tu)
    ctx = xsg.Context(0)
    shards[tu] = (ctx, xsg.Shard(ctx, t.data_ptr(), cap, chunks))
for rnd in range(2):
    for pat, mode, name in cases:
        for tu in tunes:
            ctx, sh = shards[tu]
            ctx.set_pattern(pat.encode(), xsg.FLAG_IGNORE_CASE if name == "icase" else 0)
            ms = sh.time_scan_kernel(mode, 5)
            print(json.dumps({"round": rnd, "pattern": pat, "mode": name, "tune": tu, "ms": round(ms, 3),
                              "gbs": round(total / ms / 1e6, 1)}), flush=True)
