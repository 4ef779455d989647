#!/usr/bin/env python3
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
order = [("0", "0x0"), ("12", "0xc"), ("auto", None), ("12b", "0xc"), ("0b", "0x0"), ("autob", None)]
shards = []
for name, env in order:
    if env is None:
        os.environ.pop("XSG_TUNE", None)
    else:
        os.environ["XSG_TUNE"] = env
    ctx = xsg.Context(0); ctx.set_pattern(b"Sherlock")
    shards.append((name, ctx, xsg.Shard(ctx, t.data_ptr(), cap, chunks)))
for rnd in range(3):
    for name, ctx, sh in shards:
        ms = sh.time_scan_kernel(xsg.COUNT_MATCHES, 8)
        print(json.dumps({"round": rnd, "tune": name, "ms": round(ms, 3), "gbs": round(total / ms / 1e6, 1)}), flush=True)
