#!/usr/bin/env python3
"""A/B of the in-register class verification (XSG_CLS_INREG=0|1, read when the pattern is set) on whole xsg_count calls:
class sequences proper and the prefilter route of the automaton family (whose candidate scan is a class sequence)."""
import os
os.environ.setdefault("XSG_TEST_HOOKS", "1")  # this script switches XSG_* toggles between searches (read once per process otherwise)
import argparse, json, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=20.0)
a = ap.parse_args()
import torch, corpus, xsg
args = argparse.Namespace(chunk_mib=16, templates=32, seed=0x5EED, lexicon=None)
blocks = bench.template_blocks(args, b"Sherlock")
tbytes = np.array([b.size for b in blocks], dtype=np.int64)
n = int(round(a.gib * 2**30 / (16 << 20)))
plan = bench.chunk_plan(args, 0, n)
off, ln, cap = corpus.chunk_table(tbytes[plan])
nbytes = int(ln.sum())
t = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
dts = [torch.from_numpy(b).to("cuda:0") for b in blocks]
for c in range(n):
    o = int(off[c]); t[o:o + dts[int(plan[c])].numel()].copy_(dts[int(plan[c])])
torch.cuda.synchronize()
ctx = xsg.Context(0)
for expr in ("She[r ]lock", "[Tt]he [a-z]{3} ", "Sherlock|Holmes", "Sher.*mes", "colou?r", "(the|The) +\\w{5,}"):
    for inreg in ("1", "0"):
        os.environ["XSG_CLS_INREG"] = inreg
        sh = xsg.Shard(ctx, t.data_ptr(), cap, xsg.make_chunks(off, ln))
        ctx.set_pattern(expr.encode(), xsg.FLAG_REGEX)
        got = int(sh.count(xsg.COUNT_MATCHES)[0])
        t0 = time.perf_counter()
        for _ in range(3):
            got = int(sh.count(xsg.COUNT_MATCHES)[0])
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(json.dumps({"expr": expr, "inreg": inreg, "matches": got, "ms_per_call": round(ms, 3), "tb_s": round(nbytes / ms / 1e9, 3),
                          "kernel": sh.scan_kernel_name(xsg.COUNT_MATCHES)[:90]}), flush=True)
        sh.close()
