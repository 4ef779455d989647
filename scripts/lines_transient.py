#!/usr/bin/env python3
"""count_lines of a sparse needle on a 10 GiB shard, call after call: does the scan kernel of the LINES variant run at its
back-to-back rate when a finish kernel and a host round trip sit between the launches?  (bench.py's configs leg measured
1.75-1.80 ms where the kernel alone takes 1.45.)  Run under rocprofv3 --kernel-trace for the per-launch durations."""
import argparse, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import bench
import torch, corpus, xsg
ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=10.0)
ap.add_argument("--buffer-gib", type=float, default=10.0, help="size of the buffer the shard is the head of")
ap.add_argument("--list-first", action="store_true", help="an xs::line_indices search before the series (as bench.py's configs leg does)")
a = ap.parse_args()
args = argparse.Namespace(chunk_mib=16, templates=32, seed=0x5EED, lexicon=None)
blocks = bench.template_blocks(args, b"Sherlock")
tbytes = np.array([b.size for b in blocks], dtype=np.int64)
nbuf = int(a.buffer_gib * 64)
plan = bench.chunk_plan(args, 0, nbuf)
off, ln, cap = corpus.chunk_table(tbytes[plan])
t = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
dts = [torch.from_numpy(b).to("cuda:0") for b in blocks]
for c in range(nbuf):
    o = int(off[c]); t[o:o + dts[int(plan[c])].numel()].copy_(dts[int(plan[c])])
torch.cuda.synchronize()
n = int(a.gib * 64)
ctx = xsg.Context(0)
ctx.set_pattern(b"Sherlock")
sh = xsg.Shard(ctx, t.data_ptr(), cap, xsg.make_chunks(off[:n], ln[:n]))
def series(mode, k):
    out = []
    for _ in range(k):
        t0 = time.perf_counter(); sh.count(mode); out.append(round((time.perf_counter() - t0) * 1e3, 3))
    return out
if a.list_first:
    t0 = time.perf_counter(); k = sh.search_u64_view(xsg.LINE_INDICES).size
    print("line_indices first: %d results, %.3f ms" % (k, (time.perf_counter() - t0) * 1e3), flush=True)
print("count      ", series(xsg.COUNT_MATCHES, 12), flush=True)
print("count_lines", series(xsg.COUNT_LINES, 30), flush=True)
print("count      ", series(xsg.COUNT_MATCHES, 12), flush=True)
print("kernel only, 20 back to back: count_lines %.4f ms, count %.4f ms" % (sh.time_scan_kernel(xsg.COUNT_LINES, 20), sh.time_scan_kernel(xsg.COUNT_MATCHES, 20)), flush=True)
print("count_lines", series(xsg.COUNT_LINES, 12), flush=True)
