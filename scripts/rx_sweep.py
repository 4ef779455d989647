#!/usr/bin/env python3
"""k_rx_scan (the automaton route of XSG_FLAG_REGEX) on a device-resident shard of the bench corpus: kernel time per
expression and mode, next to the literal and class-sequence kernels on the same shard.
Writes JSON lines to gpurun_out/rx_sweep.jsonl."""
import os
os.environ.setdefault("XSG_TEST_HOOKS", "1")  # this script switches XSG_* toggles between searches (read once per process otherwise)
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
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=8.0)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--out", type=str, default=str(ROOT / "gpurun_out" / "rx_sweep.jsonl"))
    ap.add_argument("--exprs", type=str, default="")
    ap.add_argument("--cases", type=str, default="rx", help="rx: the automaton route; class: class sequences in k_scan")
    a = ap.parse_args()
    import torch
    import corpus
    import xsg
    args = argparse.Namespace(chunk_mib=16, templates=32, seed=0x5EED)
    blocks = bench.template_blocks(args, b"Sherlock")
    tbytes = np.array([b.size for b in blocks], dtype=np.int64)
    nchunks = int(round(a.gib * 2**30 / (16 << 20)))
    plan = bench.chunk_plan(args, 0, nchunks)
    off, ln, cap = corpus.chunk_table(tbytes[plan])
    shard_bytes = int(ln.sum())
    dev = torch.device("cuda", 0)
    shard_t = torch.empty(cap, dtype=torch.uint8, device=dev)
    dts = [torch.from_numpy(b).to(dev) for b in blocks]
    for c in range(nchunks):
        o = int(off[c])
        shard_t[o:o + dts[int(plan[c])].numel()].copy_(dts[int(plan[c])])
    torch.cuda.synchronize()
    del dts
    chunks = xsg.make_chunks(off, ln)
    ctx = xsg.Context(0)
    ctx.set_pattern(b"Sherlock")
    sh = xsg.Shard(ctx, shard_t.data_ptr(), cap, chunks)
    out = open(a.out, "w")

    def emit(**kw):
        out.write(json.dumps(kw) + "\n")
        out.flush()
        print(json.dumps(kw), flush=True)

    exprs = [e for e in a.exprs.split("|||") if e] or [
        "Sherlock|Holmes", "Sher.*mes", "Sher.*?k", "colou?r", "lock(ed|s)?", "[A-Z][a-z]+ [A-Z][a-z]+", "\\w+ing",
        "[0-9]+", "(the|The) +\\w{5,}", "S[a-z]{3,9}k", "zzz+"]
    cases = [("Sherlock", 0), ("She[r ]lock", xsg.FLAG_REGEX)] + [(e, xsg.FLAG_REGEX) for e in exprs]
    if a.cases == "class":
        cases = [(e, xsg.FLAG_REGEX) for e in ("She[r ]lock", "[Ss]herlock", "[Tt]he [a-z]{3} ", "[0-9]{4}-[0-9]{2}",
                                               "Sherlock|She lock", "S.erlock")]
    for expr, flags in cases:
        ctx.set_pattern(expr.encode(), flags)
        for mode, name in ((xsg.COUNT_MATCHES, "count"), (xsg.COUNT_LINES, "count_lines")):
            sh.rebind(shard_t.data_ptr(), cap, chunks)  # a fresh binding: no probe result, no tile marks from the mode before
            ms = sh.time_scan_kernel(mode, a.iters)
            sh.rebind(shard_t.data_ptr(), cap, chunks)
            t0 = time.perf_counter()  # the first synchronous call: with whatever the library measures or builds once
            c = sh.count(mode)
            first_ms = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()  # the synchronous call, whatever route it takes (prefilter: candidates + automaton)
            for _ in range(a.iters):
                c = sh.count(mode)
            call_ms = (time.perf_counter() - t0) * 1e3 / a.iters
            emit(pattern=expr, mode=name, kernel=sh.scan_kernel_name(mode), ms=ms, gbs=shard_bytes / ms / 1e6,
                 result=int(c[xsg.CTR_MATCHES if mode == xsg.COUNT_MATCHES else xsg.CTR_LINES]), bytes=shard_bytes,
                 count_call_ms=call_ms, count_call_gbs=shard_bytes / call_ms / 1e6, first_call_ms=first_ms,
                 first_call_gbs=shard_bytes / first_ms / 1e6, kernel_after=sh.scan_kernel_name(mode))
    out.close()


if __name__ == "__main__":
    main()
