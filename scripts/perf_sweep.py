#!/usr/bin/env python3
"""Kernel-level A/B on one device-resident shard (one process, interleaved
rounds): tile geometry, count modes, pattern kinds, and the read-only ceiling.
Writes JSON lines to gpurun_out/perf_sweep.jsonl."""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "x-search_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=50.0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--tiles", type=str, default="16,32")
    ap.add_argument("--out", type=str, default=str(ROOT / "gpurun_out" / "perf_sweep.jsonl"))
    a = ap.parse_args()
    import torch
    import corpus
    import xsg
    args = argparse.Namespace(chunk_mib=16, templates=32, seed=0x5EED)
    pattern = b"Sherlock"
    blocks = bench.template_blocks(args, pattern)
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
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    out = open(a.out, "w")

    def emit(**kw):
        out.write(json.dumps(kw) + "\n")
        out.flush()
        print(json.dumps(kw), flush=True)

    shards = {}
    for tk in [int(x) for x in a.tiles.split(",")]:
        os.environ["XSG_TILE_KIB"] = str(tk)
        ctx = xsg.Context(0)
        ctx.set_pattern(pattern)
        shards[tk] = (ctx, xsg.Shard(ctx, shard_t.data_ptr(), cap, chunks))
    cases = [("Sherlock", xsg.COUNT_MATCHES, "count"), ("Sherlock", xsg.COUNT_MATCHES | xsg.WITH_NEWLINES, "count+nl"),
             ("Sherlock", xsg.COUNT_LINES, "count_lines"), ("e", xsg.COUNT_MATCHES, "count"), ("the", xsg.COUNT_MATCHES, "count"),
             ("that", xsg.COUNT_MATCHES, "count"), ("Sherl", xsg.COUNT_MATCHES, "count"),
             ("detective street", xsg.COUNT_MATCHES, "count"), ("Sherlock Holmes", xsg.COUNT_MATCHES, "count"),
             ("the detective", xsg.COUNT_MATCHES, "count"), ("information", xsg.COUNT_MATCHES, "count"),
             ("e", xsg.COUNT_LINES, "count_lines")]
    for r in range(a.rounds):
        for tk, (ctx, sh) in shards.items():
            import xsg_diag
            ms, nb = xsg_diag.read(shard_t.data_ptr(), cap - 65536, sink.data_ptr(), tile_bytes=tk * 1024, variant=1, iters=a.iters)
            emit(round=r, tile_kib=tk, what="read_ceiling_nt", ms=ms, gbs=nb / ms / 1e6, bytes=nb)
            for pat, mode, name in cases:
                if r > 0 and pat != "Sherlock":
                    continue
                ctx.set_pattern(pat.encode())
                ms = sh.time_scan_kernel(mode, a.iters)
                emit(round=r, tile_kib=tk, what="k_scan", pattern=pat, mode=name, ms=ms, gbs=shard_bytes / ms / 1e6)
            if r == 0:
                for pat in ("Sherlock", "Sherl", "that", "Sherlock Holmes", "the"):
                    ctx.set_pattern(pat.encode(), xsg.FLAG_IGNORE_CASE)
                    ms = sh.time_scan_kernel(xsg.COUNT_MATCHES, a.iters)
                    emit(round=r, tile_kib=tk, what="k_scan", pattern=pat, mode="icase count", ms=ms,
                         gbs=shard_bytes / ms / 1e6)
                for expr in ("She[r ]lock", "[Ss]herlock", "[0-9]{4}-[0-9]{2}", "[Tt]he [a-z]{3} "):
                    ctx.set_pattern(expr.encode(), xsg.FLAG_REGEX)
                    ms = sh.time_scan_kernel(xsg.COUNT_MATCHES, a.iters)
                    emit(round=r, tile_kib=tk, what="k_scan", pattern=expr, mode="regex count", ms=ms,
                         gbs=shard_bytes / ms / 1e6)
    out.close()


if __name__ == "__main__":
    main()
