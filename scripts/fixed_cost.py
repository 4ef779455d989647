#!/usr/bin/env python3
"""Where does a count step's time go besides the bulk scan?  For sub-shards of a
device-resident corpus (6.25 / 10 / 50 GiB = config 3's per-GPU share at 8 / - / 1
GPUs): the bulk kernel alone (HIP events), pipelined xsg_count_async steps
(what bench.py times), synchronous xsg_count calls, and the host cost of
enqueueing one step.  JSON lines on stdout and in gpurun_out/fixed_cost.jsonl."""
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
    ap.add_argument("--gib", type=float, default=50.0)
    ap.add_argument("--sizes", type=str, default="6.25,10,50")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--pattern", type=str, default="Sherlock")
    ap.add_argument("--modes", type=str, default="count")
    ap.add_argument("--out", type=str, default=str(ROOT / "gpurun_out" / "fixed_cost.jsonl"))
    a = ap.parse_args()
    import torch
    import corpus
    import xsg
    args = argparse.Namespace(chunk_mib=16, templates=32, seed=0x5EED)
    pattern = a.pattern.encode()
    blocks = bench.template_blocks(args, pattern)
    tbytes = np.array([b.size for b in blocks], dtype=np.int64)
    nchunks = int(round(a.gib * 2**30 / (16 << 20)))
    plan = bench.chunk_plan(args, 0, nchunks)
    off, ln, cap = corpus.chunk_table(tbytes[plan])
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
    ctx.set_pattern(pattern)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    counters = torch.zeros((2, xsg.NUM_COUNTERS), dtype=torch.int64, device=dev)
    out = open(a.out, "w")

    def emit(**kw):
        out.write(json.dumps(kw) + "\n")
        out.flush()
        print(json.dumps(kw), flush=True)

    modes = {"count": xsg.COUNT_MATCHES, "count_lines": xsg.COUNT_LINES, "count+nl": xsg.COUNT_MATCHES | xsg.WITH_NEWLINES}
    for gib in [float(x) for x in a.sizes.split(",")]:
        n = min(nchunks, int(round(gib * 2**30 / (16 << 20))))
        sub = chunks[:n]
        nbytes = int(sub["length"].sum())
        sh = xsg.Shard(ctx, shard_t.data_ptr(), cap, sub)
        for mname in a.modes.split(","):
            mode = modes[mname]
            kms = sh.time_scan_kernel(mode, 10)
            # pipelined async steps, device time between two events on the stream
            for i in range(3):
                sh.count_async(mode, stream.cuda_stream, counters[i & 1].data_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for i in range(a.steps):
                sh.count_async(mode, stream.cuda_stream, counters[i & 1].data_ptr())
            e1.record(stream)
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_wall = time.perf_counter() - t0
            step_ms = e0.elapsed_time(e1) / a.steps
            # synchronous calls
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                sh.count(mode)
                ts.append(time.perf_counter() - t0)
            sync_ms = float(np.median(ts)) * 1e3
            ideal_ms = nbytes / 7.3e12 * 1e3
            emit(gib=gib, chunks=n, bytes=nbytes, mode=mname, pattern=a.pattern, kernel_ms=round(kms, 4),
                 kernel_tbs=round(nbytes / kms / 1e9, 3), async_step_ms=round(step_ms, 4),
                 async_wall_ms=round(t_wall / a.steps * 1e3, 4), enqueue_us=round(t_enq / a.steps * 1e6, 1),
                 sync_call_ms=round(sync_ms, 4), ideal_ms_at_7p3=round(ideal_ms, 4),
                 async_over_ideal=round(step_ms / ideal_ms, 4), sync_over_ideal=round(sync_ms / ideal_ms, 4))
        sh.close()
    out.close()


if __name__ == "__main__":
    main()
