#!/usr/bin/env python3
"""bench.py -- xs::count on a device-resident plain-text shard, GiB/s scanned.

Metric (BASELINE.json): GiB/s scanned (+ matches/s) for xs::count (literal
'Sherlock') on >= 50 GiB of plain text at 1/2/4/8 MI355X.

One "step" = one pass of the hot path over the rank's whole shard: the bulk scan
kernel over every chunk, the finish kernel (per-tile sums + the reference's
end-of-chunk walk) and -- for N > 1 -- one RCCL all-reduce (sum) of the four
uint64 counters.  Inputs are resident in HBM when the timed region starts
(weak scaling: every rank owns --gib-per-gpu GiB, default 50).

Corpus: synthetic (no network).  T distinct '\\n'-terminated template chunks of
16 MiB (+ a few bytes, like the reference's .meta fixtures) are generated on the
host with x-search_amd/corpus.py and replicated into the shard in a seeded
pseudo-random order, so every byte of the shard is real text at a distinct HBM
address (the working set is >> the 256 MiB Infinity Cache) and the expected
count of the full shard is known exactly from the oracle's counts of the T
templates.  The result of EVERY timed step is checked against it.

Launch:  python bench.py --gpus 1            (single process)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "x-search_amd"))
sys.path.insert(0, str(ROOT / "oracle"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gib-per-gpu", type=float, default=50.0)
    ap.add_argument("--chunk-mib", type=int, default=16)
    ap.add_argument("--templates", type=int, default=32)
    ap.add_argument("--pattern", type=str, default="Sherlock")
    ap.add_argument("--seed", type=int, default=0x5EED)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-iters", type=int, default=10)
    return ap.parse_args()


def template_blocks(args, pattern: bytes):
    import corpus
    target = args.chunk_mib << 20
    blocks = []
    for i in range(args.templates):
        extra = 1 + (corpus._mix(args.seed, 1000 + i) % 61)  # fixtures: 16 MiB extended to just past the next '\n'
        blocks.append(corpus.text_block(args.seed, i, target + extra, needle=pattern))
    return blocks


def chunk_plan(args, rank: int, nchunks: int):
    """template id of every chunk of this rank's shard (seeded, rank-dependent)"""
    import corpus
    return np.array([corpus._mix(args.seed ^ 0xC0FFEE, (rank << 32) + c) % args.templates for c in range(nchunks)],
                    dtype=np.int64)


def cpu_baseline(args, blocks, pattern: bytes):
    """The reference's CPU path on this host's cores, on a bounded sample of the
    same workload: the template chunks, work-stealing threads like
    include/xsearch/Searcher.h:100-120."""
    from xs_oracle import Oracle, Reference
    import corpus
    orc = Oracle()
    kind = "port"
    ref = None
    if Reference.available():
        try:
            ref = Reference()
            orc.use_reference_primitives(ref)  # the timed work is the reference's own compiled simd_search.cpp
            kind = "reference"
        except Exception:
            ref = None
    off, ln, cap = corpus.chunk_table([b.size for b in blocks])
    buf = np.zeros(cap, dtype=np.uint8)
    for o, b in zip(off, blocks):
        buf[int(o):int(o) + b.size] = b
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    sample_bytes = int(ln.sum())
    out = {}
    want = None
    for label, nt in (("1", 1), ("all", cores)):
        budget = args.cpu_seconds / 2
        t_total, passes = 0.0, 0
        while t_total < budget and passes < 64:
            t0 = time.perf_counter()
            tot, _ = orc.count_chunks_mt(buf, off, ln, pattern, False, nt)
            t_total += time.perf_counter() - t0
            passes += 1
            want = tot if want is None else want
            assert tot == want
        out[label] = (sample_bytes * passes / t_total / 2**30, passes)
    orc.use_reference_primitives(None)
    return {
        "value": round(out["all"][0], 3),
        "unit": "GiB/s",
        "cores": cores,
        "kind": kind,
        "value_1thread": round(out["1"][0], 3),
        "sample": f"{len(blocks)} chunks x {args.chunk_mib} MiB = {sample_bytes / 2**20:.0f} MiB of the same corpus in RAM, "
                  f"{out['all'][1]} passes with {cores} threads / {out['1'][1]} passes with 1 thread "
                  f"(xs::count, chunk work-stealing)",
    }


def main():
    args = parse()
    import torch
    import xsg
    from xs_oracle import Oracle

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # Rehearsal hook (not used by the driver): XSG_BENCH_BACKEND=gloo runs the N>1 code path on a box with
    # fewer GPUs than ranks (all ranks scan on the visible GPUs, the collective goes through host memory).
    backend = os.environ.get("XSG_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # XSG_BENCH_FORCE_DIST=1 (rehearsal, not used by the driver): take the N>1 code path -- RCCL
    # process group, all_reduce per step, barriers -- with a single rank, on a one-GPU box.
    if world > 1 or os.environ.get("XSG_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)

    pattern = args.pattern.encode("latin-1")
    t_setup = time.perf_counter()

    # ---- corpus: templates on the host, replicated into the shard in HBM
    blocks = template_blocks(args, pattern)
    orc = Oracle()
    tcount = np.array([orc.count(b, pattern, False) for b in blocks], dtype=np.int64)
    tbytes = np.array([b.size for b in blocks], dtype=np.int64)
    nchunks = max(1, int(round(args.gib_per_gpu * 2**30 / (args.chunk_mib << 20))))
    plan = chunk_plan(args, rank, nchunks)
    import corpus
    off, ln, cap = corpus.chunk_table(tbytes[plan])
    shard_bytes = int(ln.sum())
    expected_local = int(tcount[plan].sum())

    shard_t = torch.empty(cap, dtype=torch.uint8, device=dev)
    dev_templates = [torch.from_numpy(b).to(dev) for b in blocks]
    for c in range(nchunks):
        o = int(off[c])
        t = dev_templates[int(plan[c])]
        shard_t[o:o + t.numel()].copy_(t)
    torch.cuda.synchronize()
    del dev_templates

    goffs = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint64) + np.uint64(rank) * np.uint64(shard_bytes)
    chunks = xsg.make_chunks(off, ln, goffs)
    ctx = xsg.Context(dev_index)
    ctx.set_pattern(pattern)
    shard = xsg.Shard(ctx, shard_t.data_ptr(), cap, chunks)
    counters = torch.zeros((2, xsg.NUM_COUNTERS), dtype=torch.int64, device=dev)  # double-buffered, see step()
    # a dedicated stream: a NULL handle would mean "the ctx's own stream" to xsg_count_async
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    setup_s = time.perf_counter() - t_setup

    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    def all_reduce_(t, op=None):
        """in-place all-reduce of a device tensor (through host memory only in the gloo rehearsal)"""
        if backend == "nccl":
            dist.all_reduce(t) if op is None else dist.all_reduce(t, op=op)
        else:
            h = t.to(coll_dev)
            dist.all_reduce(h) if op is None else dist.all_reduce(h, op=op)
            t.copy_(h)

    if dist is not None:
        exp = torch.tensor([expected_local], dtype=torch.int64, device=dev)
        all_reduce_(exp)
        expected_total = int(exp.item())
    else:
        expected_total = expected_local

    # N>1: the 32-byte all_reduce of step i runs on its own stream while the scan of step i+1 already
    # reads HBM (two counter buffers; events order scan -> all_reduce -> reuse of the buffer).
    coll_stream = torch.cuda.Stream(device=dev) if dist is not None else None
    scan_done = [torch.cuda.Event(), torch.cuda.Event()]
    coll_done = [torch.cuda.Event(), torch.cuda.Event()]

    def step(i, results=None):
        b = i & 1
        c = counters[b]
        if dist is None:
            shard.count_async(xsg.COUNT_MATCHES, stream.cuda_stream, c.data_ptr())
            if results is not None:
                results[i] = c[xsg.CTR_MATCHES]
            return
        stream.wait_event(coll_done[b])  # the all_reduce that used this buffer two steps ago
        shard.count_async(xsg.COUNT_MATCHES, stream.cuda_stream, c.data_ptr())
        scan_done[b].record(stream)
        with torch.cuda.stream(coll_stream):
            coll_stream.wait_event(scan_done[b])
            all_reduce_(c)  # RCCL sum of the 4 uint64 counters
            if results is not None:
                results[i] = c[xsg.CTR_MATCHES]
            coll_done[b].record(coll_stream)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync_all()
    got = int(counters[(args.warmup - 1) & 1][xsg.CTR_MATCHES].item()) if args.warmup else expected_total
    if got != expected_total:
        raise SystemExit(f"PARITY FAILURE before timing: count {got} != expected {expected_total}")

    results = torch.zeros(args.steps, dtype=torch.int64, device=dev)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, results)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce_(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    bad = [int(x) for x in results.cpu().tolist() if int(x) != expected_total]
    if bad:
        raise SystemExit(f"PARITY FAILURE in timed steps: {bad[:4]} != expected {expected_total}")

    total_bytes = shard_bytes * world
    gibs = total_bytes * args.steps / elapsed / 2**30
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel (k_scan), HIP events on the stream it runs on
    kernel_ms = shard.time_scan_kernel(xsg.COUNT_MATCHES, args.kernel_iters)
    achieved = shard_bytes / (kernel_ms * 1e-3) / 1e9  # algorithmic bytes: 1 byte read per input byte
    traffic = None
    tfile = ROOT / "profiles" / "pmc_traffic.json"
    if tfile.exists():
        try:
            tj = json.loads(tfile.read_text())
            if tj.get("bytes_per_gpu") == shard_bytes:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    line = None
    if rank == 0:
        line = {
            "metric": "GiB/s scanned, xs::count literal on plain text resident in HBM",
            "value": round(gibs, 2),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "matches_per_s": round(expected_total * args.steps / elapsed, 1),
            "matches_per_step": expected_total,
            "parity": "count of every timed step == oracle-derived expected count",
            "config": {
                "workload": f"xs::count literal '{args.pattern}' on {args.gib_per_gpu:g} GiB plain text per GPU "
                            f"({nchunks} newline-aligned {args.chunk_mib} MiB chunks), device-resident",
                "pattern": args.pattern,
                "bytes_per_gpu": shard_bytes,
                "chunks_per_gpu": nchunks,
                "distinct_template_chunks": args.templates,
                "sharding": "one process per GPU, contiguous chunk range per rank, no data-path collective; "
                            "one RCCL all_reduce(sum) of 4 uint64 counters per step, on its own stream under the next scan" if world > 1 else
                            "single GPU",
                "setup_s": round(setup_s, 1),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "xsg::k_scan<3, false, false, false, 4, false>",  # KIND kTwo, no NL/LINES/EMIT, 4 loads, no icase
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": shard_bytes,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, blocks, pattern)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
