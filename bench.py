#!/usr/bin/env python3
"""bench.py -- xs::count on a device-resident plain-text shard, GiB/s scanned.

Metric (BASELINE.json): GiB/s scanned (+ matches/s) for xs::count (literal
'Sherlock') on >= 50 GiB of plain text at 1/2/4/8 MI355X.

One "step" = one pass of the hot path over the rank's whole shard: the bulk scan
kernel over every chunk, the finish kernel (per-tile sums + the reference's
end-of-chunk walk) and -- for N > 1 -- one RCCL all-reduce (sum) of the four
uint64 counters.  Inputs are resident in HBM when the timed region starts.

Two forms are measured in every run:
  weak   : every rank owns --gib-per-gpu GiB (default 50)  -> `value`, "scaling": "weak"
  strong : --gib-per-gpu GiB IN TOTAL, the first 1/N of every rank's chunks       -> `strong`
           (50 GiB over 8 GPUs = 400 chunks of 16 MiB per GPU)

Corpus: synthetic (no network).  T distinct '\\n'-terminated template chunks of
16 MiB (+ a few bytes, like the reference's .meta fixtures) are generated on the
host with x-search_amd/corpus.py and replicated into the shard in a seeded
pseudo-random order, so every byte of the shard is real text at a distinct HBM
address (the working set is >> the 256 MiB Infinity Cache) and the expected
count of the full shard is known exactly from the oracle's counts of the T
templates.  The result of EVERY timed step is checked against it.

Launch:  python bench.py --gpus N      N > 1 starts its own N ranks (one per GPU, torch.distributed.run)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N      (what the driver may also do)
"""
import argparse
import json
import os
import socket
import subprocess
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
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-iters", type=int, default=10)
    ap.add_argument("--e2e-gib", type=float, default=4.0,
                    help="size of the tmpfs file of the end-to-end (file -> result) leg; 0 = skip")
    ap.add_argument("--cli-gib", type=float, default=10.0,
                    help="N=1: size of the large tmpfs file of the `cli` block (fresh-process wall times of xsgrep, the README's "
                         "my_grep, GNU grep and the reference CPU path; BASELINE config 1's 100 MB file is always timed); 0 = skip")
    ap.add_argument("--no-tune", action="store_true", help="keep the per-variant default stagger")
    ap.add_argument("--no-regex", action="store_true", help="skip the regex leg (N=1: a few expressions on the resident shard)")
    ap.add_argument("--configs-gib", type=float, default=10.0,
                    help="N=1: size of the sub-shard of the resident buffer on which BASELINE configs 2 and 4 (the list "
                         "tags) are timed and checked element by element; 0 = skip")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves.  Must happen before anything touches the GPU
# (a process that has initialised HIP must never be replaced or forked into ranks).
# ---------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout:  # relay; rank 0 prints the one JSON line
        sys.stdout.write(line)
        sys.stdout.flush()
    return p.wait()


def template_blocks(args, pattern: bytes):
    import corpus
    target = args.chunk_mib << 20
    blocks = []
    for i in range(args.templates):
        extra = 1 + (corpus._mix(args.seed, 1000 + i) % 61)  # fixtures: 16 MiB extended to just past the next '\n'
        blocks.append(corpus.text_block(args.seed, i, target + extra, needle=pattern, lexicon=getattr(args, "lexicon", None)))
    return blocks


def chunk_plan(args, rank: int, nchunks: int):
    """template id of every chunk of this rank's shard (seeded, rank-dependent)"""
    import corpus
    return np.array([corpus._mix(args.seed ^ 0xC0FFEE, (rank << 32) + c) % args.templates for c in range(nchunks)],
                    dtype=np.int64)


# ---------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------
def host_topology():
    """hardware threads we may run on, physical cores and sockets among them, NUMA nodes, cgroup CPU quota"""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores, sockets = set(), set()
    try:
        cpu, phys, core = None, None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip():
                if cpu in allowed and phys is not None and core is not None:
                    cores.add((phys, core))
                    sockets.add(phys)
                cpu = phys = core = None
    except OSError:
        pass
    try:
        nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        nodes = 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    mem_avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                mem_avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    return {"hw_threads": len(allowed), "physical_cores": len(cores) or len(allowed), "sockets": len(sockets) or 1,
            "numa_nodes": nodes, "cgroup_cpu_quota": quota, "mem_available": mem_avail}


def cpu_baseline(args, blocks, pattern: bytes):
    """The reference's CPU path on this host's cores, on a bounded sample of the same workload: chunks of the same
    corpus in RAM, a persistent pool of worker threads pulling chunk indices from a shared counter like
    include/xsearch/Searcher.h:100-120 (one thread pool per search, >= 4 chunks per thread, pages first-touched by
    the workers so that they spread over the NUMA nodes), T = 1 / physical cores / hardware threads."""
    from xs_oracle import CpuPoolCorpus, Oracle, Reference
    orc = Oracle()
    kind = "port"
    if Reference.available():
        try:
            orc.use_reference_primitives(Reference())  # the timed work is the reference's own compiled simd_search.cpp
            kind = "reference"
        except Exception:
            kind = "port"
    topo = host_topology()
    hw, pc = topo["hw_threads"], topo["physical_cores"]
    chunk_bytes = int(blocks[0].size)
    want_chunks = max(64, 4 * hw)
    if topo["mem_available"]:
        want_chunks = min(want_chunks, max(16, int(topo["mem_available"] * 0.25) // chunk_bytes))
    plan = chunk_plan(args, 0x7777, want_chunks)
    corp = CpuPoolCorpus(orc, blocks, plan, nthreads_touch=hw)
    tcount = [orc.count(b, pattern, False) for b in blocks]
    expect_all = int(sum(tcount[int(t)] for t in plan))
    ts = {1, pc, hw}
    if topo["cgroup_cpu_quota"]:  # the container may run only that many CPUs' worth of time: one thread each
        ts.add(max(1, int(topo["cgroup_cpu_quota"])))
    ts = sorted(ts)
    per_t = args.cpu_seconds / (len(ts) + 1)
    rates = {}
    for t in ts:
        first = min(want_chunks, 64) if t == 1 else want_chunks  # one thread: 1 GiB (beyond the L3) is plenty
        tot, sec, nb = corp.count(pattern, t, 1, first_chunks=first)  # calibration pass
        passes = max(1, min(200, int(per_t / max(sec, 1e-4))))
        tot, sec, nb = corp.count(pattern, t, passes, first_chunks=first)
        exp = int(sum(tcount[int(x)] for x in plan[:first]))
        if tot != exp:
            raise SystemExit(f"CPU baseline: count {tot} != expected {exp}")
        rates[t] = (nb * passes / sec / 2**30, passes, first)
    corp.close()
    orc.use_reference_primitives(None)
    best_t = max((t for t in ts if t > 1), key=lambda t: rates[t][0], default=1)
    r1, rbest = rates[1][0], rates[best_t][0]
    note = (f"{best_t} threads are {rbest / r1:.1f}x one thread"
            + (f"; the cgroup of this process allows {topo['cgroup_cpu_quota']} CPUs' worth of time (cpu.max), so "
               f"thread counts beyond that are throttled, not scaled" if topo["cgroup_cpu_quota"] else "")
            + ("; beyond that the search is DRAM-bandwidth-bound" if rbest / r1 < 0.5 * best_t and not topo["cgroup_cpu_quota"] else ""))
    return {
        "value": round(rbest, 3),
        "unit": "GiB/s",
        "cores": best_t,
        "kind": kind,
        "value_1thread": round(r1, 3),
        "by_threads": {str(t): round(rates[t][0], 3) for t in ts},
        "physical_cores": pc, "hw_threads": hw, "sockets": topo["sockets"], "numa_nodes": topo["numa_nodes"],
        "cgroup_cpu_quota": topo["cgroup_cpu_quota"],
        # every pass of every thread count returned exactly the count the oracle's per-template counts add up to (a
        # difference is a SystemExit above); the whole sample holds `expected_matches_in_sample` occurrences
        "every_pass_count_equals_expected": True,
        "expected_matches_in_sample": expect_all,
        "note": note,
        "sample": f"{want_chunks} chunks x {args.chunk_mib} MiB = {want_chunks * chunk_bytes / 2**30:.1f} GiB of the same corpus in RAM "
                  f"(first-touched by the workers), persistent thread pool, chunk work-stealing, xs::count; "
                  + ", ".join(f"T={t}: {rates[t][1]} passes over {rates[t][2]} chunks" for t in ts),
    }


# ---------------------------------------------------------------------------------------------------
# end-to-end leg: tmpfs file -> reader threads -> pinned buffers -> H2D -> scan -> result
# ---------------------------------------------------------------------------------------------------
def e2e_leg(args, blocks, pattern: bytes, tcount, rank, world, dist, dev_index):
    """file -> result wall time of xs::extern_search's pipeline (xsg_job_*) on a tmpfs file: PCIe- and
    host-read-bound, reported beside -- never as -- `value`.  N > 1: every rank searches its contiguous chunk range
    of the SAME file (config 3/5's fan-out); the rate is file bytes / the slowest rank's wall time."""
    import xsg
    # the file grows with the number of GPUs (every rank gets --e2e-gib of it, 16 GiB at most in all)
    n = max(world, int(min(args.e2e_gib * world, 16.0) * 2**30 / (args.chunk_mib << 20)))
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    # the file, a quarter of it again for the LZ4 leg and its compressed copy must fit the tmpfs: rank 0 looks, all ranks agree
    box = [n]
    if rank == 0:
        try:
            st = os.statvfs(d)
            room = int(st.f_bavail * st.f_frsize * 0.8)
            box[0] = min(n, int(room / 1.7) // (args.chunk_mib << 20))
        except OSError:
            pass
    if dist is not None:
        dist.broadcast_object_list(box, src=0)
    n = int(box[0])
    if n < max(world, 4):
        return {"skipped": f"no room in {d} for a file of at least {max(world, 4)} chunks"}
    plan = chunk_plan(args, 0xE2E, n)
    path = os.path.join(d, f"xsg_bench_e2e_{os.environ.get('MASTER_PORT', os.getpid())}.txt")
    out = {}
    made = []
    try:
        if rank == 0:
            with open(path, "wb") as f:
                for c in plan:
                    f.write(blocks[int(c)].tobytes())
            made.append(path)
        if dist is not None:
            dist.barrier()
        size = os.path.getsize(path)
        want_total = int(sum(tcount[int(c)] for c in plan))
        # The ranks share the file by ranges of the LIBRARY's chunk plan of it (16 MiB extended to just past the next
        # newline: xsg_plan_chunks), which is not the list of template chunks the file was written from -- a template
        # ends a few bytes behind the first newline past its 16 MiB, so the plan's cuts drift by tens of bytes.  What a
        # rank must return is therefore the oracle's result on ITS chunks as the library cuts them: byte offsets global,
        # line indices counted from the range's first line (xsg_file.cpp: publish).
        chunk_bytes = args.chunk_mib << 20
        fplan = xsg.plan_chunks(path, chunk_bytes)
        nf = len(fplan)
        lo, hi = (nf * rank) // world, (nf * (rank + 1)) // world
        # threads from the CPU time this process group may use (cgroup quota, else hardware threads), shared by the
        # ranks of the node: 4 device workers + 8 readers per GPU when there is room (one GPU in a 16-CPU cgroup), one
        # of each when 8 ranks share 16 CPUs
        topo = host_topology()
        budget = max(2, int((topo["cgroup_cpu_quota"] or topo["hw_threads"]) // max(world, 1)))
        nthreads = int(os.environ.get("XSG_E2E_WORKERS", str(max(1, min(4, budget // 3)))))
        nreaders = int(os.environ.get("XSG_E2E_READERS", str(max(1, min(8, budget - max(1, min(4, budget // 3)))))))

        def run(mode, meta=None, data=None, nrd=None, rng=None):
            rng = rng or (lo, hi)
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            j = xsg.Job(pattern, data or path, mode, meta_path=meta, device=dev_index, num_threads=nthreads,
                        num_max_readers=nrd or nreaders, chunk_bytes=chunk_bytes, chunk_range=rng if world > 1 else None)
            r = j.result()
            dt = time.perf_counter() - t0
            st = j.stats()
            j.close()
            if dist is not None:  # wall time of the slowest rank (this leg is not a device measurement)
                dt = _allreduce_max_host(dist, dt)
            return r, dt, st

        def oracle_on(file_path, table, lo_, hi_, lists=True):
            """the oracle over chunks [lo_, hi_) of `table` (original_offset / original_size) of a plain-text file"""
            from xs_oracle import Oracle
            orc = Oracle()
            mm = np.memmap(file_path, dtype=np.uint8, mode="r")
            cnt, nl_run, ms, lis, lns = 0, 0, [], [], []
            for i in range(lo_, hi_):
                o, ln = int(table[i]["original_offset"]), int(table[i]["original_size"])
                b = np.array(mm[o:o + ln])
                cnt += orc.count(b, pattern, False)
                if lists:
                    ms.append(orc.byte_offsets_match(b, pattern).astype(np.uint64) + np.uint64(o))
                    lis.append(orc.line_indices(b, pattern, nl_run).astype(np.uint64))
                    lns += orc.lines(b, pattern)
                    nl_run += orc.count_newlines(b)
            del mm
            cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.uint64)
            return cnt, cat(ms), cat(lis), lns

        want_local, want_m, want_li, want_lines = oracle_on(path, fplan, lo, hi)

        run(xsg.COUNT_MATCHES)  # warm: thread/buffer pools, page cache
        r, dt, st = run(xsg.COUNT_MATCHES)
        if r != want_local:
            raise SystemExit(f"e2e PARITY FAILURE: count {r} != {want_local}")
        out["count_gib_s"] = round(size / dt / 2**30, 2)
        r, dt, st = run(xsg.MATCH_BYTE_OFFSETS)
        if not np.array_equal(np.asarray(r, dtype=np.uint64), want_m):
            raise SystemExit(f"e2e PARITY FAILURE: match_byte_offsets differ from the oracle's ({len(r)} against {want_m.size})")
        out["match_byte_offsets_gib_s"] = round(size / dt / 2**30, 2)
        r, dt, st = run(xsg.LINE_INDICES)
        if not np.array_equal(np.asarray(r, dtype=np.uint64), want_li):
            raise SystemExit(f"e2e PARITY FAILURE: line_indices differ from the oracle's ({len(r)} against {want_li.size})")
        out["line_indices_gib_s"] = round(size / dt / 2**30, 2)
        r, dt, st = run(xsg.LINES)
        if list(r) != want_lines:
            raise SystemExit(f"e2e PARITY FAILURE: lines differ from the oracle's ({len(r)} against {len(want_lines)})")
        out["lines_gib_s"] = round(size / dt / 2**30, 2)
        out["parity"] = (f"count == expected; {want_m.size} match offsets, {want_li.size} line indices and {len(want_lines)} lines "
                         f"== the oracle's, element by element")
        # config 5: LZ4 blocks + metafile, host decode overlapped with H2D + scan
        mp, dp = path + ".lz4.meta", path + ".lz4"
        if rank == 0:
            small = path + ".part"
            nsmall = min(max(world, n // 4), 128)
            with open(small, "wb") as f:
                for c in plan[:nsmall]:
                    f.write(blocks[int(c)].tobytes())
            made.append(small)
            xsg.meta_write(small, mp, dp, xsg.COMPRESSION_LZ4)
            made += [mp, dp]
        if dist is not None:
            dist.barrier()
        nsmall = min(max(world, n // 4), 128)
        small = path + ".part"  # (rank 0 unlinks it at the end; every rank reads its own chunks of it for the oracle)
        _, mchunks = xsg.meta_read(mp)  # the ranges of THIS leg are ranges of the metafile's chunk table
        nm = len(mchunks)
        llo, lhi = (nm * rank) // world, (nm * (rank + 1)) // world
        want_l = oracle_on(small, mchunks, llo, lhi, lists=False)[0]
        ndecoders = int(os.environ.get("XSG_E2E_DECODERS", str(max(1, min(16, budget)))))  # the decoders are the work; the device workers mostly sleep
        run(xsg.COUNT_MATCHES, mp, dp, ndecoders, (llo, lhi))
        r, dt, st = run(xsg.COUNT_MATCHES, mp, dp, ndecoders, (llo, lhi))
        if r != want_l:
            raise SystemExit(f"e2e PARITY FAILURE (lz4): count {r} != {want_l}")
        small_bytes = int(sum(blocks[int(c)].size for c in plan[:nsmall]))
        out["lz4_metafile_count_gib_s"] = round(small_bytes / dt / 2**30, 2)
        out["lz4_compressed_fraction"] = round(os.path.getsize(dp) / small_bytes, 3) if rank == 0 else None
        # config 5's accounting (this rank's job): which decoder ran, what one decoder thread moves, where the wall time went
        dec_s, rd_s, dev_s = st["seconds_decompress"], st["seconds_read"], st["seconds_device"]
        out["lz4"] = {
            "decoder": xsg.codec_name(xsg.COMPRESSION_LZ4),
            "reader_decoder_threads": ndecoders, "device_workers": nthreads,
            "decode_gib_s_per_thread": round(st["bytes_scanned"] / max(dec_s, 1e-9) / 2**30, 2),
            "decode_thread_seconds": round(dec_s, 4), "read_thread_seconds": round(rd_s, 4),
            "device_worker_seconds": round(dev_s, 4), "wall_seconds": round(dt, 4),
            "decode_share_of_reader_time": round(dec_s / max(dec_s + rd_s, 1e-9), 3),
            "bound": "host LZ4 decode: wall time ~ decode thread-seconds / threads; the H2D copies and scans of the same bytes "
                     "take device_worker_seconds in all and overlap with it",
        }
        out["file_gib"] = round(size / 2**30, 2)
        out["threads"] = (f"{nthreads} device workers + {nreaders} readers per GPU ({ndecoders} reader/decoder threads for LZ4); "
                          f"CPU budget per rank {budget}")
        out["what"] = ("wall time from xsg_job_start to join on a tmpfs file (page-cache read -> pinned buffers -> "
                       "hipMemcpyAsync -> scan -> ordered result); bounded by PCIe Gen5 x16 and the host read path, "
                       "not by HBM")
        out["total_matches"] = want_total
    finally:
        if dist is not None:
            dist.barrier()
        for f in made:
            if os.path.exists(f):
                os.unlink(f)
    return out


def _allreduce_max_host(dist, x: float) -> float:
    objs = [None] * dist.get_world_size()
    dist.all_gather_object(objs, float(x))
    return max(objs)


def regex_leg(xsg, torch, ctx, shard, stream, shard_bytes):
    """The regex row on the resident shard, N=1 only, reported next to the headline (never `value`):
    whole synchronous xsg_count calls per expression -- class sequences inside k_scan, variable-length expressions on
    the automaton route (prefilter where the expression starts selectively, k_rx_scan otherwise).  Every count is
    cross-checked between the two device routes that can serve it: the synchronous call and xsg_count_async."""
    out = []
    c = torch.zeros(xsg.NUM_COUNTERS, dtype=torch.int64, device=f"cuda:{torch.cuda.current_device()}")
    for expr in ("She[r ]lock", "[Ss]herlock", "Sherlock|Holmes", "Sher.*mes", "colou?r", "\\w+ing"):
        ctx.set_pattern(expr.encode(), xsg.FLAG_REGEX)
        # the first call on a (binding, pattern) also pays what the library measures or builds once: the hot-filter probe
        # of the window kinds, the factor prefilter's tile marks.  Reported on its own; for an expression whose later
        # calls reuse such marks (they do not read the whole shard again) the rate is the FIRST call's.
        t0 = time.perf_counter()
        got = int(shard.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES])
        first_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for _ in range(3):
            got = int(shard.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES])
        ms = (time.perf_counter() - t0) / 3 * 1e3
        marked = "factor prefilter" in shard.scan_kernel_name(xsg.COUNT_MATCHES)
        rate_ms = first_ms if marked else ms
        shard.count_async(xsg.COUNT_MATCHES, stream.cuda_stream, c.data_ptr())
        torch.cuda.synchronize()
        other = int(c[xsg.CTR_MATCHES])
        if other != got:
            raise SystemExit(f"regex PARITY FAILURE: {expr!r}: xsg_count {got} != xsg_count_async {other}")
        out.append({"expr": expr, "matches": got, "first_call_ms": round(first_ms, 3), "ms_per_call": round(ms, 3),
                    "gbs": round(shard_bytes / rate_ms / 1e6, 1),
                    "frac_of_hbm_peak": round(shard_bytes / rate_ms / 1e6 / HBM_PEAK_GBS, 4),
                    "rate_is_of": "the first call (later calls reuse the tile marks)" if marked else "repeated calls",
                    "kernel": shard.scan_kernel_name(xsg.COUNT_MATCHES)})
    return {"what": "whole synchronous xsg_count(COUNT_MATCHES) calls with XSG_FLAG_REGEX on the same resident shard; "
                    "each count equals the one xsg_count_async computes on its own route", "cases": out}


def configs_leg(xsg, ctx, orc, shard_t, cap, chunks, plan, blocks, ln, pattern, gib):
    """BASELINE configs 2 and 4 device-resident (N=1, never `value`): xs::match_byte_offsets, xs::line_byte_offsets,
    xs::line_indices and xs::lines on the first `gib` GiB of the resident shard, whole calls at the C ABI (search +
    the result in host memory, scripts/config_times.py), next to xs::count on the same sub-shard.  Every result is
    compared element by element with the oracle's results on the template chunks (chunks are independent units:
    offsets = template-local offsets + the chunk's global offset, line indices = local indices + newlines before the
    chunk, lines = the template's lines), as tests/test_gpu_fullsize.py does."""
    import ctypes as C
    sys.path.insert(0, str(ROOT / "scripts"))
    from config_times import timed_calls
    n = min(len(plan), max(1, int(round(gib * 2**30 / (16 << 20)))))
    sub = chunks[:n].copy()
    sub["global_offset"] = np.concatenate([[0], np.cumsum(ln[:n])[:-1]]).astype(np.uint64)
    nbytes = int(ln[:n].sum())
    sh = xsg.Shard(ctx, shard_t.data_ptr(), cap, sub)
    lib = xsg.load()
    ctx.set_pattern(pattern)
    out = {"what": f"whole C-ABI calls (search + result in host memory) on the first {n} chunks = {nbytes / 2**30:.2f} GiB of "
                   f"the resident shard; every list compared element by element with the oracle", "bytes": nbytes, "tags": {}}
    # first xs::line_indices of a binding also counts the newlines per tile (cached afterwards)
    t0 = time.perf_counter()
    first_idx = sh.search_u64_view(xsg.LINE_INDICES).copy()
    out["line_indices_first_call_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    # 21 calls per tag, the median: the first handful of count_lines calls of this leg run up to 25 % slower than the rest
    # (GPU timestamps: 1.52, 1.92, 1.93, 1.81, 1.74, 1.66, 1.57, 1.57 ms ... for one kernel on the same data, only inside
    # this process, after seconds of full-rate scanning; scripts/lines_transient.py does not reproduce it on a fresh
    # process) -- min and max are in the record
    spread = {}
    r = timed_calls(lib, sh, 21, spread=spread)
    out["kernels"] = {"count": sh.scan_kernel_name(xsg.COUNT_MATCHES), "count_lines": sh.scan_kernel_name(xsg.COUNT_LINES)}
    base = r["count"][1]
    for tag, (cnt, ms) in r.items():
        out["tags"][tag] = {"results": cnt, "ms": round(ms, 4), "ms_min": round(spread[tag][0], 4), "ms_max": round(spread[tag][1], 4),
                            "gbs": round(nbytes / ms / 1e6, 1),
                            "frac_of_hbm_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4), "vs_count": round(ms / base, 3)}
    # ---- element-wise check
    goffs = sub["global_offset"]
    pl = plan[:n]
    nl_t = np.array([orc.count_newlines(b) for b in blocks], dtype=np.uint64)
    nl_before = np.concatenate([[0], np.cumsum(nl_t[pl])[:-1]]).astype(np.uint64)
    want_m = np.concatenate([orc.byte_offsets_match(blocks[int(c)], pattern) + goffs[i] for i, c in enumerate(pl)])
    lo_t = [orc.byte_offsets_line(b, pattern) for b in blocks]
    want_lo = np.concatenate([lo_t[int(c)] + goffs[i] for i, c in enumerate(pl)])
    li_t = [orc.line_indices(b, pattern, 0) for b in blocks]
    want_li = np.concatenate([li_t[int(c)] + nl_before[i] for i, c in enumerate(pl)])
    lines_t = [orc.lines(b, pattern) for b in blocks]
    want_lines = [l for c in pl for l in lines_t[int(c)]]
    bad = []
    if not np.array_equal(sh.search_u64(xsg.MATCH_BYTE_OFFSETS), want_m):
        bad.append("match_byte_offsets")
    if not np.array_equal(sh.search_u64(xsg.LINE_BYTE_OFFSETS), want_lo):
        bad.append("line_byte_offsets")
    if not (np.array_equal(sh.search_u64(xsg.LINE_INDICES), want_li) and np.array_equal(first_idx, want_li)):
        bad.append("line_indices")
    got_lines, got_off = sh.search_lines()
    if got_lines != want_lines or not np.array_equal(got_off, want_lo):
        bad.append("lines")
    if bad:
        raise SystemExit(f"configs PARITY FAILURE: {bad}")
    out["parity"] = f"{want_m.size} offsets, {want_lo.size} line offsets, {want_li.size} line indices, {len(want_lines)} lines == oracle"
    # ---- a needle that is in most lines: the list tags are then bound by what they write and move, not by the scan
    ctx.set_pattern(b"She")
    d = timed_calls(lib, sh, 3)
    out["dense_needle"] = {"pattern": "She", "tags": {t: {"results": c, "ms": round(ms, 3)} for t, (c, ms) in d.items()},
                           "note": "tens of millions of results: the pinned D2H of the lists dominates (xs::lines: lengths, offsets and 2 GB of line bytes through xsg_result_lines_view)"}
    tm = np.array([orc.count(b, b"She", False) for b in blocks], dtype=np.int64)
    if d["count"][0] != int(tm[pl].sum()) or d["match_byte_offsets"][0] != int(tm[pl].sum()):
        raise SystemExit("configs PARITY FAILURE: dense needle count")
    ctx.set_pattern(pattern)
    sh.close()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    # ---- N = 1, first (before this process touches the GPU): the reference's one published measurement on this box --
    # `time grep Sherlock FILE` against `time my_grep Sherlock FILE` (README.md:44-62): fresh processes, wall time, outputs
    # compared byte for byte.  Never `value`.
    cli = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and os.environ.get("XSG_BENCH_CLI", "1") != "0":
        sys.path.insert(0, str(ROOT / "scripts"))
        try:
            from cli_clock import cli_block
            sizes = [("config1_100MB", 100_000_000, 5)]
            if args.cli_gib > 0:
                sizes.append((f"{args.cli_gib:g}GiB", int(args.cli_gib * 2**30), 3))
            cli = cli_block(tuple(sizes), pattern=args.pattern.encode("latin-1"))
        except SystemExit:
            raise
        except Exception as ex:  # a missing tool, no room in /dev/shm: reported, not fatal
            cli = {"error": f"{type(ex).__name__}: {ex}"}

    import torch
    import xsg
    from xs_oracle import Oracle

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Rehearsal hook (not used by the driver): XSG_BENCH_BACKEND=gloo runs the N>1 code path on a box with
    # fewer GPUs than ranks (all ranks scan on the visible GPUs, the collective goes through host memory).
    backend = os.environ.get("XSG_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()  # does not initialise the GPU
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    if dev_index >= ndev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({ndev} visible)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # XSG_BENCH_FORCE_DIST=1 (rehearsal, not used by the driver): take the N>1 code path -- RCCL
    # process group, all_reduce per step, barriers -- with a single rank, on a one-GPU box.
    if world > 1 or os.environ.get("XSG_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist_mod
        dist = dist_mod
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
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
    nchunks = max(world, int(round(args.gib_per_gpu * 2**30 / (args.chunk_mib << 20))))
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
    # strong form: the same --gib-per-gpu GiB IN TOTAL, i.e. the first 1/world of this rank's chunks
    n_strong = max(1, nchunks // world)
    shard_strong = shard if world == 1 else xsg.Shard(ctx, shard_t.data_ptr(), cap, chunks[:n_strong])
    strong_bytes = int(ln[:n_strong].sum())
    expected_strong_local = int(tcount[plan[:n_strong]].sum())
    counters = torch.zeros((2, xsg.NUM_COUNTERS), dtype=torch.int64, device=dev)  # double-buffered, see step()
    # a dedicated stream: a NULL handle would mean "the ctx's own stream" to xsg_count_async
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    # What a caller WITHOUT xsg_shard_tune gets, first call included (N=1; never `value`): a fresh binding, the
    # synchronous xsg_count -- the first call also runs the library's hot-filter probe.
    untuned = None
    if world == 1:
        su = xsg.Shard(ctx, shard_t.data_ptr(), cap, chunks)
        t0 = time.perf_counter()
        got = int(su.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES])
        first_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for _ in range(5):
            got = int(su.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES])
        ms = (time.perf_counter() - t0) / 5 * 1e3
        if got != expected_local:
            raise SystemExit(f"PARITY FAILURE (untuned xsg_count): {got} != {expected_local}")
        untuned = {"first_call_ms": round(first_ms, 3), "first_call_gib_s": round(shard_bytes / first_ms / 2**30 * 1e3, 1),
                   "ms_per_call": round(ms, 4), "gib_s": round(shard_bytes / ms / 2**30 * 1e3, 1),
                   "kernel": su.scan_kernel_name(xsg.COUNT_MATCHES),
                   "what": "synchronous xsg_count on a fresh binding, no xsg_shard_tune: the first call includes the "
                           "library's per-(binding, pattern) probe, later calls use the default stagger"}
        su.close()
    # the wave stagger of the bulk kernel: measured on this shard (a few launches), not a constant
    stagger = None if args.no_tune else shard.tune(xsg.COUNT_MATCHES)
    stagger_strong = stagger if (world == 1 or args.no_tune) else shard_strong.tune(xsg.COUNT_MATCHES)
    # one untimed launch per shard in any case: the library's first pass on a (binding, pattern) measures which hot
    # filter suits the data (a few short launches and a sync) -- set-up, not a step
    shard.time_scan_kernel(xsg.COUNT_MATCHES, 1)
    if shard_strong is not shard:
        shard_strong.time_scan_kernel(xsg.COUNT_MATCHES, 1)
    kernel_name = shard.scan_kernel_name(xsg.COUNT_MATCHES)
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

    def reduce_int(v):
        if dist is None:
            return int(v)
        x = torch.tensor([int(v)], dtype=torch.int64, device=dev)
        all_reduce_(x)
        return int(x.item())

    expected_total = reduce_int(expected_local)
    expected_strong = reduce_int(expected_strong_local)

    # N>1: the 32-byte all_reduce of step i runs on its own stream while the scan of step i+1 already
    # reads HBM (two counter buffers; events order scan -> all_reduce -> reuse of the buffer).
    coll_stream = torch.cuda.Stream(device=dev) if dist is not None else None
    scan_done = [torch.cuda.Event(), torch.cuda.Event()]
    coll_done = [torch.cuda.Event(), torch.cuda.Event()]

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(sh, expected, steps, warmup, lib_comm=None):
        """W untimed + exactly K timed steps on shard `sh`; -> seconds of the K steps (max over ranks).
        lib_comm: the per-step all-reduce goes through the library's own RCCL communicator instead of torch's"""
        def step(i, results=None):
            b = i & 1
            c = counters[b]
            if dist is None:
                c.fill_(-1)  # poison on the scan's stream (32 bytes): a stale value of two steps ago cannot pass the check
                sh.count_async(xsg.COUNT_MATCHES, stream.cuda_stream, c.data_ptr())
                if results is not None:
                    results[i] = c[xsg.CTR_MATCHES]
                return
            stream.wait_event(coll_done[b])  # the all_reduce that used this buffer two steps ago
            c.fill_(-1)
            sh.count_async(xsg.COUNT_MATCHES, stream.cuda_stream, c.data_ptr())
            scan_done[b].record(stream)
            with torch.cuda.stream(coll_stream):
                coll_stream.wait_event(scan_done[b])
                if lib_comm is not None:  # ncclAllReduce(sum, uint64 x 4) on the library's communicator
                    lib_comm.reduce_counts_async(c.data_ptr(), xsg.NUM_COUNTERS, coll_stream.cuda_stream)
                else:
                    all_reduce_(c)  # RCCL sum of the 4 uint64 counters
                if results is not None:
                    results[i] = c[xsg.CTR_MATCHES]
                coll_done[b].record(coll_stream)

        for i in range(warmup):
            step(i)
        sync_all()
        if warmup:
            got = int(counters[(warmup - 1) & 1][xsg.CTR_MATCHES].item())
            if got != expected:
                raise SystemExit(f"PARITY FAILURE before timing: count {got} != expected {expected}")
        results = torch.zeros(steps, dtype=torch.int64, device=dev)
        sync_all()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, results)
        sync_all()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            all_reduce_(el, op=dist.ReduceOp.MAX)
            elapsed = float(el.item())
        bad = [int(x) for x in results.cpu().tolist() if int(x) != expected]
        if bad:
            raise SystemExit(f"PARITY FAILURE in timed steps: {bad[:4]} != expected {expected}")
        return elapsed

    elapsed = measure(shard, expected_total, args.steps, args.warmup)
    total_bytes = shard_bytes * world
    gibs = total_bytes * args.steps / elapsed / 2**30
    ms_per_step = elapsed / args.steps * 1e3

    if world == 1:
        elapsed_s, strong_total = elapsed, total_bytes
    else:
        elapsed_s = measure(shard_strong, expected_strong, args.steps, args.warmup)
        sb = torch.tensor([strong_bytes], dtype=torch.int64, device=dev)
        all_reduce_(sb)
        strong_total = int(sb.item())
    strong = {
        "value": round(strong_total * args.steps / elapsed_s / 2**30, 2),
        "unit": "GiB/s",
        "ms_per_step": round(elapsed_s / args.steps * 1e3, 4),
        "total_gib": round(strong_total / 2**30, 3),
        "chunks_per_gpu": n_strong,
        "bytes_per_gpu": strong_bytes,
        "ideal_ms_at_kernel_rate": None,  # filled below
        "stagger": stagger_strong,
        "what": f"{args.gib_per_gpu:g} GiB in total over {world} GPU(s) (SURVEY 8d config 3), same step, same checks",
    }

    # ---- the collective alone: latency of one all_reduce of the 4 counters (N > 1)
    allreduce_us = None
    if dist is not None:
        c = counters[0]
        for _ in range(5):
            all_reduce_(c)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(50):
            all_reduce_(c)
        torch.cuda.synchronize()
        allreduce_us = (time.perf_counter() - t0) / 50 * 1e6

    # ---- roofline of the dominant kernel (k_scan), HIP events on the stream it runs on
    kernel_ms = shard.time_scan_kernel(xsg.COUNT_MATCHES, args.kernel_iters)
    achieved = shard_bytes / (kernel_ms * 1e-3) / 1e9  # algorithmic bytes: 1 byte read per input byte
    strong["ideal_ms_at_kernel_rate"] = round(strong_bytes / (achieved * 1e9) * 1e3, 4)
    traffic, traffic_source = None, None
    for tfile in sorted((ROOT / "profiles").glob("*pmc_traffic*.json"), reverse=True):
        try:
            tj = json.loads(tfile.read_text())
        except Exception:
            continue
        # only a record of THIS workload and of the very instantiation timed here counts (profiles/r03_pmc_traffic.json
        # holds both hot-filter instantiations of the plain count; what xsg_shard_tune picks moves from box to box)
        rec = tj.get("by_kernel", {}).get(kernel_name.split(" stagger")[0])
        if rec and tj.get("bytes_per_gpu") == shard_bytes and tj.get("pattern", "Sherlock") == args.pattern:
            traffic = rec.get("hbm_bytes_per_launch")
            traffic_source = f"profiles/{tfile.name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command with " \
                             f"this instantiation (separate runs; not measured in this run)"
            break

    regex = None
    if world == 1 and not args.no_regex:
        regex = regex_leg(xsg, torch, ctx, shard, stream, shard_bytes)
        ctx.set_pattern(pattern)

    configs = None
    if world == 1 and args.configs_gib > 0:
        configs = configs_leg(xsg, ctx, orc, shard_t, cap, chunks, plan, blocks, ln, pattern, args.configs_gib)

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
            "parity": "count of every timed step == oracle-derived expected count (the counter buffer is poisoned on the "
                      "scan's stream before every step)",
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
            "strong": strong,
            "rccl_ranks": (world if (dist is not None and backend == "nccl") else 0),
            "collective_backend": (backend if dist is not None else None),
            "collective_through": None if dist is None else "torch.distributed",
            "allreduce_us": None if allreduce_us is None else round(allreduce_us, 1),
            # N > 1: the same strong-form steps with the per-step all-reduce on the LIBRARY's RCCL communicator
            # (xsg_comm_create_rank + xsg_reduce_counts_async); null when a rank has no librccl or the leg did not finish
            "strong_lib": None,
            "allreduce_us_lib": None,
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,  # from the library: the instantiation this pattern and mode launch
                "stagger": stagger,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": shard_bytes,
            },
        }
        if cli is not None:
            line["cli"] = cli
        if untuned is not None:
            line["untuned"] = untuned
        if configs is not None:
            line["configs"] = configs
        if regex is not None:
            line["regex"] = regex

    # ---- Everything the contract asks for is measured and in `line`.  The legs below add to it; at N > 1 they are
    # collective (a file all ranks search, a second communicator) and have never met more than one GPU before the
    # driver's own multi-GPU run, so they run under a watchdog: a rank that waits for one that failed gives up after the
    # limit, rank 0 prints the line with what it has, every rank exits 0.
    import threading

    def give_up(what):
        def fire():
            if rank == 0:
                line.setdefault("notes", []).append(f"{what} did not finish within its limit and was abandoned")
                print(json.dumps(line), flush=True)
            os._exit(0)
        return fire

    e2e = None
    if args.e2e_gib > 0:
        dog = threading.Timer(300.0, give_up("the end-to-end leg")) if world > 1 else None
        if dog:
            dog.daemon = True
            dog.start()
        try:
            e2e = e2e_leg(args, blocks, pattern, tcount, rank, world, dist, dev_index)
        except SystemExit:
            raise  # a parity failure is a failure
        except Exception as ex:  # no room in /dev/shm, a job that could not start, ...: reported, not fatal
            if world > 1:  # the other ranks may be waiting in the leg's barriers: let the watchdog end the run
                if rank == 0:
                    line.setdefault("notes", []).append(f"e2e leg failed on rank 0: {type(ex).__name__}: {ex}")
                dog = None
                give_up("the end-to-end leg (an exception on this rank)")()
            e2e = {"error": f"{type(ex).__name__}: {ex}"}
        finally:
            if dog:
                dog.cancel()
    if rank == 0:
        if e2e is not None:
            line["e2e"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, blocks, pattern)

    # ---- N > 1, last: the product's own exchange step (include/xsg.h "Multi-GPU"; reference: the only parallelism is
    # include/xsearch/Searcher.h:141-145).  Everything above is already measured; this leg runs under a watchdog so
    # that a communicator that never forms (it has never met 8 GPUs before the driver's run) costs the two lib
    # fields, not the record: after 120 s every rank gives up, rank 0 prints the line with nulls.
    if dist is not None and backend == "nccl" and world > 1 and os.environ.get("XSG_BENCH_LIB_COLL", "1") != "0":
        dog = threading.Timer(120.0, give_up("the library-communicator leg"))
        dog.daemon = True
        dog.start()
        lib_comm = None
        try:
            have = torch.tensor([1 if xsg.comm_library() else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(have, op=dist.ReduceOp.MIN)
            if int(have.item()) == 1:  # every rank agrees it has a librccl: nobody waits in ncclCommInitRank for a rank that cannot come
                box = [xsg.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                ok = torch.ones(1, dtype=torch.int64, device=dev)
                try:
                    lib_comm = xsg.Comm.rank(ctx, world, rank, box[0])
                except Exception:
                    ok.zero_()
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 1:
                    el = measure(shard_strong, expected_strong, args.steps, args.warmup, lib_comm=lib_comm)
                    c = counters[0]
                    for _ in range(5):
                        lib_comm.reduce_counts_async(c.data_ptr(), xsg.NUM_COUNTERS, stream.cuda_stream)
                    sync_all()
                    t0 = time.perf_counter()
                    for _ in range(50):
                        lib_comm.reduce_counts_async(c.data_ptr(), xsg.NUM_COUNTERS, stream.cuda_stream)
                    torch.cuda.synchronize()
                    us = (time.perf_counter() - t0) / 50 * 1e6
                    if rank == 0:
                        line["strong_lib"] = {"value": round(strong_total * args.steps / el / 2**30, 2), "unit": "GiB/s",
                                              "ms_per_step": round(el / args.steps * 1e3, 4),
                                              "through": "libxsg: xsg_comm_create_rank + xsg_reduce_counts_async "
                                                         f"({xsg.comm_library()})"}
                        line["allreduce_us_lib"] = round(us, 1)
                if lib_comm is not None:
                    torch.cuda.synchronize()
                    lib_comm.close()
        finally:
            dog.cancel()

    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
