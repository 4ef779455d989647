"""The N>1 path on CPU: world_size-2 gloo.  Sharding (contiguous chunk ranges),
the all_reduce(sum) of the counters and the all_gather of newline totals for
line-index bases are exercised with a CPU stand-in for the per-rank scanner (the
oracle; on the GPU box the default scanner is the HIP pipeline)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, q):
    for p in (ROOT / "x-search_amd", ROOT / "oracle", ROOT / "tests"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import dist_search
    import xsg
    from xs_oracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    data = np.fromfile(path, dtype=np.uint8)
    pat = b"Sherlock"
    plan = dist_search.file_plan(path, None, 1 << 18)
    out = {}
    for mode in (xsg.COUNT_MATCHES, xsg.COUNT_LINES, xsg.LINE_INDICES, xsg.MATCH_BYTE_OFFSETS):
        def scan(lo, hi, mode=mode):
            res, nl = [], 0
            cnt = 0
            for c in plan[lo:hi]:
                b = data[int(c["original_offset"]):int(c["original_offset"] + c["original_size"])]
                if mode == xsg.COUNT_MATCHES:
                    cnt += orc.count(b, pat, False)
                elif mode == xsg.COUNT_LINES:
                    cnt += orc.count(b, pat, True)
                elif mode == xsg.LINE_INDICES:
                    res += [int(x) for x in orc.line_indices(b, pat, nl)]  # local to the range, like the job
                else:
                    res += [int(x) + int(c["original_offset"]) for x in orc.byte_offsets_match(b, pat)]
                nl += orc.count_newlines(b)
            return (cnt if mode in (xsg.COUNT_MATCHES, xsg.COUNT_LINES) else np.array(res, dtype=np.uint64)), nl
        r = dist_search.distributed_search(pat, path, mode, dist=dist, chunk_bytes=1 << 18, scan_range=scan)
        out[mode] = r if isinstance(r, int) else [int(x) for x in r]
    out["range"] = dist_search.chunk_range(len(plan), world, rank)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_matches_single_process(tmp_path, oracle):
    import torch.multiprocessing as mp
    import corpus
    import xsg
    data = np.concatenate([corpus.text_block(55, i, 400_000 + i, needle_rate=1e-3) for i in range(5)])
    path = str(tmp_path / "c.txt")
    data.tofile(path)
    plan = xsg.plan_chunks(path, 1 << 18)
    chunks = [data[int(c["original_offset"]):int(c["original_offset"] + c["original_size"])] for c in plan]
    from gpu_util import oracle_all_modes
    want = oracle_all_modes(oracle, chunks, b"Sherlock")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][xsg.COUNT_MATCHES] == got[1][xsg.COUNT_MATCHES] == want["count_matches"]
    assert got[0][xsg.COUNT_LINES] == got[1][xsg.COUNT_LINES] == want["count_lines"]
    # list tags: rank order == file order, no exchange needed
    assert got[0][xsg.MATCH_BYTE_OFFSETS] + got[1][xsg.MATCH_BYTE_OFFSETS] == want["match_byte_offsets"]
    assert got[0][xsg.LINE_INDICES] + got[1][xsg.LINE_INDICES] == want["line_indices"]
    assert got[0]["range"][1] == got[1]["range"][0] and got[1]["range"][1] == len(plan)


def test_chunk_ranges_partition():
    import dist_search
    for n in (0, 1, 7, 8, 3200):
        for w in (1, 2, 4, 8):
            r = [dist_search.chunk_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` (how the driver calls it) must spawn its two ranks itself -- before anything
    touches a GPU -- instead of asking for torch.distributed.run.  On this GPU-less host every rank then refuses
    loudly (there is no CPU path to bench); what is checked is that the ranks exist and that the parent relays
    their exit."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("GPU present: the real thing is scripts/gpu_r2.sh's gloo rehearsal")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    err = r.stdout + r.stderr
    assert "launch N>1 with" not in err                      # round 1's refusal
    assert err.count("needs an MI355X") >= 2 or "ChildFailedError" in err, err[-2000:]
