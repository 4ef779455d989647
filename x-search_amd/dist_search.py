"""Multi-GPU file search: one process per GPU (torch.distributed; backend "nccl"
is RCCL on ROCm), contiguous chunk ranges per rank, and a collective only where
the path has a real exchange step:

  xs::count / xs::count_lines   one all_reduce(sum) of a small int64 vector
  xs::line_indices (no metafile) one all_gather of per-rank newline totals ->
                                 exclusive prefix -> this rank's line-index base
  offsets / lines                no collective: every rank keeps its own part
                                 (already globally ordered by concatenation)

Corpus bytes never move between GPUs.

`scan_range` is the per-rank scanner; the default runs the HIP pipeline
(xsg.Job) on this rank's GPU.  tests/test_dist_gloo.py injects a CPU stand-in to
exercise the sharding + collectives on gloo without a GPU.
"""
from __future__ import annotations

import numpy as np

import xsg


def chunk_range(nchunks: int, world: int, rank: int) -> tuple[int, int]:
    """GPU g gets chunks [g*C/G, (g+1)*C/G): contiguous, so results concatenate in file order."""
    return rank * nchunks // world, (rank + 1) * nchunks // world


def file_plan(path: str, meta_path: str | None = None, chunk_bytes: int = 16 << 20) -> np.ndarray:
    if meta_path:
        return xsg.meta_read(meta_path)[1]
    return xsg.plan_chunks(path, chunk_bytes)


def _gpu_scan_range(pattern, path, mode, meta_path, lo, hi, device, num_threads, chunk_bytes):
    """-> (result, newlines_in_range)"""
    j = xsg.Job(pattern, path, mode, meta_path=meta_path, device=device, num_threads=num_threads,
                num_max_readers=num_threads, chunk_bytes=chunk_bytes, chunk_range=(lo, hi))
    try:
        res = j.result()
        return res, j.stats()["newlines"]
    finally:
        j.close()


class LibraryCollective:
    """The exchange step through the library's own RCCL communicator (include/xsg.h "Multi-GPU", rank form) instead
    of torch.distributed's: rank 0 makes the id, `dist` (any backend) only carries those 128 bytes to the others."""

    def __init__(self, dist, device: int):
        import torch
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
        box = [xsg.comm_unique_id() if rank == 0 else None]
        if dist is not None and world > 1:
            dist.broadcast_object_list(box, src=0)
        self.ctx = xsg.Context(device)
        self.comm = xsg.Comm.rank(self.ctx, world, rank, box[0])
        self.buf = torch.zeros(64, dtype=torch.int64, device=f"cuda:{device}")
        self.world, self.rank = world, rank

    def sum(self, values):
        import torch
        k = len(values)
        self.buf[:k] = torch.tensor([int(v) for v in values], dtype=torch.int64)
        torch.cuda.synchronize(self.buf.device)
        return [int(x) for x in self.comm.reduce_counts(self.buf.data_ptr(), k)]

    def gather(self, value: int):
        return [int(x) for x in self.comm.allgather_u64([int(value)])]

    def close(self):
        self.comm.close()
        self.ctx.close()


def distributed_search(pattern: bytes, path: str, mode: int, meta_path: str | None = None, *, dist=None,
                       tensor_device="cpu", device: int = 0, num_threads: int = 2, chunk_bytes: int = 16 << 20,
                       scan_range=None, collective=None):
    """Run one xs:: tag over `path` with the chunks sharded across the ranks of `dist`.

    Count tags return the global count on every rank.  List tags return this
    rank's part (global byte offsets / global line indices / lines).
    collective: a LibraryCollective to run the exchange step over the library's RCCL communicator; default:
    torch.distributed's all_reduce / all_gather on `tensor_device`.
    """
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    plan = file_plan(path, meta_path, chunk_bytes)
    lo, hi = chunk_range(len(plan), world, rank)
    if scan_range is None:
        def scan_range(lo_, hi_):
            return _gpu_scan_range(pattern, path, mode, meta_path, lo_, hi_, device, num_threads, chunk_bytes)
    if hi > lo:
        result, newlines = scan_range(lo, hi)
    else:
        result, newlines = (0 if mode in (xsg.COUNT_MATCHES, xsg.COUNT_LINES) else ([] if mode == xsg.LINES else
                                                                                    np.zeros(0, np.uint64))), 0
    if mode in (xsg.COUNT_MATCHES, xsg.COUNT_LINES):
        if collective is not None:
            total, chunks = collective.sum([int(result), hi - lo])
            assert chunks == len(plan)
            return total
        t = torch.tensor([int(result), hi - lo], dtype=torch.int64, device=tensor_device)
        if dist is not None:
            dist.all_reduce(t)  # sum; 16 bytes: latency-bound, xGMI bandwidth is irrelevant
        assert int(t[1]) == len(plan)
        return int(t[0])
    if mode == xsg.LINE_INDICES and not meta_path:
        # line-index base of this rank = newlines in all lower ranks' ranges
        if collective is not None:
            allnl = collective.gather(int(newlines))
            return np.asarray(result, dtype=np.uint64) + np.uint64(sum(allnl[:rank]))
        mine = torch.tensor([int(newlines)], dtype=torch.int64, device=tensor_device)
        if dist is not None:
            allnl = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allnl, mine)
            base = int(sum(int(x) for x in allnl[:rank]))
        else:
            base = 0
        return np.asarray(result, dtype=np.uint64) + np.uint64(base)
    return result
