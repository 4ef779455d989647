#!/usr/bin/env python3
"""Dense-needle list search (1 match per ~140 bytes) for a kernel-level profile."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import xsg  # noqa: E402
from test_gpu_fullsize import build_shard  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
t, blocks, plan, chunks, goffs, cap = build_shard(gib)
ctx = xsg.Context(0)
sh = xsg.Shard(ctx, t.data_ptr(), cap, chunks)
ctx.set_pattern(b"She")
for mode, name in ((xsg.MATCH_BYTE_OFFSETS, "match_byte_offsets"), (xsg.LINE_BYTE_OFFSETS, "line_byte_offsets"),
                   (xsg.LINE_INDICES, "line_indices")):
    for rep in range(3):
        t0 = time.perf_counter()
        r = sh.search_u64(mode)
        print(name, rep, len(r), round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
