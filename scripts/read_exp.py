#!/usr/bin/env python3
"""Burst-shape experiments with a pure nt read kernel: loads per lane, workgroup
size, wave stagger, pause between a wave's loads."""
import ctypes as C, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "x-search_amd"))
import torch, xsg  # noqa: E402
gib = 48.0
lib = xsg.load()
lib.xsg_diag_read_exp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
n = int(gib * 2**30)
t = torch.empty(n, dtype=torch.uint8, device="cuda:0"); t.random_(32, 127)
ctx = xsg.Context(0); ctx.set_pattern(b"Sherlock")
sh = xsg.Shard(ctx, t.data_ptr(), n, xsg.make_chunks([0], [n - 4096]))
cases = []
for loads, block in ((4, 256), (8, 256), (2, 256), (4, 512), (4, 128), (2, 512), (8, 128), (4, 1024), (1, 1024)):
    for stagger in (0, 6, 12, 18, 24):
        cases.append((loads, block, stagger, 0))
for gap in (2, 4, 8):
    cases.append((4, 256, 0, gap))
    cases.append((4, 256, 12, gap))
for rnd in range(2):
    for loads, block, stagger, gap in cases:
        ms, nb = C.c_float(0), C.c_uint64(0)
        rc = lib.xsg_diag_read_exp(sh.h, loads, block, stagger, gap, 4, C.byref(ms), C.byref(nb))
        assert rc == 0, lib.xsg_last_error()
        print(json.dumps({"round": rnd, "loads": loads, "block": block, "stagger": stagger, "gap": gap, "ms": round(ms.value, 3),
                          "gbs": round(nb.value / ms.value / 1e6, 1)}), flush=True)
