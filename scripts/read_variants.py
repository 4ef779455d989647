#!/usr/bin/env python3
"""Which raw access pattern reads HBM fastest?  (diagnostic kernels only)"""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "x-search_amd"))
import torch  # noqa: E402
import xsg  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
lib = xsg.load()
lib.xsg_diag_read_variant.restype = C.c_int
lib.xsg_diag_read_variant.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
n = int(gib * 2**30)
t = torch.empty(n, dtype=torch.uint8, device="cuda:0")
t.random_(32, 127)
ctx = xsg.Context(0)
ctx.set_pattern(b"Sherlock")
sh = xsg.Shard(ctx, t.data_ptr(), n, xsg.make_chunks([0], [n - 4096]))
names = {0: "plain", 1: "nontemporal", 2: "xcd-contiguous", 3: "wave-interleaved"}
for rnd in range(2):
    for tile in (4096, 8192, 16384, 32768):
        for var in (0, 1, 2, 3):
            ms, nb = C.c_float(0), C.c_uint64(0)
            rc = lib.xsg_diag_read_variant(sh.h, tile, var, 5, C.byref(ms), C.byref(nb))
            assert rc == 0, lib.xsg_last_error()
            print(json.dumps({"round": rnd, "tile": tile, "variant": names[var], "ms": round(ms.value, 3),
                              "gbs": round(nb.value / ms.value / 1e6, 1)}), flush=True)
