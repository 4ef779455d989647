#!/usr/bin/env python3
"""Which raw access pattern reads HBM fastest?  (libxsg_diag.so: diagnostic kernels, not the product library)"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "x-search_amd"))
import torch  # noqa: E402
import xsg_diag  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
n = int(gib * 2**30)
t = torch.empty(n, dtype=torch.uint8, device="cuda:0")
t.random_(32, 127)
sink = torch.zeros(4, dtype=torch.int32, device="cuda:0")
names = {0: "plain", 1: "nontemporal", 2: "xcd-contiguous", 3: "wave-interleaved", 4: "sc1 nt", 5: "sc0 sc1 nt", 6: "sc0 nt", 7: "buffer sc1 nt", 8: "buffer sc0 sc1 nt"}
for rnd in range(2):
    for tile in (16384,):
        for var in (0, 1, 4, 5, 7, 8):
            ms, nb = xsg_diag.read(t.data_ptr(), n, sink.data_ptr(), tile_bytes=tile, variant=var, iters=5)
            print(json.dumps({"round": rnd, "tile": tile, "variant": names[var], "ms": round(ms, 3),
                              "gbs": round(nb / ms / 1e6, 1)}), flush=True)
