#!/usr/bin/env python3
"""Burst-shape experiments with a pure nt read kernel (libxsg_diag.so): loads per lane, workgroup
size, wave stagger, pause between a wave's loads."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "x-search_amd"))
import torch  # noqa: E402
import xsg_diag  # noqa: E402

gib = 48.0
n = int(gib * 2**30)
t = torch.empty(n, dtype=torch.uint8, device="cuda:0")
t.random_(32, 127)
sink = torch.zeros(4, dtype=torch.int32, device="cuda:0")
cases = []
for loads, block in ((4, 256), (8, 256), (2, 256), (4, 512), (4, 128), (2, 512), (8, 128), (4, 1024), (1, 1024)):
    for stagger in (0, 6, 12, 18, 24):
        cases.append((loads, block, stagger, 0))
for gap in (2, 4, 8):
    cases.append((4, 256, 0, gap))
    cases.append((4, 256, 12, gap))
for rnd in range(2):
    for loads, block, stagger, gap in cases:
        ms, nb = xsg_diag.read_exp(t.data_ptr(), n, sink.data_ptr(), loads, block, stagger, gap, iters=4)
        print(json.dumps({"round": rnd, "loads": loads, "block": block, "stagger": stagger, "gap": gap, "ms": round(ms, 3),
                          "gbs": round(nb / ms / 1e6, 1)}), flush=True)
