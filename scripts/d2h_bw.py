#!/usr/bin/env python3
"""Raw device -> pinned-host copy rate of this box (the ceiling of a dense list's way out): one copy, slices, streams;
alone and while a scan-like read kernel keeps HBM busy."""
import time
import torch

n = 3 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
busy = torch.empty(8 << 30, dtype=torch.uint8, device="cuda:0")
for streams, slices in ((1, 1), (1, 8), (2, 8), (2, 2), (4, 4)):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    part = n // slices
    for load in (False, True):
        torch.cuda.synchronize()
        best = 0
        for rep in range(4):
            t0 = time.perf_counter()
            if load:
                for _ in range(6):
                    busy.sum()  # an HBM-bound read on the default stream beside the copies
            for i in range(slices):
                with torch.cuda.stream(ss[i % streams]):
                    h[i * part:(i + 1) * part].copy_(d[i * part:(i + 1) * part], non_blocking=True)
            for s in ss:
                s.synchronize()
            dt = time.perf_counter() - t0
            best = max(best, n / dt / 1e9)
        print(f"D2H pinned 3 GiB, {streams} stream(s), {slices} slice(s){', HBM busy' if load else ''}: {best:.1f} GB/s", flush=True)
