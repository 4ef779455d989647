#!/usr/bin/env python3
"""Raw pinned-host -> device copy rate of this box (the ceiling of the end-to-end pipeline)."""
import time, torch
n = 1 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
for streams in (1, 2, 4):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    part = n // streams
    torch.cuda.synchronize()
    best = 0
    for rep in range(5):
        t0 = time.perf_counter()
        for i, s in enumerate(ss):
            with torch.cuda.stream(s):
                d[i * part:(i + 1) * part].copy_(h[i * part:(i + 1) * part], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = max(best, n / dt / 2**30)
    print(f"H2D pinned, {streams} stream(s): {best:.1f} GiB/s")
