#!/usr/bin/env python3
"""Where do hipHostMalloc'ed buffers of different sizes live (NUMA node), and how fast does a device-to-host copy into
each run?  (A 2 GB pinned mirror took a D2H copy at 37 GB/s where a 0.5 GB one took 55: profiles/r04_dense_timeline.txt.)"""
import ctypes as C
import re
import time

import torch

hip = C.CDLL("libamdhip64.so")
torch.cuda.init()
dev = torch.empty(3 << 30, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()


def numa_of(addr):
    want = f"{addr:x}"
    for line in open("/proc/self/numa_maps"):
        if line.startswith(want):
            return " ".join(re.findall(r"N\d+=\d+|kernelpagesize_kB=\d+|bind:\S+|interleave:\S+|default|prefer\S*", line))
    return "?"


def run(nbytes, flags, label):
    p = C.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipHostMalloc(C.byref(p), C.c_size_t(nbytes), C.c_uint(flags))
    t_alloc = time.perf_counter() - t0
    if rc:
        print(label, "hipHostMalloc failed", rc)
        return
    best = 0
    for rep in range(4):
        t0 = time.perf_counter()
        hip.hipMemcpy(p, C.c_void_p(dev.data_ptr()), C.c_size_t(nbytes), C.c_int(2))  # D2H
        best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
    print(f"{label}: {nbytes / 2**30:.2f} GiB, alloc {t_alloc * 1e3:.0f} ms, D2H {best:.1f} GB/s, numa_maps: {numa_of(p.value)}", flush=True)
    hip.hipHostFree(p)


print(open("/proc/self/status").read().split("Cpus_allowed_list:")[1].split("\n")[0].strip(), "<- cpus allowed;",
      open("/proc/self/status").read().split("Mems_allowed_list:")[1].split("\n")[0].strip(), "<- mems allowed")
for n in (256 << 20, 512 << 20, 1 << 30, 2 << 30, 3 << 30):
    run(n, 0, "default      ")
for n in (2 << 30,):
    run(n, 0x20000000, "numa_user    ")   # hipHostMallocNumaUser
    run(n, 0x2, "mapped       ")           # hipHostMallocMapped
    run(n, 0x40000000, "coherent     ")    # hipHostMallocCoherent
    run(n, 0x80000000, "non_coherent ")    # hipHostMallocNonCoherent
