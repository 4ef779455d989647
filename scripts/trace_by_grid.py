#!/usr/bin/env python3
"""rocprofv3 --kernel-trace csv -> per (kernel, grid size): calls, average / min / max ms.

`--stats` averages every launch of a kernel name; the library's hot-filter probe launches the timed instantiation on a
2 GiB prefix of the shard (4 launches of 0.3 ms), which pulls that average below the full-size launches' -- this splits them.
usage: trace_by_grid.py <kernel_trace.csv> [name substring]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(list)
sub = sys.argv[2] if len(sys.argv) > 2 else "k_scan<"
for r in csv.DictReader(open(sys.argv[1])):
    if sub in r["Kernel_Name"]:
        g = r.get("Grid_Size_X") or r.get("Grid_Size")
        acc[(r["Kernel_Name"].replace("void ", "").split("(")[0], int(g))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel, grid (work-items), calls, average ms, min ms, max ms")
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k}, {g}, {len(v)}, {sum(v) / len(v) / 1e6:.4f}, {min(v) / 1e6:.4f}, {max(v) / 1e6:.4f}")
