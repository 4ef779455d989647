#!/usr/bin/env python3
"""Kernels and memory copies of a rocprofv3 run on one time axis: the LAST burst of activity that ends with the given kernel
(default k_line_gather), everything shorter than --min-us folded away.  usage: timeline.py <dir> [--last k_line_gather]"""
import argparse
import csv
import glob

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--last", default="k_line_gather")
ap.add_argument("--window-ms", type=float, default=120.0)
ap.add_argument("--min-us", type=float, default=200.0)
a = ap.parse_args()
ev = []
for f in glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].replace("void ", "").split("(")[0][:60], ""))
for f in glob.glob(a.dir + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy")), r.get("Bytes", "")))
ev.sort()
ends = [e for e in ev if a.last in e[2]]
if not ends:
    raise SystemExit("no such kernel in the trace")
t_end = max(e[1] for e in ev if e[0] < ends[-1][1] + 100_000_000 and e[0] >= ends[-1][0] - 1)  # copies behind the last gather too
t_end = max(e[1] for e in ev if ends[-1][0] - a.window_ms * 1e6 <= e[0] <= ends[-1][1] + 60e6)
t0 = t_end - a.window_ms * 1e6
win = [e for e in ev if e[0] >= t0 and e[1] <= t_end + 1]
if win:
    base = win[0][0]
    print(f"# window of {a.window_ms} ms before the end of the last activity; events >= {a.min_us} us (others summed)")
    small = 0.0
    for s, e, name, extra in win:
        d = (e - s) / 1e3
        if d < a.min_us:
            small += d
            continue
        gb = f"  {int(extra) / 1e6:9.1f} MB  {int(extra) / max(e - s, 1):6.1f} GB/s" if extra not in ("", None) and str(extra).isdigit() else ""
        print(f"{(s - base) / 1e6:9.3f} ms  +{d / 1e3:8.3f} ms  {name}{gb}")
    print(f"# events under {a.min_us} us: {small / 1e3:.3f} ms in all")
