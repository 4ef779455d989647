#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two rocprofv3 --pmc passes of scripts/gpu_bench.sh
(FETCH_SIZE and WRITE_SIZE, collected separately): HBM bytes per k_scan launch, with the
gfx950 correction of MI355X_MICROARCH.md's HBM section (FETCH_SIZE counts a 128-byte
request as 64 bytes -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact).
Usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <BENCH json> [tag]"""
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def avg_counter(path, counter):
    """average over the FULL-SHARD k_scan launches only (largest grid): the library's hot-filter probe also launches
    k_scan, on a 2 GiB prefix of the shard"""
    rows = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and "k_scan<" in row["Kernel_Name"]:
                rows.append((int(row["Grid_Size"]), float(row["Counter_Value"]), row["Kernel_Name"]))
    if not rows:
        raise SystemExit(f"no {counter} rows for k_scan in {path}")
    full = max(g for g, _, _ in rows)
    vals = [v for g, v, _ in rows if g == full]
    names = sorted({n for g, _, n in rows if g == full})
    return sum(vals) / len(vals), len(vals), names


def main():
    fetch_csv, write_csv, bench_json = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else ""
    bench = json.loads(Path(bench_json).read_text().splitlines()[-1])
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    fetch_kb, nf, names = avg_counter(fetch_csv, "FETCH_SIZE")
    write_kb, nw, _ = avg_counter(write_csv, "WRITE_SIZE")
    rd, wr = 2.0 * fetch_kb * 1024.0, write_kb * 1024.0
    out = {
        "kernel": bench["roofline"]["kernel"],
        "kernels_profiled": names,
        "pattern": bench["config"]["pattern"],
        "config": bench["config"]["workload"],
        "bytes_per_gpu": bench["config"]["bytes_per_gpu"],
        "FETCH_SIZE_KB_avg": fetch_kb, "WRITE_SIZE_KB_avg": write_kb, "launches_averaged": [nf, nw],
        "correction": "HBM read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-B requests as 64 B, "
                      "MI355X_MICROARCH.md HBM section); WRITE_SIZE x 1024 exact",
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": alg, "ratio_traffic_over_algorithmic": (rd + wr) / alg,
        "collected": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (scripts/gpu_profiles.sh {tag})",
    }
    name = f"{tag}_pmc_traffic.json" if tag else "pmc_traffic.json"
    (ROOT / "profiles" / name).write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
