#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the rocprofv3 --pmc passes of scripts/gpu_profiles.sh (FETCH_SIZE and WRITE_SIZE,
collected separately, one pair of passes per kernel instantiation): HBM bytes per k_scan launch, with the gfx950
correction of MI355X_MICROARCH.md's HBM section (FETCH_SIZE counts a 128-byte request as 64 bytes -> read bytes =
2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact).
Usage: pmc_traffic.py <BENCH json> <tag> <fetch csv> <write csv> [<fetch csv> <write csv> ...]"""
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
    bench_json, tag = sys.argv[1:3]
    pairs = sys.argv[3:]
    bench = json.loads(Path(bench_json).read_text().splitlines()[-1])
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    by = {}
    for fetch_csv, write_csv in zip(pairs[0::2], pairs[1::2]):
        fetch_kb, nf, names = avg_counter(fetch_csv, "FETCH_SIZE")
        write_kb, nw, _ = avg_counter(write_csv, "WRITE_SIZE")
        if len(names) != 1:
            raise SystemExit(f"{fetch_csv}: full-shard launches of more than one instantiation: {names}")
        rd, wr = 2.0 * fetch_kb * 1024.0, write_kb * 1024.0
        key = names[0].replace("void ", "").split("(")[0]  # "xsg::k_scan<3, false, false, false, 4, false, true>"
        by[key] = {"FETCH_SIZE_KB_avg": fetch_kb, "WRITE_SIZE_KB_avg": write_kb, "launches_averaged": [nf, nw],
                   "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                   "ratio_traffic_over_algorithmic": (rd + wr) / alg}
    out = {
        "timed_kernel": bench["roofline"]["kernel"],
        "pattern": bench["config"]["pattern"],
        "config": bench["config"]["workload"],
        "bytes_per_gpu": bench["config"]["bytes_per_gpu"],
        "algorithmic_bytes_per_launch": alg,
        "correction": "HBM read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-B requests as 64 B, "
                      "MI355X_MICROARCH.md HBM section); WRITE_SIZE x 1024 exact",
        "by_kernel": by,
        "collected": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes per instantiation, pinned with "
                     f"XSG_HOT / XSG_TUNE at the timed run's stagger (scripts/gpu_profiles.sh {tag}); written by this script, "
                     f"nothing edited by hand",
    }
    (ROOT / "profiles" / f"{tag}_pmc_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
