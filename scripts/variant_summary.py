#!/usr/bin/env python3
"""Condenses gpurun_out/variants_<tag>/ (scripts/gpu_variants.sh) into the table that is committed under profiles/:
per variant the rocprofv3 kernel-stats duration, the SQ instruction mix per wave (= per 4 KiB) and the HBM read
traffic per launch against the algorithmic bytes (FETCH_SIZE x 2 x 1024 on gfx950, MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
sweep = {}
for name in ("sweep.jsonl", "sweep_tuned.jsonl"):
    f = root / name
    if f.exists():
        for line in f.read_text().splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                sweep.setdefault(d["case"], {})[name] = d
print("# HIP-event sweep, 50 GiB shard (scripts/variant_profile.py --case all [--tune])")
print("case | pattern | mode | kernel | TB/s (default stagger) | TB/s (xsg_shard_tune) | frac of 8 TB/s (best)")
for c, d in sweep.items():
    a = d.get("sweep.jsonl")
    b = d.get("sweep_tuned.jsonl")
    best = max(x["tb_s"] for x in (a, b) if x)
    print(f"{c} | {a['pattern']} | {a['mode']} | {a['kernel']} | {a['tb_s']} | "
          f"{b['tb_s'] if b else '-'} (stagger {b['tuned_stagger'] if b else '-'}) | {best / 8.0:.3f}")
print()
print("# rocprofv3 passes, 20 GiB shard, per variant (kernel-trace --stats; --pmc SQ_*; --pmc FETCH_SIZE: separate runs)")
for d in sorted(p for p in root.iterdir() if p.is_dir()):
    case = d.name
    line = {"case": case}
    want = None  # the instantiation this case's timed launches ran (the library's hot-filter probe launches others)
    log = d / "stats.log"
    if log.exists():
        for l in log.read_text().splitlines():
            if l.startswith("{"):
                j = json.loads(l)
                line["bytes"] = j["bytes"]
                line["hip_event_ms"] = j["ms"]
                want = j["kernel"].split(" stagger")[0].split(" states")[0]
                line["kernel"] = j["kernel"]
    if want is None:
        continue
    full = "void " + want + "("  # k_scan(ScanArgs), k_rx_scan(ScanArgs, unsigned int)
    def newest(pattern):  # gpurun_out/ is merged across calls: an older run's files may still lie next to the new ones
        fs = glob.glob(pattern)
        return [max(fs, key=os.path.getmtime)] if fs else []
    for f in newest(str(d / "stats" / "*" / "*kernel_trace.csv")):
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"])) for r in csv.DictReader(open(f))
                if r["Kernel_Name"].startswith(full)]
        if durs:
            g = max(x[1] for x in durs)
            ds = [x[0] for x in durs if x[1] == g]  # full-shard launches only
            line["calls"] = len(ds)
            line["avg_ms"] = round(sum(ds) / len(ds) / 1e6, 4)
            line["min_ms"] = round(min(ds) / 1e6, 4)
    acc = defaultdict(list)
    for f in newest(str(d / "sq" / "*" / "*counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(full)]
        if rows:
            g = max(int(r["Grid_Size"]) for r in rows)
            for r in rows:
                if int(r["Grid_Size"]) == g:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc.get("SQ_WAVES"):
        waves = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
        for k, v in acc.items():
            if k != "SQ_WAVES":
                line[k + "_per_wave"] = round(sum(v) / len(v) / waves, 1)
        if "SQ_WAIT_ANY" in acc and "SQ_WAVE_CYCLES" in acc:
            line["wait_share"] = round(sum(acc["SQ_WAIT_ANY"]) / sum(acc["SQ_WAVE_CYCLES"]), 3)
    fs = []
    for f in newest(str(d / "fetch" / "*" / "*counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(full) and r["Counter_Name"] == "FETCH_SIZE"]
        if rows:
            g = max(int(r["Grid_Size"]) for r in rows)
            fs += [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == g]
    if fs and "bytes" in line:
        line["hbm_read_bytes_per_launch"] = 2.0 * 1024.0 * sum(fs) / len(fs)
        line["read_traffic_over_algorithmic"] = round(line["hbm_read_bytes_per_launch"] / line["bytes"], 4)
    if "avg_ms" in line and "bytes" in line:
        line["tb_s_rocprof_avg"] = round(line["bytes"] / line["avg_ms"] / 1e9, 3)
    print(json.dumps(line))
