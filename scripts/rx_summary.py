#!/usr/bin/env python3
"""gpurun_out/rx_sweep_{pre,nopre}.jsonl (scripts/rx_sweep.py, default and XSG_RX_PRE=0 XSG_RX_FAC=0) + gpurun_out/rx_pmc/
(scripts/gpu_rx_pmc.sh) -> the tables committed as profiles/r02_rx_sweep.txt and profiles/r02_rx_pmc.txt."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
out = ROOT / "gpurun_out"


def load(name):
    p = out / name
    return [json.loads(l) for l in p.read_text().splitlines() if l.startswith("{")] if p.exists() else []


pre, nopre = load("rx_sweep_pre.jsonl"), load("rx_sweep_nopre.jsonl")
lines = ["# The regex row on one MI355X, 8 GiB of the bench corpus resident in HBM (scripts/rx_sweep.py).",
         "# kernel: HIP-event time of the bulk kernel alone (xsg_time_scan_kernel: k_scan for literals and class sequences,",
         "#   k_rx_scan for the automaton route); call: a whole synchronous xsg_count() incl. finish kernel and host sync.",
         "# default = the library's own routing (prefilter route where the expression starts selectively and the shard is",
         "#   >= 512 MiB; a count whose candidates turn out dense falls back to k_rx_scan and remembers it);",
         "# XSG_RX_PRE=0 XSG_RX_FAC=0 = k_rx_scan on every tile, for every expression of the automaton route.  frac = of 8 TB/s.",
         "# first call: the first xsg_count() on a fresh binding (hot-filter probe, factor prefilter's tile marks included);",
         "#   later calls of an expression with a factor prefilter reuse the marks and do not read the whole shard again:",
         "#   their time is given, a rate is not.",
         "pattern | mode | matches | kernel GB/s | first call ms | GB/s | later calls ms | GB/s | frac | no prefilters: later calls ms | GB/s | default route"]
for a, b in zip(pre, nopre):
    ka = a.get("kernel_after", a["kernel"])
    marked = "factor prefilter" in ka
    route = ("prefilter (k_scan<kClass> + k_rx_verify ...)" if "prefilter route" in ka else
             "k_rx_scan on the tiles the factor prefilter marked" if marked else ka.split(" stagger")[0].split(" states")[0])
    later = f"{a['count_call_ms']:.2f} | - | -" if marked else f"{a['count_call_ms']:.2f} | {a['count_call_gbs']:.0f} | {a['count_call_gbs'] / 8000:.3f}"
    lines.append(f"{a['pattern']} | {a['mode']} | {a['result']} | {b['gbs']:.0f} | {a.get('first_call_ms', 0):.2f} | {a.get('first_call_gbs', 0):.0f} | "
                 f"{later} | {b['count_call_ms']:.2f} | {b['count_call_gbs']:.0f} | {route}")
(ROOT / "profiles" / "r02_rx_sweep.txt").write_text("\n".join(lines) + "\n")
pm = out / "rx_pmc" / "summary.txt"
if pm.exists():
    body = [l for l in pm.read_text().splitlines() if l.startswith("{") or "LDS pass" in l]
    head = ["# k_rx_scan under rocprofv3 on one MI355X, 8 GiB shard (scripts/gpu_rx_pmc.sh; XSG_RX_PRE=0 so that the kernel is",
            "# what the timed launches run): --kernel-trace --stats; --pmc SQ_* (per wave = per 4 KiB of text); --pmc FETCH_SIZE",
            "# (x 2 x 1024 on gfx950); a second SQ pass with the LDS counters.  Separate runs, no trace flags next to --pmc.",
            "# cases: rx_none `zzz+` (no trigger byte in the text: the staging phase alone), rx_alt `Sherlock|Holmes`,",
            "# rx_dotstar `Sher.*mes`, rx_word `\\w+ing` (a trigger at every word: no skipping)"]
    (ROOT / "profiles" / "r02_rx_pmc.txt").write_text("\n".join(head + body) + "\n")
print("\n".join(lines[9:]))
