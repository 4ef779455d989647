#!/usr/bin/env python3
"""Long differential fuzz run of the HIP path against the oracle (the same generators as
tests/test_gpu_fuzz.py and, every third seed each, the class-sequence and the variable-length rounds of
tests/test_gpu_regex.py; many more seeds).  Runs for --minutes, prints a progress line
every few seeds, stops at the first difference (the assertion message reproduces it).
Result summary -> gpurun_out/fuzz_campaign.json."""
import os
os.environ.setdefault("XSG_TEST_HOOKS", "1")  # this campaign switches the list routes between seeds
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for d in ("", "x-search_amd", "oracle", "tests"):
    sys.path.insert(0, str(ROOT / d))
import xs_oracle  # noqa: E402
from gpu_util import GpuSearch  # noqa: E402
from test_gpu_fuzz import first_call_rounds, fuzz_rounds  # noqa: E402
from test_gpu_regex import regex_rounds, rx_rounds  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=8.0)
ap.add_argument("--first-seed", type=int, default=100)
ap.add_argument("--hot", type=int, default=-1, help="pin the hot filter of the window kinds (0 window filter, 1 aligned trigger); -1 = alternate per seed")
a = ap.parse_args()
oracle = xs_oracle.Oracle()
searchers = {0: GpuSearch(hot=0), 1: GpuSearch(hot=1), 2: GpuSearch(probe=True)}
t0 = time.monotonic()
seed = a.first_seed
out = ROOT / "gpurun_out" / "fuzz_campaign.json"
out.parent.mkdir(exist_ok=True)
status = {"first_seed": a.first_seed, "seeds_done": 0, "failed": None, "hot": a.hot}
while time.monotonic() - t0 < a.minutes * 60:
    gs = searchers[a.hot if a.hot >= 0 else seed % 3]
    # the list tags' routes (x-search_amd/csrc/xsg_api.cpp): one-sync (default), exact only, tiny capacities (overflow -> exact)
    import os
    for k in ("XSG_LIST_FAST", "XSG_LIST_CAP"):
        os.environ.pop(k, None)
    if seed % 5 == 0:
        os.environ["XSG_LIST_CAP"] = str(1 + seed % 7)
    elif seed % 5 == 1:
        os.environ["XSG_LIST_FAST"] = "0"
    try:
        fuzz_rounds(seed, oracle, gs, rounds=14, max_chunk=60000 if seed % 4 else 600000)
        if seed % 4 == 2:  # every search the FIRST call of a fresh context and binding, one random tag per case
            first_call_rounds(seed, oracle, rounds=4)
        if seed % 3 == 0:
            regex_rounds(seed, oracle, gs, rounds=10)
        if seed % 3 == 1:  # variable-length expressions: the automaton route, prefilter on (default) / off per seed
            import os
            os.environ["XSG_RX_PRE"] = os.environ["XSG_RX_FAC"] = "0" if seed % 2 else "1"  # both prefilters forced on / off
            rx_rounds(seed, oracle, gs, rounds=6)
            os.environ.pop("XSG_RX_PRE")
            os.environ.pop("XSG_RX_FAC")
    except AssertionError as e:
        status["failed"] = {"seed": seed, "message": str(e)[:2000]}
        print("FAIL", seed, str(e)[:2000], flush=True)
        break
    seed += 1
    status["seeds_done"] = seed - a.first_seed
    status["elapsed_s"] = round(time.monotonic() - t0, 1)
    if (seed - a.first_seed) % 5 == 0:
        print(json.dumps(status), flush=True)
        out.write_text(json.dumps(status) + "\n")
out.write_text(json.dumps(status) + "\n")
print(json.dumps(status), flush=True)
sys.exit(1 if status["failed"] else 0)
