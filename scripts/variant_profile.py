#!/usr/bin/env python3
"""One k_scan variant on a device-resident shard, a few launches: the unit that scripts/gpu_variants.sh runs under
rocprofv3 (kernel trace / SQ counters / FETCH_SIZE), once per variant, so that every variant -- not only the plain
count the bench times -- has tracked evidence under profiles/.  Prints one JSON line (HIP-event timing)."""
import os
os.environ.setdefault("XSG_TEST_HOOKS", "1")  # this script switches XSG_* toggles between searches (read once per process otherwise)
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import bench  # noqa: E402

CASES = {
    # name: (pattern, flags, mode)
    "count_Sherlock": ("Sherlock", 0, "count"),
    "count_nl_Sherlock": ("Sherlock", 0, "count+nl"),
    "lines_Sherlock": ("Sherlock", 0, "count_lines"),
    "icase_Sherlock": ("Sherlock", "icase", "count"),
    "icase_that": ("that", "icase", "count"),
    "mask1_e": ("e", 0, "count"),
    "mask1_the": ("the", 0, "count"),
    "lines_e": ("e", 0, "count_lines"),
    "lines_the": ("the", 0, "count_lines"),
    "lines_She": ("She", 0, "count_lines"),
    "icase_lines_the": ("the", "icase", "count_lines"),
    "icase_the": ("the", "icase", "count"),
    "icase_Holmes": ("holmes", "icase", "count"),
    "one_that": ("that", 0, "count"),
    "mask2_Holmes": ("Holmes", 0, "count"),
    "mask2_Sherl": ("Sherl", 0, "count"),
    "lines_Holmes": ("Holmes", 0, "count_lines"),
    "lines_Sherl": ("Sherl", 0, "count_lines"),
    "lines_that": ("that", 0, "count_lines"),
    "two_detectiv": ("detectiv", 0, "count"),
    "lines_detectiv": ("detectiv", 0, "count_lines"),
    "mask2_detecti": ("detecti", 0, "count"),
    "lines_detecti": ("detecti", 0, "count_lines"),
    "long_Sherlock_Holmes": ("Sherlock Holmes", 0, "count"),
    "long_detective_street": ("detective street", 0, "count"),
    "class_She_r_lock": ("She[r ]lock", "regex", "count"),
    "class_Ss_herlock": ("[Ss]herlock", "regex", "count"),
    "class_digits": ("[0-9]{4}-[0-9]{2}", "regex", "count"),
    "class_The_az3": ("[Tt]he [a-z]{3} ", "regex", "count"),
    # the automaton route (k_rx_scan): no trigger in the text, rare triggers, a match in most lines, a trigger at
    # every word (no skipping), and the matching-lines count
    "rx_none": ("zzz+", "regex", "count"),
    "rx_alt": ("Sherlock|Holmes", "regex", "count"),
    "rx_dotstar": ("Sher.*mes", "regex", "count"),
    "rx_optional": ("lock(ed|s)?", "regex", "count"),
    "rx_word": ("\\w+ing", "regex", "count"),
    "rx_lines_alt": ("Sherlock|Holmes", "regex", "count_lines"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="count_Sherlock", help="one case, a comma-separated list (one shard, one process), or 'all'")
    ap.add_argument("--gib", type=float, default=20.0)
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--warm", type=int, default=20, help="untimed launches first: the clocks of an idle card ramp for several ms, and the\n                    VALU-heavier variants feel it (5 timed launches right after the shard fill read 3-7 %% low)")
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--lexicon", choices=["bench", "plain", "nosh"], default="bench",
                    help="plain: the bench lexicon without the words that are pieces of the needle (She, lock, locked, Sher)")
    a = ap.parse_args()
    import torch
    import corpus
    import xsg
    args = argparse.Namespace(chunk_mib=16, templates=32, seed=0x5EED,
                              lexicon={"plain": corpus.LEXICON_PLAIN, "nosh": corpus.LEXICON_NOSH}.get(a.lexicon))
    blocks = bench.template_blocks(args, b"Sherlock")
    tbytes = np.array([b.size for b in blocks], dtype=np.int64)
    nchunks = int(round(a.gib * 2**30 / (16 << 20)))
    plan = bench.chunk_plan(args, 0, nchunks)
    off, ln, cap = corpus.chunk_table(tbytes[plan])
    nbytes = int(ln.sum())
    dev = torch.device("cuda", 0)
    shard_t = torch.empty(cap, dtype=torch.uint8, device=dev)
    dts = [torch.from_numpy(b).to(dev) for b in blocks]
    for c in range(nchunks):
        o = int(off[c])
        shard_t[o:o + dts[int(plan[c])].numel()].copy_(dts[int(plan[c])])
    torch.cuda.synchronize()
    del dts
    ctx = xsg.Context(0)
    sh = xsg.Shard(ctx, shard_t.data_ptr(), cap, xsg.make_chunks(off, ln))
    names = list(CASES) if a.case == "all" else a.case.split(",")
    modes = {"count": xsg.COUNT_MATCHES, "count_lines": xsg.COUNT_LINES, "count+nl": xsg.COUNT_MATCHES | xsg.WITH_NEWLINES}
    import os
    for name in names:
        pat, fl, mode = CASES[name]
        # the rx_ cases are about k_rx_scan itself: keep the prefilter route (which the synchronous calls would take for
        # expressions with a selective start) out of the kernel name and the timing
        os.environ["XSG_RX_PRE"] = "0" if name.startswith("rx_") else "1"
        flags = {0: 0, "icase": xsg.FLAG_IGNORE_CASE, "regex": xsg.FLAG_REGEX}[fl]
        ctx.set_pattern(pat.encode(), flags)
        st = sh.tune(modes[mode]) if a.tune else None
        # the count first: the timed launches are then what every call after the first one launches (a needle the first
        # count found dense runs with a smaller wave stagger, x-search_amd/csrc/xsg_api.cpp: density_serial)
        cm = int(sh.count(modes[mode])[xsg.CTR_LINES if mode == "count_lines" else xsg.CTR_MATCHES])
        if a.warm > 0:
            sh.time_scan_kernel(modes[mode], a.warm)
        ms = sh.time_scan_kernel(modes[mode], a.iters)
        print(json.dumps({"case": name, "pattern": pat, "flags": fl, "mode": mode, "gib": a.gib, "bytes": nbytes, "lexicon": a.lexicon,
                          "result": cm, "bytes_per_result": round(nbytes / max(cm, 1), 1),
                          "kernel": sh.scan_kernel_name(modes[mode]), "tuned_stagger": st, "ms": round(ms, 4),
                          "tb_s": round(nbytes / ms / 1e9, 3), "frac_of_8tbs": round(nbytes / ms / 1e9 / 8.0, 4)}), flush=True)


if __name__ == "__main__":
    main()
