#!/usr/bin/env python3
"""A throughput record on text nobody generated for the purpose: the Python sources, .txt / .rst /
.md files of the standard library and the installed packages of this image (the text tests/test_gpu_natural.py checks
parity on; ~1 GiB here and on the GPU box), cut into newline-aligned 16 MiB chunks and chunk-replicated in a seeded
order into a device-resident shard of --gib GiB, like bench.py replicates its synthetic templates.

Per case (an absent word, a rare one, a common one, a two-byte needle, ignore_case, a long pattern, ...): whole C-ABI
calls of xs::count / count_lines and the four list tags (scripts/config_times.py: timed_calls), the scan kernel the
library launches (its probe's choice of hot filter shows as the last template argument) and, bare, the kernel's own
time for the two count modes.  Counts are checked against the oracle's counts on the distinct chunks."""
import argparse
import json
import sys
import sysconfig
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle", ROOT / "scripts"):
    sys.path.insert(0, str(p))
import corpus  # noqa: E402
import xsg  # noqa: E402
from config_times import timed_calls  # noqa: E402
from xs_oracle import Oracle  # noqa: E402

CASES = [  # (name, pattern, flags)
    ("absent_word", b"Sherlock", 0),
    ("rare_word", b"Copyright", 0),
    ("common_word", b"return", 0),
    ("common_short", b"self", 0),
    ("two_bytes", b"in", 0),
    ("one_byte", b"e", 0),
    ("three_bytes", b"the", 0),
    ("icase_word", b"error", xsg.FLAG_IGNORE_CASE),
    ("long_pattern", b"raise ValueError(", 0),
    ("indent_def", b"    def __init__(self", 0),
]


def natural_text(limit: int) -> np.ndarray:
    files = []
    for key in ("stdlib", "purelib"):
        root = Path(sysconfig.get_paths()[key])
        files += sorted(p for p in root.rglob("*") if p.suffix in (".py", ".txt", ".rst", ".md", ".cfg") and p.is_file())
    parts, n = [], 0
    for p in files:
        try:
            b = p.read_bytes()
        except OSError:
            continue
        if not b:
            continue
        parts.append(b if b.endswith(b"\n") else b + b"\n")
        n += len(parts[-1])
        if n >= limit:
            break
    return np.frombuffer(b"".join(parts), dtype=np.uint8)


def cut(data: np.ndarray, target: int):
    out, at = [], 0
    while at < data.size:
        end = min(at + target, data.size)
        if end < data.size:
            nl = np.flatnonzero(data[end - 1:end + (4 << 20)] == 10)
            end = end + int(nl[0]) if nl.size else data.size
        out.append(data[at:end].copy())
        at = end
    if len(out) > 1 and out[-1].size < target // 2:  # a short rest: not a chunk of its own
        out.pop()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=10.0)
    ap.add_argument("--distinct-mib", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    import torch
    t0 = time.perf_counter()
    data = natural_text(a.distinct_mib << 20)
    blocks = cut(data, 16 << 20)
    nl_density = float(np.count_nonzero(data == 10)) / data.size
    hi = float(np.count_nonzero(data >= 0x80)) / data.size
    print(json.dumps({"what": "corpus", "distinct_bytes": int(sum(b.size for b in blocks)), "distinct_chunks": len(blocks),
                      "mean_line_bytes": round(1.0 / max(nl_density, 1e-12), 1), "non_ascii_fraction": round(hi, 6),
                      "read_s": round(time.perf_counter() - t0, 1)}), flush=True)
    nchunks = int(round(a.gib * 2**30 / (16 << 20)))
    plan = np.array([corpus._mix(0xA7, c) % len(blocks) for c in range(nchunks)], dtype=np.int64)
    tbytes = np.array([b.size for b in blocks], dtype=np.int64)
    off, ln, cap = corpus.chunk_table(tbytes[plan])
    nbytes = int(ln.sum())
    dev = torch.device("cuda", 0)
    shard_t = torch.empty(cap, dtype=torch.uint8, device=dev)
    for lo in range(0, len(blocks), 8):  # templates to the device a few at a time (1 GiB of them need not sit there twice)
        dts = {i: torch.from_numpy(blocks[i]).to(dev) for i in range(lo, min(lo + 8, len(blocks)))}
        for c in range(nchunks):
            t = dts.get(int(plan[c]))
            if t is not None:
                shard_t[int(off[c]):int(off[c]) + t.numel()].copy_(t)
        del dts
    torch.cuda.synchronize()
    goffs = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint64)
    ctx = xsg.Context(0)
    sh = xsg.Shard(ctx, shard_t.data_ptr(), cap, xsg.make_chunks(off, ln, goffs))
    lib = xsg.load()
    orc = Oracle()
    mult = np.bincount(plan, minlength=len(blocks))
    for name, pat, flags in CASES:
        ctx.set_pattern(pat, flags)
        t0 = time.perf_counter()
        first = int(sh.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES])
        first_ms = (time.perf_counter() - t0) * 1e3
        if flags & xsg.FLAG_IGNORE_CASE:
            low = orc.lower(pat).tobytes()
            want = int(sum(int(m) * orc.count(orc.lower(b), low, False) for b, m in zip(blocks, mult)))
            want_l = int(sum(int(m) * orc.count(orc.lower(b), low, True) for b, m in zip(blocks, mult)))
        else:
            want = int(sum(int(m) * orc.count(b, pat, False) for b, m in zip(blocks, mult)))
            want_l = int(sum(int(m) * orc.count(b, pat, True) for b, m in zip(blocks, mult)))
        r = timed_calls(lib, sh, a.reps)
        if first != want or r["count"][0] != want or r["count_lines"][0] != want_l:
            raise SystemExit(f"PARITY FAILURE {name}: count {first}/{r['count'][0]} want {want}; lines {r['count_lines'][0]} want {want_l}")
        kern = {m: sh.scan_kernel_name(mode) for m, mode in (("count", xsg.COUNT_MATCHES), ("count_lines", xsg.COUNT_LINES))}
        for mode in (xsg.COUNT_MATCHES, xsg.COUNT_LINES):  # the list calls above leave the card mostly idle (D2H, host work): 30 launches
            sh.time_scan_kernel(mode, 30)                  # bring the clocks back before the kernel's own time is taken
        kms = {m: sh.time_scan_kernel(mode, 7) for m, mode in (("count", xsg.COUNT_MATCHES), ("count_lines", xsg.COUNT_LINES))}
        print(json.dumps({"case": name, "pattern": pat.decode(), "icase": bool(flags), "gib": a.gib, "bytes": nbytes,
                          "matches": want, "matching_lines": want_l, "bytes_per_match": round(nbytes / max(want, 1), 1),
                          "first_count_ms": round(first_ms, 3),
                          "kernel": kern, "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
                          "kernel_frac_of_8tbs": {k: round(nbytes / v / 1e6 / 8000, 4) for k, v in kms.items()},
                          "calls_ms": {t: round(ms, 3) for t, (n, ms) in r.items()},
                          "calls_results": {t: n for t, (n, ms) in r.items()},
                          "calls_frac_of_8tbs": {t: round(nbytes / ms / 1e6 / 8000, 4) for t, (n, ms) in r.items()}}), flush=True)


if __name__ == "__main__":
    main()
