#!/usr/bin/env python3
"""Where does the byte-parallel route of the 4..8-byte kinds start to pay?  (note_density, xsg_api.cpp.)

Words of the natural-text corpus (scripts/natural_variants.py) between one occurrence per 2 KiB and one per 100 KiB, each
counted with the route pinned either way (XSG_DENSE_PER, a test hook): the scan kernel's own time for count and
count_lines.  The crossover this prints is the threshold note_density carries."""
import argparse
import json
import os
import sys
from pathlib import Path

os.environ["XSG_TEST_HOOKS"] = "1"
import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "x-search_amd", ROOT / "oracle", ROOT / "scripts"):
    sys.path.insert(0, str(p))
import corpus  # noqa: E402
import xsg  # noqa: E402
from natural_variants import cut, natural_text  # noqa: E402

WORDS = [(b"self", 0), (b"return", 0), (b"import", 0), (b"None", 0), (b"error", xsg.FLAG_IGNORE_CASE), (b"class", 0),
         (b"else", 0), (b"raise", 0), (b"False", 0), (b"except", 0), (b"lambda", 0), (b"assert", 0), (b"yield", 0),
         (b"while", 0), (b"warning", xsg.FLAG_IGNORE_CASE), (b"finally", 0), (b"global", 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=10.0)
    ap.add_argument("--distinct-mib", type=int, default=1024)
    a = ap.parse_args()
    import torch
    data = natural_text(a.distinct_mib << 20)
    blocks = cut(data, 16 << 20)
    nchunks = int(round(a.gib * 2**30 / (16 << 20)))
    plan = np.array([corpus._mix(0xA7, c) % len(blocks) for c in range(nchunks)], dtype=np.int64)
    tbytes = np.array([b.size for b in blocks], dtype=np.int64)
    off, ln, cap = corpus.chunk_table(tbytes[plan])
    nbytes = int(ln.sum())
    dev = torch.device("cuda", 0)
    shard_t = torch.empty(cap, dtype=torch.uint8, device=dev)
    for lo in range(0, len(blocks), 8):
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
    for pat, flags in WORDS:
        rec = {"pattern": pat.decode(), "icase": bool(flags), "plen": len(pat)}
        for route, per in (("hot_filter", "1"), ("byte_parallel", str(1 << 40))):
            os.environ["XSG_DENSE_PER"] = per
            ctx.set_pattern(pat, flags)
            n = int(sh.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES])  # teaches the shard the density
            sh.count(xsg.COUNT_MATCHES)
            sh.count(xsg.COUNT_LINES)
            rec["bytes_per_match"] = round(nbytes / max(n, 1))
            rec[route] = {m: round(nbytes / sh.time_scan_kernel(mode, 7) / 1e6 / 8000, 4)
                          for m, mode in (("count", xsg.COUNT_MATCHES), ("count_lines", xsg.COUNT_LINES))}
            rec[route]["kernel"] = sh.scan_kernel_name(xsg.COUNT_MATCHES).split(" stagger")[0][-40:]
        print(json.dumps(rec), flush=True)
    os.environ.pop("XSG_DENSE_PER", None)


if __name__ == "__main__":
    main()
