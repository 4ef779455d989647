#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the file pipeline: a page-cache/tmpfs
resident text file -> pread into pinned buffers -> hipMemcpyAsync -> scan -> result.
Bounded by PCIe Gen5 x16 (~63 GB/s) and the host read path, never by HBM; reported
next to -- never instead of -- bench.py's device-resident number."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "x-search_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=8.0)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--threads", default="1,2,4,8,16",
                    help="comma list of worker counts; W:R sets the reader (pread + decompress) threads separately")
    ap.add_argument("--modes", default="count,count_lines,match_byte_offsets,lines")
    ap.add_argument("--meta", default="", help="also run through a metafile: comma list of none,lz4,zst")
    a = ap.parse_args()
    import xsg
    from xs_oracle import Oracle
    args = argparse.Namespace(chunk_mib=16, templates=16, seed=0x5EED)
    blocks = bench.template_blocks(args, b"Sherlock")
    n = int(a.gib * 2**30 / (16 << 20))
    plan = bench.chunk_plan(args, 0, n)
    path = os.path.join(a.dir, f"xsg_e2e_{os.getpid()}.txt")
    with open(path, "wb") as f:
        for c in plan:
            f.write(blocks[int(c)].tobytes())
    size = os.path.getsize(path)
    orc = Oracle()
    tc = [orc.count(b, b"Sherlock", False) for b in blocks]
    want = sum(tc[int(c)] for c in plan)
    modes = {"count": xsg.COUNT_MATCHES, "count_lines": xsg.COUNT_LINES, "match_byte_offsets": xsg.MATCH_BYTE_OFFSETS,
             "lines": xsg.LINES, "line_indices": xsg.LINE_INDICES, "line_byte_offsets": xsg.LINE_BYTE_OFFSETS}
    variants = [("plain", path, None)]
    made = []
    for kind in [k for k in a.meta.split(",") if k]:
        comp = {"none": xsg.COMPRESSION_NONE, "lz4": xsg.COMPRESSION_LZ4, "zst": xsg.COMPRESSION_ZSTD}[kind]
        mp, dp = path + f".{kind}.meta", path + f".{kind}"
        t0 = time.perf_counter()
        xsg.meta_write(path, mp, dp, comp)
        made += [mp, dp]
        dsz = os.path.getsize(dp) if comp != xsg.COMPRESSION_NONE else size
        print(json.dumps({"preprocess": kind, "seconds": round(time.perf_counter() - t0, 2),
                          "compressed_gib": round(dsz / 2**30, 3)}), flush=True)
        variants.append((kind, dp if comp != xsg.COMPRESSION_NONE else path, mp))
    # warm-up: the first job of a process pays HIP init, thread start-up and the pinned/device buffer pools
    j = xsg.Job(b"Sherlock", path, xsg.COUNT_MATCHES, num_threads=8, num_max_readers=8)
    j.result()
    j.close()
    try:
      for vname, dpath, mpath in variants:
        for name in a.modes.split(","):
            for spec in a.threads.split(","):
                th, rd = (int(x) for x in spec.split(":")) if ":" in spec else (int(spec), int(spec))
                t0 = time.perf_counter()
                j = xsg.Job(b"Sherlock", dpath, modes[name], meta_path=mpath, num_threads=th, num_max_readers=rd)
                r = j.result()
                dt = time.perf_counter() - t0
                st = j.stats()
                j.close()
                got = r if isinstance(r, int) else len(r)
                ok = (got == want) if name in ("count", "match_byte_offsets") else None
                print(json.dumps({"input": vname, "mode": name, "threads": th, "readers": rd, "gib": round(size / 2**30, 2), "seconds": round(dt, 3),
                                  "gib_per_s": round(size / dt / 2**30, 2), "result": got, "parity": ok,
                                  "read_s": round(st["seconds_read"], 2), "decompress_s": round(st["seconds_decompress"], 2),
                                  "device_s": round(st["seconds_device"], 2)}),
                      flush=True)
    finally:
        os.unlink(path)
        for f in made:
            if os.path.exists(f):
                os.unlink(f)


if __name__ == "__main__":
    main()
