#!/usr/bin/env python3
"""Lifts the golden DATA of the reference's integration tests into JSON (build container only: needs /root/reference).

  test/src/xsearchTest.cpp:17-24    the eight counts (46 / 48 / 53 / 59, lines and matches)
  test/src/xsearchTest.cpp:25-59    line byte offsets     (literal / regex  x  case / ignore-case)
  test/src/xsearchTest.cpp:60-94    match byte offsets
  test/src/xsearchTest.cpp:95-125   line indices
  test/src/xsearchTest.cpp:126-335  the matching lines themselves
      -> tests/golden/ref_xsearchtest_vectors.json

  test/files/sample.meta            for every golden line byte offset, the two mapping entries (globalByteOffset,
                                    globalLineIndex) that bracket it, read through the product's own parser
                                    (xsg_meta_read), plus the chunk table
      -> tests/golden/ref_xsearchtest_mapping_brackets.json

Only data is taken -- numbers and the test's expected strings -- no code.  The corpus these vectors were recorded on
(test/files/sample.txt, exactly 100 000 000 bytes) is missing from the snapshot (.MISSING_LARGE_BLOBS), so the vectors
cannot be replayed on it; tests/test_ref_xsearchtest_vectors.py checks them against each other and against the mapping,
and tests/sample_standin.py builds a stand-in corpus that agrees with every one of these facts.
"""
import json
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "x-search_amd"))
REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def unescape(s: str) -> str:
    out, i = [], 0
    while i < len(s):
        if s[i] == "\\":
            nxt = s[i + 1]
            out.append({"n": "\n", "t": "\t", "\\": "\\", '"': '"', "'": "'"}[nxt])
            i += 2
        else:
            out.append(s[i])
            i += 1
    return "".join(out)


def main():
    src = (REF / "test/src/xsearchTest.cpp").read_text()
    head = src[:src.index("// _____ Reading plain text file without metadata")]
    scalars = {m.group(1): int(m.group(2)) for m in re.finditer(r"static const uint64_t (\w+) = (\d+);", head)}
    strings = {m.group(1): unescape(m.group(2)) for m in re.finditer(r'static const std::string (\w+)\("((?:[^"\\]|\\.)*)"\);', head)}
    u64 = {m.group(1): [int(x) for x in re.findall(r"\d+", m.group(2))]
           for m in re.finditer(r"static const std::vector<uint64_t> (\w+)\{([^}]*)\};", head)}
    strs = {m.group(1): [unescape(x) for x in re.findall(r'"((?:[^"\\]|\\.)*)"', m.group(2))]
            for m in re.finditer(r"static const std::vector<std::string> (\w+)\{((?:[^}\"]|\"(?:[^\"\\]|\\.)*\")*)\};", head)}
    out = {"source": "test/src/xsearchTest.cpp:8-335", "file": strings["file_path"], "file_size": 100_000_000,
           "patterns": {"literal": strings["pattern"], "regex": strings["re_pattern"]}, "families": {}}
    for kind in ("literal", "regex"):
        for case in ("case", "icase"):
            k = f"{kind}_{case}"
            fam = {"count_lines": scalars[f"{k}_count"], "count_matches": scalars[f"{k}_count_match"],
                   "line_byte_offsets": u64[f"{k}_line_byte_offsets"], "match_byte_offsets": u64[f"{k}_match_byte_offsets"],
                   "line_indices": u64[f"{k}_line_indices"], "lines": strs[f"{k}_lines"]}
            n = fam["count_lines"]
            assert all(len(fam[f]) == n for f in ("line_byte_offsets", "match_byte_offsets", "line_indices", "lines")), k
            out["families"][k] = fam
    (OUT / "ref_xsearchtest_vectors.json").write_text(json.dumps(out, indent=1))

    import xsg
    comp, chunks, maps = xsg.meta_read(str(REF / "test/files/sample.meta"), with_mappings=True)
    maps = np.asarray(maps, dtype=np.uint64).reshape(-1, 2)
    offs = sorted({o for fam in out["families"].values() for o in fam["line_byte_offsets"]})
    br = {}
    for o in offs:
        k = int(np.searchsorted(maps[:, 0], np.uint64(o), side="right")) - 1
        lo = maps[k]
        hi = maps[k + 1] if k + 1 < len(maps) else None
        br[str(o)] = [int(lo[0]), int(lo[1])] + ([int(hi[0]), int(hi[1])] if hi is not None else [])
    meta = {"source": "test/files/sample.meta", "compression_type": int(comp),
            "chunks": [{"original_offset": int(c["original_offset"]), "original_size": int(c["original_size"]),
                        "first_line": int(c["first_line"])} for c in chunks],
            "first_mapping": [int(maps[0][0]), int(maps[0][1])], "last_mapping": [int(maps[-1][0]), int(maps[-1][1])],
            "brackets": br}
    (OUT / "ref_xsearchtest_mapping_brackets.json").write_text(json.dumps(meta, indent=1))
    print("ok", {k: v["count_lines"] for k, v in out["families"].items()}, len(br), "distinct line offsets")


if __name__ == "__main__":
    main()
