"""The reference's integration goldens (test/src/xsearchTest.cpp:17-335) and its metafile (test/files/sample.meta),
held against each other -- and replayed on a stand-in for the missing corpus.

This is what pins xs::line_indices (no implementation in the snapshot) with reference-held data:
the goldens' (line byte offset, line index) pairs live in the same index space as the metafile's (globalByteOffset,
globalLineIndex) mapping, whose first entry is (0, 0) and whose entries are line starts -- a line's index is the number
of '\\n' before its first byte, 0-based, global over the file.  No GPU needed.
"""
import re
from pathlib import Path

import numpy as np
import pytest

import golden_util as G
import sample_standin as S
import xsg
from xs_oracle import Oracle

REF_META = Path("/root/reference/test/files/sample.meta")
VEC = G.load("ref_xsearchtest_vectors.json")
MAP = G.load("ref_xsearchtest_mapping_brackets.json")
FAMILIES = ("literal_case", "literal_icase", "regex_case", "regex_icase")


def first_match(family: str, line: str):
    if family.startswith("literal"):
        hay, needle = (line.lower(), "sherlock") if family.endswith("icase") else (line, "Sherlock")
        return hay.find(needle), 8
    m = re.search("She[r ]lock", line, re.I if family.endswith("icase") else 0)
    return (m.start(), m.end() - m.start()) if m else (-1, 0)


@pytest.mark.parametrize("family", FAMILIES)
def test_vectors_of_a_family_agree_with_each_other(family):
    f = VEC["families"][family]
    n = {"literal_case": 46, "literal_icase": 48, "regex_case": 53, "regex_icase": 59}[family]  # xsearchTest.cpp:17-24
    assert f["count_lines"] == f["count_matches"] == n
    lo, mo, li, ln = f["line_byte_offsets"], f["match_byte_offsets"], f["line_indices"], f["lines"]
    assert len(lo) == len(mo) == len(li) == len(ln) == n
    assert lo == sorted(set(lo)) and li == sorted(set(li)) and mo == sorted(set(mo))
    for k in range(n):
        pos, mlen = first_match(family, ln[k])
        assert pos >= 0, (family, ln[k])
        assert mo[k] == lo[k] + pos, "global match offset = global line offset + position of the FIRST match in the line"
        assert "\n" not in ln[k], "lines exclude the newline (search_wrappers.h:204)"
        assert mo[k] + mlen <= lo[k] + len(ln[k])
        if k:
            # every line between two golden lines owns at least its newline
            assert lo[k] - lo[k - 1] >= len(ln[k - 1]) + 1 + (li[k] - li[k - 1] - 1)
            if li[k] == li[k - 1] + 1:
                assert lo[k] == lo[k - 1] + len(ln[k - 1]) + 1
    assert mo[-1] < VEC["file_size"]


def test_families_nest_and_share_their_lines():
    fam = VEC["families"]
    rows = {k: {o: (i, t, m) for o, i, t, m in zip(f["line_byte_offsets"], f["line_indices"], f["lines"], f["match_byte_offsets"])}
            for k, f in fam.items()}
    for small, big in (("literal_case", "literal_icase"), ("literal_case", "regex_case"), ("literal_icase", "regex_icase"),
                       ("regex_case", "regex_icase")):
        for o, row in rows[small].items():
            assert rows[big][o][:2] == row[:2], (small, big, o)


def check_against_mapping(offset, index, lo, hi):
    """a golden (line offset, line index) against the mapping entries around it: all three are line starts"""
    assert lo[0] <= offset and (hi is None or offset < hi[0])
    if offset == lo[0]:
        assert index == lo[1], "a golden pair that coincides with a mapping entry agrees with it"
    else:
        assert lo[1] < index and index - lo[1] <= offset - lo[0]
    if hi is not None:
        assert index < hi[1] and hi[1] - index <= hi[0] - offset


def test_line_indices_live_in_the_mappings_index_space():
    assert MAP["first_mapping"] == [0, 0], "offset 0 is line 0: indices are 0-based counts of the newlines before a line"
    coincide = 0
    for fam in VEC["families"].values():
        for o, i in zip(fam["line_byte_offsets"], fam["line_indices"]):
            b = MAP["brackets"][str(o)]
            check_against_mapping(o, i, b[:2], b[2:] or None)
            coincide += o == b[0]
    assert coincide >= 10  # ten pairs are mapping entries themselves
    # and in the chunk table's: a chunk's first_line is the index of its first line
    chunks = MAP["chunks"]
    assert sum(c["original_size"] for c in chunks) == VEC["file_size"]
    for fam in VEC["families"].values():
        for o, i in zip(fam["line_byte_offsets"], fam["line_indices"]):
            c = [c for c in chunks if c["original_offset"] <= o < c["original_offset"] + c["original_size"]]
            assert len(c) == 1
            assert c[0]["first_line"] <= i and i - c[0]["first_line"] <= o - c[0]["original_offset"]


@pytest.mark.skipif(not REF_META.exists(), reason="reference tree not mounted")
def test_brackets_are_what_xsg_meta_read_finds_in_sample_meta():
    comp, chunks, maps = xsg.meta_read(str(REF_META), with_mappings=True)
    maps = np.asarray(maps, dtype=np.uint64).reshape(-1, 2)
    assert comp == xsg.COMPRESSION_NONE and [int(x) for x in maps[0]] == [0, 0]
    assert [int(x) for x in maps[-1]] == MAP["last_mapping"]
    assert [(int(c["original_offset"]), int(c["original_size"]), int(c["first_line"])) for c in chunks] == \
           [(c["original_offset"], c["original_size"], c["first_line"]) for c in MAP["chunks"]]
    assert bool(np.all(np.diff(maps[:, 0].astype(np.int64)) > 0)) and bool(np.all(np.diff(maps[:, 1].astype(np.int64)) > 0))
    for fam in VEC["families"].values():
        for o, i in zip(fam["line_byte_offsets"], fam["line_indices"]):
            k = int(np.searchsorted(maps[:, 0], np.uint64(o), side="right")) - 1
            lo = [int(x) for x in maps[k]]
            hi = [int(x) for x in maps[k + 1]] if k + 1 < len(maps) else None
            assert MAP["brackets"][str(o)] == lo + (hi or [])
            check_against_mapping(o, i, lo, hi)


# ---- the goldens replayed on the stand-in corpus (sample_standin.py), CPU side: the oracle ---------------------------
@pytest.fixture(scope="module")
def standin(tmp_path_factory):
    path = tmp_path_factory.mktemp("standin") / "sample.txt"
    data = S.build(str(path))
    return str(path), data


def test_standin_has_the_references_chunk_plan(standin):
    path, data = standin
    assert data.size == VEC["file_size"] and data[-1] == 10
    plan = xsg.plan_chunks(path, 16 << 20)
    assert [(int(c["original_offset"]), int(c["original_size"])) for c in plan] == \
           [(c["original_offset"], c["original_size"]) for c in MAP["chunks"]]
    nl = np.flatnonzero(data == 10)
    for c in MAP["chunks"]:  # the number of newlines before every chunk is its first_line
        assert int(np.searchsorted(nl, c["original_offset"])) == c["first_line"]


@pytest.mark.parametrize("family", FAMILIES)
def test_oracle_reproduces_the_references_goldens_on_the_standin(standin, family):
    from gpu_util import oracle_all_modes, oracle_regex_all_modes
    path, data = standin
    want = S.expected(family)
    blocks = [data[c["original_offset"]:c["original_offset"] + c["original_size"]] for c in MAP["chunks"]]
    icase = family.endswith("icase")
    if family.startswith("literal"):
        got = oracle_all_modes(Oracle(), blocks, b"Sherlock", ignore_case=icase)
    else:
        got, with_lines = oracle_regex_all_modes(Oracle(), blocks, b"She[r ]lock", ignore_case=icase)
        assert with_lines
    got["lines"] = [x.decode("latin-1") for x in got["lines"]]
    for k, v in want.items():
        assert got[k] == v, (family, k)
    assert got["lines_offsets"] == want["line_byte_offsets"]
