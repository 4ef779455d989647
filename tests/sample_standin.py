"""A stand-in for the reference's missing 100 MB test corpus (test/files/sample.txt, .MISSING_LARGE_BLOBS).

TEST INFRASTRUCTURE.  The reference's integration tests (test/src/xsearchTest.cpp, 96 TESTs) assert golden vectors that
were recorded on a file the snapshot does not hold.  Everything the snapshot DOES say about that file is committed as
data under tests/golden/ (extract_ref_xsearchtest_vectors.py):

  * it is exactly 100 000 000 bytes, cut into six chunks of 16 MiB extended to just past the next newline, whose first
    lines have the global indices 0, 579882, 1175517, 1750949, 2328466, 2894949 (test/files/sample.meta);
  * 59 of its lines: byte offset, 0-based line index and text of every line that matches `Sherlock` / `She[r ]lock`,
    case-sensitively or not (xsearchTest.cpp:25-335), and the offset of the match inside each;
  * around each of those lines, two (byte offset, line index) pairs of the metafile's mapping; the mapping's last
    entry (99 999 691 -> line 3 447 129).

build() writes a file that agrees with every one of these facts: the 59 lines stand at their offsets, every anchor
offset is a line start with exactly its recorded number of newlines before it, the chunk plan of the file is the
reference's chunk table, and nothing else in the file can match (the filler has no 's' or 'S').  On such a file the
reference's expected values hold verbatim, so its integration suite becomes replayable: the oracle on the CPU
(tests/test_ref_xsearchtest_vectors.py) and the product on the GPU (tests/test_gpu_xsearchtest.py) must return the
reference's own vectors -- counts, global byte offsets, global line indices, lines -- for all four pattern families.
"""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
FILE_SIZE = 100_000_000
TARGET = 16 << 20
TAIL_LINES = 10  # lines after the mapping's last entry (99 999 691): not recorded anywhere; 309 bytes of ~30-byte lines

_WORDS = (b"the of and to a in that it for on you be with by at not but they from which or we an been their would what "
          b"will there if can all her who one up them could him into time detective").split()


def vectors():
    return json.loads((GOLDEN / "ref_xsearchtest_vectors.json").read_text())


def mapping():
    return json.loads((GOLDEN / "ref_xsearchtest_mapping_brackets.json").read_text())


def anchors():
    """sorted (offset, line index, text or None): every position the reference pins to a line number"""
    vec, mp = vectors(), mapping()
    pts = {}

    def add(o, i, text=None):
        if o in pts:
            assert pts[o][0] == i, (o, i, pts[o])
            if text is not None:
                assert pts[o][1] in (None, text), (o, text, pts[o])
                pts[o] = (i, text)
        else:
            pts[o] = (i, text)

    for c in mp["chunks"]:
        add(c["original_offset"], c["first_line"])
    for b in mp["brackets"].values():
        add(b[0], b[1])
        if len(b) == 4:
            add(b[2], b[3])
    add(*mp["last_mapping"])
    for fam in vec["families"].values():
        for o, i, t in zip(fam["line_byte_offsets"], fam["line_indices"], fam["lines"]):
            add(o, i, t)
    # the last line of every chunk but the last must cover the byte at (chunk start + 16 MiB - 1): the chunk is "16 MiB
    # extended to just past the next newline", so no newline may lie between that byte and the chunk's last one
    for c, nxt in zip(mp["chunks"], mp["chunks"][1:]):
        end = nxt["original_offset"]
        start_of_last_line = c["original_offset"] + TARGET - 1 - 20
        assert all(not (start_of_last_line <= o < end) for o in pts), "an anchor inside a chunk's last line"
        add(start_of_last_line, nxt["first_line"] - 1)
    add(FILE_SIZE, mp["last_mapping"][1] + TAIL_LINES)
    return sorted((o, i, t) for o, (i, t) in pts.items())


def build(path=None) -> np.ndarray:
    """the stand-in as a uint8 array (and written to `path` if given)"""
    rng = np.random.default_rng(0x5A4D)
    ids = rng.integers(0, len(_WORDS), size=200_000)
    stream = np.frombuffer(b" ".join(_WORDS[i] for i in ids) + b" ", dtype=np.uint8)
    assert b"s" not in stream.tobytes() and b"S" not in stream.tobytes() and b"\n" not in stream.tobytes()
    data = np.resize(stream, FILE_SIZE).copy()
    pts = anchors()
    assert pts[0][:2] == (0, 0), "the mapping's first entry: offset 0 is line 0 (indices are 0-based)"
    for (o, i, text), (o2, i2, _) in zip(pts, pts[1:]):
        nbytes, nlines = o2 - o, i2 - i
        assert nlines >= 1 and nbytes >= nlines, ("inconsistent anchors", o, i, o2, i2)
        at = o
        if text is not None:
            raw = text.encode("latin-1")
            assert nbytes >= len(raw) + 1 + (nlines - 1), ("no room behind a golden line", o, text)
            data[at:at + len(raw)] = np.frombuffer(raw, dtype=np.uint8)
            data[at + len(raw)] = 10
            at += len(raw) + 1
            nlines -= 1
            nbytes -= len(raw) + 1
        assert (nlines == 0) == (nbytes == 0), ("bytes without a line", o, o2)
        if nlines:
            base, extra = divmod(nbytes, nlines)
            lens = np.full(nlines, base, dtype=np.int64)
            lens[nlines - extra:] += 1  # the longer lines last
            ends = at + np.cumsum(lens) - 1
            data[ends] = 10
    if path is not None:
        data.tofile(path)
    return data


def expected(family: str):
    """the reference's golden vectors of one family ('literal_case', 'literal_icase', 'regex_case', 'regex_icase')"""
    return vectors()["families"][family]
