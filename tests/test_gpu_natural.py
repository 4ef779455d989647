"""Parity on text nobody generated for the purpose: the sources and data files of the Python standard library and of
the installed pure-Python packages
(present in this image on the GPU box too), concatenated into a few chunks -- real words, real line lengths,
tabs, long lines, some non-ASCII bytes -- against the oracle, every tag."""
import sys
import sysconfig
from pathlib import Path

import numpy as np
import pytest

import xsg
from gpu_util import GpuSearch, oracle_all_modes, oracle_regex_all_modes

pytestmark = pytest.mark.gpu


def stdlib_text(limit=48 << 20):
    files = []
    for key in ("stdlib", "purelib"):  # the standard library, then installed pure-Python packages
        root = Path(sysconfig.get_paths()[key])
        files += sorted(p for p in root.rglob("*") if p.suffix in (".py", ".txt", ".rst", ".cfg") and p.is_file())
    parts, n = [], 0
    for p in files:
        try:
            b = p.read_bytes()
        except OSError:
            continue
        parts.append(b)
        n += len(b)
        if n >= limit:
            break
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    return data


def chunks_of(data, target):
    """newline-aligned cuts, like the reader's plan"""
    out, at = [], 0
    while at < data.size:
        end = min(at + target, data.size)
        if end < data.size:
            nl = np.flatnonzero(data[end:end + (1 << 20)] == 10)
            end = end + int(nl[0]) + 1 if nl.size else data.size
        out.append(data[at:end].copy())
        at = end
    return out


def test_standard_library_sources_every_tag(oracle):
    data = stdlib_text()
    if data.size < (4 << 20):
        pytest.skip(f"only {data.size} bytes of standard-library text on this host")
    blocks = chunks_of(data, 6 << 20)
    gs = GpuSearch()
    gs.bind(blocks)
    for pat in (b"import", b"self.", b"def ", b"Exception", b"the", b"e", b"\t", b"return None", b"raise ValueError(",
                b"    def __init__(self", b"\xc3\xa9", b"Copyright (c) 2001-2023 Python Software Foundation"):
        for flags in (0, xsg.FLAG_IGNORE_CASE, xsg.FLAG_EXACT_TAIL):
            got = gs.all_modes(pat, flags)
            want = oracle_all_modes(oracle, blocks, pat, bool(flags & xsg.FLAG_EXACT_TAIL),
                                    ignore_case=bool(flags & xsg.FLAG_IGNORE_CASE))
            for k in want:
                assert got[k] == want[k], (pat, flags, k)
    for expr in (b"[Ee]rror", b"def [a-z_]{4}\\(", b"[0-9]{4}-[0-9]{2}-[0-9]{2}", b"x[0-9]\\]"):
        for icase in (False, True):
            want, with_lines = oracle_regex_all_modes(oracle, blocks, expr, icase)
            got = gs.all_modes(expr, xsg.FLAG_REGEX | (xsg.FLAG_IGNORE_CASE if icase else 0), lines=with_lines)
            for k in want:
                assert got[k] == want[k], (expr, icase, k)
    total = oracle_all_modes(oracle, blocks, b"import")
    assert total["count_matches"] > 1000 and total["count_lines"] > 1000
