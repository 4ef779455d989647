"""The list tags have two routes behind xsg_search (x-search_amd/csrc/xsg_api.cpp): the one-sync route (capacities,
counts on the device, results mirrored into pinned memory; run_list_fast) and the exact route (every array sized
from a fetched count; also the fallback when a capacity is exceeded).  Both must give the oracle's lists:
search_wrappers.h:136-154,187-207 + the line-index definition of SURVEY 8a row a13."""
import os

import numpy as np
import pytest

import corpus
import xsg
from gpu_util import GpuSearch, oracle_all_modes, oracle_regex_all_modes

pytestmark = pytest.mark.gpu

LIST_KEYS = ("match_byte_offsets", "line_byte_offsets", "line_indices", "lines", "lines_offsets")


class route:
    """environment of one route for the calls inside the block (the library reads it per search)"""

    def __init__(self, **env):
        self.env = {k: str(v) for k, v in env.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


ROUTES = [("one-sync", {}), ("exact", {"XSG_LIST_FAST": 0}), ("overflow->exact", {"XSG_LIST_CAP": 3}),
          ("tight", {"XSG_LIST_CAP": 64})]


def blocks_text():
    bl = [corpus.text_block(7, i, 150_000 + 999 * i, needle_rate=4e-4) for i in range(4)]
    # an unterminated last chunk with a decoy in its tail zone, and a chunk that is one long line
    bl[3] = np.concatenate([bl[3][:-1], np.frombuffer(b" SheSherlock", dtype=np.uint8)])
    long_line = np.frombuffer((b"Sherlock x" * 9000) + b"\n", dtype=np.uint8).copy()
    return bl + [long_line]


@pytest.mark.parametrize("name,env", ROUTES)
def test_every_route_gives_the_oracles_lists(oracle, name, env):
    gs = GpuSearch()
    blocks = blocks_text()
    gs.bind(blocks)
    for pat in (b"Sherlock", b"She", b"lock", b"detective street", b"aa", b"e", b"\n", b"Sherlock x"):
        want = oracle_all_modes(oracle, blocks, pat)
        with route(**env):
            got = gs.all_modes(pat)
        for k in want:
            assert got[k] == want[k], (name, pat, k)
    for expr in (b"She[r ]lock", b"[Ss]her"):
        want, _ = oracle_regex_all_modes(oracle, blocks, expr, False)
        with route(**env):
            got = gs.all_modes(expr, xsg.FLAG_REGEX)
        for k in want:
            assert got[k] == want[k], (name, expr, k)


@pytest.mark.parametrize("name,env", ROUTES)
def test_routes_on_small_alphabet_shards(oracle, name, env):
    """bordered patterns, dense overlaps, many newlines, chunk lengths around the tile and tail-zone sizes"""
    gs = GpuSearch()
    for seed, n in ((1, 40), (2, 700), (3, 16384), (4, 16385), (5, 50_000)):
        blocks = [corpus.small_alphabet(seed * 10 + k, n + 17 * k, terminate=bool(k & 1)) for k in range(3)]
        gs.bind(blocks)
        for pat in (b"a", b"ab", b"aa", b"aba", b"abab", b"bab"):
            want = oracle_all_modes(oracle, blocks, pat)
            with route(**env):
                got = gs.all_modes(pat)
            for k in want:
                assert got[k] == want[k], (name, seed, pat, k)


def test_one_sync_route_keeps_explicit_line_bases_and_global_offsets(oracle):
    gs = GpuSearch()
    blocks = [corpus.text_block(11, i, 70_000, needle_rate=1e-3) for i in range(3)]
    goffs = [1 << 33, (1 << 33) + 100_000, (1 << 34) + 7]
    bases = [1000, 50_000, 7]
    gs.bind(blocks, goffs, bases)
    want = oracle_all_modes(oracle, blocks, b"Sherlock", global_offsets=goffs, line_bases=bases)
    got = gs.all_modes(b"Sherlock")
    for k in want:
        assert got[k] == want[k], k
    # the view is the pinned mirror itself
    v = gs.shard.search_u64_view(xsg.MATCH_BYTE_OFFSETS)
    assert v.tolist() == want["match_byte_offsets"]


def test_a_binding_remembers_that_a_pattern_overflowed(oracle):
    """after a capacity overflow the exact route serves the pattern at once (no wasted one-sync attempt), and a
    re-bind or a new pattern tries the one-sync route again; results identical throughout"""
    gs = GpuSearch()
    blocks = [corpus.text_block(3, 0, 300_000, needle_rate=1e-3)]
    gs.bind(blocks)
    want = oracle_all_modes(oracle, blocks, b"the")
    with route(XSG_LIST_CAP=16):
        for _ in range(2):
            got = gs.all_modes(b"the")
            for k in LIST_KEYS:
                assert got[k] == want[k], k
    got = gs.all_modes(b"the")  # default capacity again, same binding: still the exact route; same lists
    for k in LIST_KEYS:
        assert got[k] == want[k], k
    gs.bind(blocks)
    got = gs.all_modes(b"the")
    for k in LIST_KEYS:
        assert got[k] == want[k], k


@pytest.mark.parametrize("env", [{}, {"XSG_LINES_EAGER": 0}], ids=["pinned-mirrors", "on-demand"])
def test_xs_lines_of_a_result_of_hundreds_of_thousands_of_lines(oracle, env):
    """(both ways a large result leaves: written into the pinned mirrors by the kernels, or gathered on the device and copied
    when an accessor asks)  xs::lines on the exact route with a result large enough for the output-centric gather (k_line_gather_span, from 65 536
    lines): short lines, empty lines, lines of several KB, an unterminated last line (dropped: search_wrappers.h:199-202),
    chunk sizes that leave partial 16-byte units at every slice and workgroup boundary."""
    rng = np.random.default_rng(20260404)
    words = [b"e", b"be", b"tree", b"x", b"", b"zz", b"seven eleven", b"q", b"the end", b"yyyy"]
    blocks = []
    for c in range(3):
        parts = []
        for i in range(120_000 + 777 * c):
            k = int(rng.integers(0, 40))
            if k == 0:
                parts.append(b"e" * int(rng.integers(300, 9000)))  # a long line
            elif k < 8:
                parts.append(b"")  # an empty line
            else:
                parts.append(b" ".join(words[int(j)] for j in rng.integers(0, len(words), size=int(rng.integers(1, 6)))))
        text = b"\n".join(parts) + (b"\n" if c != 2 else b" e unterminated")
        blocks.append(np.frombuffer(text, dtype=np.uint8).copy())
    gs = GpuSearch()
    gs.bind(blocks)
    for pat in (b"e", b"ee", b"seven"):
        want = oracle_all_modes(oracle, blocks, pat)
        with route(**env):
            got = gs.all_modes(pat)
        assert len(want["lines"]) > (65536 if pat == b"e" else 0)
        for k in want:
            assert got[k] == want[k], (pat, k)
