"""CPU check of the GPU algorithm's decomposition (bulk + keep + tail walk + line
summaries), built from the product's own host/device headers, against the
oracle.  No GPU needed."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import golden_util as G

HERE = Path(__file__).resolve().parent
SRC = HERE / "cpu_model" / "host_model.cpp"
LIB = HERE / "cpu_model" / "build" / "libhost_model.so"
_u64p = C.POINTER(C.c_uint64)


@pytest.fixture(scope="module")
def hm():
    LIB.parent.mkdir(exist_ok=True)
    deps = [SRC, HERE.parent / "x-search_amd/csrc/xsg_tail.h", HERE.parent / "x-search_amd/csrc/xsg_linesum.h"]
    if not LIB.exists() or LIB.stat().st_mtime < max(p.stat().st_mtime for p in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", str(LIB), str(SRC)])
    lib = C.CDLL(str(LIB))
    vp, u64, u32, ci = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    lib.hm_list.argtypes = [vp, u64, C.c_char_p, u32, ci, ci, _u64p, u64]
    lib.hm_list.restype = u64
    lib.hm_lines.argtypes = [vp, u64, C.c_char_p, u32, ci, _u64p, _u64p, u64]
    lib.hm_lines.restype = u64
    lib.hm_count_matches_borderfree.argtypes = [vp, u64, C.c_char_p, u32, ci]
    lib.hm_count_matches_borderfree.restype = u64
    lib.hm_count_lines.argtypes = [vp, u64, C.c_char_p, u32, ci, u32]
    lib.hm_count_lines.restype = u64
    lib.hm_set_icase.argtypes = [ci]
    lib.hm_check_lane_combine.argtypes = [u64, u64]
    lib.hm_check_lane_combine.restype = u64
    for name in ("hm_check_tail_masks", "hm_check_span_combine"):
        getattr(lib, name).argtypes = [u64, u64]
        getattr(lib, name).restype = u64
    return lib


def _list(lib, data, p, exact, line_mode):
    n = lib.hm_list(data.ctypes.data, data.size, p, len(p), exact, line_mode, None, 0)
    out = np.empty(n, dtype=np.uint64)
    lib.hm_list(data.ctypes.data, data.size, p, len(p), exact, line_mode, out.ctypes.data_as(_u64p), n)
    return out.tolist()


def _lines(lib, data, p, exact):
    n = lib.hm_lines(data.ctypes.data, data.size, p, len(p), exact, None, None, 0)
    b = np.empty(n, dtype=np.uint64)
    l = np.empty(n, dtype=np.uint64)
    lib.hm_lines(data.ctypes.data, data.size, p, len(p), exact, b.ctypes.data_as(_u64p), l.ctypes.data_as(_u64p), n)
    return b.tolist(), l.tolist()


def bordered(p: bytes) -> bool:
    return any(p[:k] == p[-k:] for k in range(1, len(p)))


def check_case(hm, oracle, data, p, exact):
    data = np.ascontiguousarray(data)
    if data.size == 0:
        data = np.zeros(1, dtype=np.uint8)[:0]
    oracle.set_exact(bool(exact))
    try:
        want_m = oracle.byte_offsets_match(data, p).tolist()
        assert _list(hm, data, p, exact, 0) == want_m
        if not bordered(p):
            assert hm.hm_count_matches_borderfree(data.ctypes.data, data.size, p, len(p), exact) == len(want_m)
        if b"\n" not in p:
            assert _list(hm, data, p, exact, 1) == oracle.byte_offsets_line(data, p).tolist()
            wb, wl = oracle.lines_spans(data, p)
            assert _lines(hm, data, p, exact) == (wb.tolist(), wl.tolist())
            want_c = oracle.count(data, p, True)
            for group in (1, 7, 64, 1024):
                assert hm.hm_count_lines(data.ctypes.data, data.size, p, len(p), exact, group) == want_c
    finally:
        oracle.set_exact(False)


def test_model_on_golden_inputs(hm, oracle):
    for name, data, e in G.generated_cases():
        p = e["pattern"].encode("latin-1")
        if name == "text_1m":
            continue  # brute-force bulk is quadratic-ish; covered on the GPU
        for exact in (0, 1):
            check_case(hm, oracle, data, p, exact)


def test_model_random_small_alphabet(hm, oracle):
    rng = np.random.default_rng(2024)
    pats = [b"a", b"aa", b"ab", b"aba", b"abab", b"bab", b"aab", b"abc", b"b a", b"ab\n", b"aaaa", b"abcabcab",
            b"ababababa"]
    alph = np.frombuffer(b"ab \nc", dtype=np.uint8)
    for it in range(6000):
        n = int(rng.integers(0, 400))
        k = 2 + it % 4
        data = alph[rng.integers(0, k, size=n)].copy()
        p = pats[int(rng.integers(0, len(pats)))]
        check_case(hm, oracle, data, p, it & 1)


def test_model_tail_decoys(hm, oracle):
    """Partial-prefix decoys right at the end of the chunk: the lossy-tail zone."""
    rng = np.random.default_rng(99)
    for it in range(3000):
        p = [b"Sherlock", b"aab", b"abcab", b"xyxz", b"lock"][it % 5]
        n = int(rng.integers(0, 200))
        body = np.frombuffer(bytes(rng.choice(list(b"xyzab Sherlock\n"), size=n).astype(np.uint8)), dtype=np.uint8)
        # glue decoy + pattern fragments at the end
        parts = []
        for _ in range(int(rng.integers(1, 5))):
            cut = int(rng.integers(1, len(p) + 1))
            parts.append(p[:cut])
            if rng.random() < 0.5:
                parts.append(p)
            if rng.random() < 0.3:
                parts.append(b"\n")
        tail = np.frombuffer(b"".join(parts), dtype=np.uint8)
        data = np.concatenate([body, tail])
        check_case(hm, oracle, data, p, 0)
        check_case(hm, oracle, data, p, 1)


def test_model_ignore_case(hm, oracle):
    """ignore_case = search(toLower(chunk), toLower(pattern)): the device folds bytes
    on the fly (xsg::fold / fold4) and uses a lowered pattern."""
    rng = np.random.default_rng(31337)
    alph = np.frombuffer(b"aAbB \nxX", dtype=np.uint8)
    pats = [b"a", b"Ab", b"aBa", b"ABAB", b"b A", b"Xx", b"aab", b"abABabAB"]
    hm.hm_set_icase(1)
    try:
        for it in range(3000):
            n = int(rng.integers(0, 300))
            data = alph[rng.integers(0, len(alph), size=n)].copy()
            p = pats[int(rng.integers(0, len(pats)))]
            lowered = oracle.lower(data)
            pl = p.lower()
            for exact in (0, 1):
                oracle.set_exact(bool(exact))
                want_m = oracle.byte_offsets_match(lowered, pl).tolist()
                want_l = oracle.byte_offsets_line(lowered, pl).tolist()
                want_c = oracle.count(lowered, pl, True)
                oracle.set_exact(False)
                assert _list(hm, data, pl, exact, 0) == want_m
                assert _list(hm, data, pl, exact, 1) == want_l
                assert hm.hm_count_lines(data.ctypes.data, data.size, pl, len(pl), exact, 64) == want_c
    finally:
        hm.hm_set_icase(0)
        oracle.set_exact(False)


def test_lane_mask_combine_equals_ordered_reduction(hm):
    """xsg::sum_combine_lanes (O(1) ballot algebra, used per wave-load on the GPU)
    == the ordered 64-way sum_combine of the same unit summaries."""
    assert hm.hm_check_lane_combine(1, 300000) == 0
    assert hm.hm_check_lane_combine(99, 300000) == 0


def test_span_summaries_combine_by_lane_masks(hm):
    """k_count_finish combines 64 per-wave tile summaries at a time with the same mask algebra: it must also hold
    for summaries of whole spans (closed-segment counts beyond a unit's 7, identity and preset entries)."""
    assert hm.hm_check_span_combine(5, 200000) == 0
    assert hm.hm_check_span_combine(77, 200000) == 0


def test_tail_walk_on_masks_equals_the_sequential_walk(hm):
    """xsg::tail_walk_masks (the zone's positions decided in parallel, the walk on bit masks: what a wave of
    k_count_finish runs) == xsg::tail_walk (byte by byte; itself checked against the oracle above)."""
    assert hm.hm_check_tail_masks(3, 400000) == 0
    assert hm.hm_check_tail_masks(1234, 400000) == 0
