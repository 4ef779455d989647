"""Randomised differential test of the HIP path against the oracle: random
alphabets, chunk sizes around every tile / wave / load boundary, random patterns
(random bytes, substrings of the data -- with newlines in them, up to a few KiB long --, self-overlapping ones), all
flags, all tags.
Fixed seeds: a failure reproduces."""
import numpy as np
import pytest

import xsg
from gpu_util import GpuSearch, oracle_all_modes

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 39, 40, 41, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097,
         16383, 16384, 16385, 16384 * 2 - 1, 16384 * 2 + 1, 16384 * 3 + 5]


def rand_pattern(rng, data, alphabet):
    kind = rng.integers(0, 6)
    if kind == 5 and data.size > 200:  # long substring: the filter window may sit anywhere in it
        n = int(rng.integers(9, 90))
        if data.size > 6000 and rng.random() < 0.15:  # beyond the KiB of the pattern the kernel keeps in LDS
            n = int(rng.integers(1025, 5000))
        o = int(rng.integers(0, data.size - n))
        p = data[o:o + n].tobytes()
    elif kind == 0 and data.size > 40:  # substring of the data
        n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 33]))
        o = int(rng.integers(0, data.size - n))
        p = data[o:o + n].tobytes()
    elif kind == 1:  # self-overlapping
        unit = bytes(alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(1, 4)))])
        p = (unit * 6)[:int(rng.integers(2, 13))]
    elif kind == 2:  # from the end of the data: lands in the reference's lossy tail zone
        n = int(rng.integers(2, 10))
        p = data[-n - int(rng.integers(0, 20)):][:n].tobytes() if data.size > 40 else b"ab"
    else:
        p = bytes(alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(1, 11)))])
    return p if p else b"a"


def fuzz_cases(seed, rounds=14, max_chunk=60000):
    """`rounds` random shards x 5 random patterns each -> (round, blocks, new shard?, pattern, exact, icase)"""
    rng = np.random.default_rng(1000 + seed)
    alphabets = [np.frombuffer(b"ab", dtype=np.uint8), np.frombuffer(b"ab\n", dtype=np.uint8),
                 np.frombuffer(b"abcAB \n\n", dtype=np.uint8), np.arange(256, dtype=np.uint8),
                 np.frombuffer(b"Sherlock Holmes\n", dtype=np.uint8)]
    words = [b"the ", b"detective ", b"street ", b"She", b"Sherlock ", b"Holmes ", b"a ", b"of the ", b"\n", b"B ", b"lock"]
    for it in range(rounds):
        alphabet = alphabets[int(rng.integers(0, len(alphabets)))]
        wordy = rng.random() < 0.3  # text made of a few words: long patterns with repeated parts
        nchunks = int(rng.integers(1, 7))
        blocks = []
        for _ in range(nchunks):
            n = int(rng.choice(SIZES)) if rng.random() < 0.7 else int(rng.integers(0, max_chunk))
            if wordy:
                idx = rng.integers(0, len(words), size=n // 3 + 1)
                b = np.frombuffer(b"".join(words[i] for i in idx), dtype=np.uint8)[:n].copy()
            else:
                b = alphabet[rng.integers(0, len(alphabet), size=n)].copy()
            if n and rng.random() < 0.6:
                b[-1] = 10
            blocks.append(b)
        big = max(blocks, key=lambda x: x.size)
        for k in range(5):
            p = rand_pattern(rng, big, alphabet)
            exact = bool(rng.integers(0, 2))
            icase = bool(rng.integers(0, 2))
            yield it, blocks, k == 0, p, exact, icase


def fuzz_rounds(seed, oracle, gs, rounds=14, max_chunk=60000):
    """every tag for every case of fuzz_cases on one context and shard; raises AssertionError on the first difference."""
    for it, blocks, fresh, p, exact, icase in fuzz_cases(seed, rounds, max_chunk):
        if fresh:
            gs.bind(blocks)
        flags = (xsg.FLAG_EXACT_TAIL if exact else 0) | (xsg.FLAG_IGNORE_CASE if icase else 0)
        got = gs.all_modes(p, flags)
        want = oracle_all_modes(oracle, blocks, p, exact, ignore_case=icase)
        for k in want:
            assert got[k] == want[k], (f"seed={seed} it={it} pat={p!r} exact={exact} icase={icase} "
                                       f"sizes={[b.size for b in blocks]} key={k}")


def first_call_rounds(seed, oracle, rounds=6):
    """The same cases, but every search is the FIRST call of a fresh context and binding, one random tag per case, the
    probe forced on: nothing an earlier tag allocated, measured or cached is there (round 4: a long pattern's first plain
    count stored through a null pointer -- gpu_util's all_modes begins with a count that wants newlines and hid it)."""
    import os
    from gpu_util import upload
    rng = np.random.default_rng(77 + seed)
    tags = ["count_matches", "count_lines", "match_byte_offsets", "line_byte_offsets", "line_indices", "lines"]
    old = os.environ.get("XSG_PROBE_MIN_BYTES")
    os.environ["XSG_PROBE_MIN_BYTES"] = "0"
    try:
        for it, blocks, fresh, p, exact, icase in fuzz_cases(seed, rounds):
            if sum(b.size for b in blocks) == 0:
                continue
            t, chunks = upload(blocks)
            key = tags[int(rng.integers(0, len(tags)))]
            flags = (xsg.FLAG_EXACT_TAIL if exact else 0) | (xsg.FLAG_IGNORE_CASE if icase else 0)
            want = oracle_all_modes(oracle, blocks, p, exact, ignore_case=icase)
            ctx = xsg.Context(0)
            sh = xsg.Shard(ctx, t.data_ptr(), t.numel(), chunks)
            ctx.set_pattern(p, flags)
            if key == "count_matches":
                got = int(sh.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES])
            elif key == "count_lines":
                got = int(sh.count(xsg.COUNT_LINES)[xsg.CTR_LINES])
            elif key == "lines":
                got = sh.search_lines()[0]
            else:
                got = sh.search_u64({"match_byte_offsets": xsg.MATCH_BYTE_OFFSETS, "line_byte_offsets": xsg.LINE_BYTE_OFFSETS,
                                     "line_indices": xsg.LINE_INDICES}[key]).tolist()
            sh.close()
            ctx.close()
            assert got == want[key], f"seed={seed} it={it} pat={p!r} exact={exact} icase={icase} key={key} (first call of a fresh binding)"
    finally:
        if old is None:
            os.environ.pop("XSG_PROBE_MIN_BYTES", None)
        else:
            os.environ["XSG_PROBE_MIN_BYTES"] = old


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_fuzz(seed, oracle):
    fuzz_rounds(seed, oracle, GpuSearch())


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_with_the_aligned_dword_trigger(seed, oracle):
    """the other hot filter of the 8-byte-window kinds (k_scan<..., ALIGNED = true>), which the library only picks
    by measurement on shards of 64 MiB and more: pinned here"""
    fuzz_rounds(seed, oracle, GpuSearch(hot=1))


@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_with_the_probe_choosing(seed, oracle):
    """hot filter and, for long patterns, the filter window picked by the library's measurement (on shards this small:
    at random) -- results must not depend on the choice"""
    fuzz_rounds(seed, oracle, GpuSearch(probe=True))


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_fuzz_where_every_search_is_a_first_call(seed, oracle):
    first_call_rounds(seed, oracle)
