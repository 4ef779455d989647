"""Parity of the HIP path (through the C ABI, libxsg.so) with the oracle and
with the committed golden vectors.  Bit-exact: this is integer/byte work.
Run on the GPU box: python -m pytest tests -m gpu"""
import numpy as np
import pytest

import corpus
import golden_util as G
import xsg
from gpu_util import GpuSearch, oracle_all_modes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gs():
    return GpuSearch()


def _b(s):
    return s.encode("latin-1")


def assert_same(got, want, ctx=""):
    for k in want:
        assert got[k] == want[k], f"{ctx}: {k} differs: got {str(got[k])[:200]} want {str(want[k])[:200]}"


def test_library_is_the_hip_one(gs):
    info = gs.ctx.info()
    assert info["arch"].startswith("gfx950"), info
    assert info["compute_units"] >= 200


def test_reference_unit_test_known_answers(gs):
    # test/src/string_search/simd_searchTest.cpp:83-99, search_wrappersTest.cpp:26-69
    ka = G.load("ref_simd_search_known_answers.json")
    text = np.frombuffer(_b(ka["text"]), dtype=np.uint8)
    gs.bind([text])
    for p, want in ka["countMatches"]:
        gs.ctx.set_pattern(_b(p))
        assert int(gs.shard.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES]) == want, p
    for p, want in ka["countMatchingLines"]:
        gs.ctx.set_pattern(_b(p))
        assert int(gs.shard.count(xsg.COUNT_LINES)[xsg.CTR_LINES]) == want, p
    for p, shift, want in ka["findNext"]:
        if shift == 0:
            gs.ctx.set_pattern(_b(p))
            offs = gs.shard.search_u64(xsg.MATCH_BYTE_OFFSETS).tolist()
            assert (offs[0] if offs else -1) == want, p
    # findNext / findNextNewLine with shift != 0 (simd_searchTest.cpp:62-81): the reference searches str + shift
    # for len - shift bytes (simd_search.cpp:289-303), i.e. the suffix as a chunk of its own -- its 32-byte blocks and
    # its scalar tail are anchored at the shift -- and adds the shift back
    for p, shift, want in ka["findNext"]:
        gs.bind([text[shift:].copy()])
        gs.ctx.set_pattern(_b(p))
        offs = gs.shard.search_u64(xsg.MATCH_BYTE_OFFSETS).tolist()
        assert (offs[0] + shift if offs else -1) == want, (p, shift)
    for shift, want in ka["findNextNewLine"]:
        gs.bind([text[shift:].copy()])
        gs.ctx.set_pattern(b"\n")
        offs = gs.shard.search_u64(xsg.MATCH_BYTE_OFFSETS).tolist()
        assert (offs[0] + shift if offs else -1) == want, shift
    kw = G.load("ref_search_wrappers_known_answers.json")
    gs.bind([np.frombuffer(_b(kw["text"]), dtype=np.uint8)])
    r = gs.all_modes(_b(kw["pattern"]))
    assert r["match_byte_offsets"] == kw["byte_offsets_match"]
    assert r["line_byte_offsets"] == kw["byte_offsets_line"]
    assert r["count_lines"] == kw["count"]
    assert r["lines"] == [_b(s) for s in kw["line"]]
    assert r["line_indices"] == [0, 2, 3, 8]


def test_golden_vectors_from_the_reference_build(gs):
    """tests/golden/ref_generated_vectors.json was produced by oracle/_ref
    (the reference's simd_search.cpp compiled unmodified)."""
    last = None
    n = 0
    for name, data, e in G.generated_cases():
        if name != last:
            gs.bind([data])
            last = name
        p = _b(e["pattern"])
        r = gs.all_modes(p)
        ctx = f"{name} pat={e['pattern']!r}"
        assert r["count_matches"] == e["countMatches"], ctx
        assert r["match_byte_offsets"] == e["byte_offsets_match"], ctx
        if "byte_offsets_line" in e:
            assert r["count_lines"] == e["countMatchingLines"], ctx
            assert r["line_byte_offsets"] == e["byte_offsets_line"], ctx
            assert r["line_indices"] == e["line_indices"], ctx
            assert r["lines_offsets"] == e["lines_begin"], ctx
            assert [len(x) for x in r["lines"]] == e["lines_len"], ctx
            assert r["lines"] == [data[b:b + l].tobytes() for b, l in zip(e["lines_begin"], e["lines_len"])], ctx
        n += 1
    assert n > 100


def test_golden_ignore_case_vectors_from_the_reference_build(gs):
    """The ignore_case part of ref_generated_vectors.json: expected values produced by the reference's own
    simd::strcasestr under the wrapper loops; the GPU runs with XSG_FLAG_IGNORE_CASE on the original bytes."""
    last, n = None, 0
    for name, data, e in G.generated_icase_cases():
        if name != last:
            gs.bind([data])
            last = name
        r = gs.all_modes(_b(e["pattern"]), xsg.FLAG_IGNORE_CASE)
        ctx = f"{name} pat={e['pattern']!r}"
        assert r["count_matches"] == e["count_noskip"], ctx
        assert r["count_lines"] == e["count_skip"], ctx
        assert r["match_byte_offsets"] == e["byte_offsets_match"], ctx
        assert r["line_byte_offsets"] == e["byte_offsets_line"], ctx
        assert r["line_indices"] == e["line_indices"], ctx
        assert r["lines_offsets"] == e["lines_begin"], ctx
        assert r["lines"] == [data[b:b + l].tobytes() for b, l in zip(e["lines_begin"], e["lines_len"])], ctx  # original case
        n += 1
    assert n >= 40


PATTERNS = [b"a", b"ab", b"aa", b"aba", b"abab", b"bab", b"abc ", b"ab ab", b"abcabca", b"abababab", b"ab ab ab a",
            b"a" * 17, b"b\na"]


@pytest.mark.parametrize("exact", [False, True])
def test_random_multi_chunk_shards_vs_oracle(gs, oracle, exact):
    rng = np.random.default_rng(4242 + int(exact))
    alph = np.frombuffer(b"ab \nc", dtype=np.uint8)
    flags = xsg.FLAG_EXACT_TAIL if exact else 0
    for it in range(12):
        nchunks = int(rng.integers(1, 6))
        blocks = []
        for _ in range(nchunks):
            n = int(rng.choice([0, 1, 15, 16, 17, 1023, 1024, 1025, 4096, 16383, 16384, 16385, 40000,
                                int(rng.integers(2, 70000))]))
            blocks.append(alph[rng.integers(0, 3 + it % 3, size=n)].copy())
        gs.bind(blocks)
        for p in PATTERNS:
            got = gs.all_modes(p, flags)
            want = oracle_all_modes(oracle, blocks, p, exact)
            assert_same(got, want, f"it={it} exact={exact} pat={p!r} sizes={[b.size for b in blocks]}")


def test_text_blocks_all_pattern_kinds(gs, oracle):
    blocks = [corpus.text_block(11, i, 300_000 + 4111 * i, needle_rate=3e-4) for i in range(5)]
    blocks[2] = blocks[2][:-1].copy()  # an unterminated chunk end
    long_pat = bytes(blocks[1][1000:1110])
    assert b"\n" not in long_pat or True
    gs.bind(blocks)
    pats = [b"e", b"th", b"the", b"that", b"She", b"lock", b"Sherl", b"Holmes", b"Sherlock", b"detective",
            b"detective street", b"Sherlock Holmes", b"\x00", b"\xff\xfe", bytes(blocks[0][77:77 + 33])]
    if b"\n" not in long_pat:
        pats.append(long_pat)
    for exact in (False, True):
        for p in pats:
            got = gs.all_modes(p, xsg.FLAG_EXACT_TAIL if exact else 0)
            want = oracle_all_modes(oracle, blocks, p, exact)
            assert_same(got, want, f"exact={exact} pat={p[:20]!r}")


@pytest.mark.parametrize("exact", [False, True])
def test_dense_needles_of_4_to_8_bytes_take_the_byte_parallel_route(gs, oracle, exact):
    """A 4..8-byte needle that a first count finds dense (more than one result per 2 KiB) is decided byte-parallel from
    then on (k_scan<0,...>, 'dense: byte-parallel' in the kernel name): every tag, both tail modes, against the oracle."""
    blocks = [corpus.text_block(77, i, n, needle=b"Sherlock", needle_rate=0.03) for i, n in enumerate((70001, 16384, 33, 250000))]
    blocks.append(corpus.small_alphabet(5, 40000, b"abc \n", terminate=True))
    gs.bind(blocks)
    for pat in (b"that", b"which", b"Holmes", b"locked ", b"Sherlock", b"abca", b"bcabc", b"aaaa", b"abababab", b"was "):
        flags = xsg.FLAG_EXACT_TAIL if exact else 0
        got = gs.all_modes(pat, flags)
        want = oracle_all_modes(oracle, blocks, pat, exact=exact)
        assert_same(got, want, f"exact={exact} pat={pat!r}")
        if got["count_matches"] * 2048 > sum(b.size for b in blocks):
            assert "byte-parallel" in gs.shard.scan_kernel_name(xsg.COUNT_MATCHES), pat
    # ignore_case: a needle of letters needs no fold ((x | 0x20) == p is exact) and takes the route as well; one with
    # another byte in it keeps the hot filter and its properly folded slow path
    for pat, routed in ((b"That", True), (b"HOLMES", True), (b"sHeRlOcK", True), (b"was ", False), (b"E", None), (b"tH", None), (b"She", None), (b"a\n"[:1] + b" ", None)):
        flags = xsg.FLAG_IGNORE_CASE | (xsg.FLAG_EXACT_TAIL if exact else 0)
        got = gs.all_modes(pat, flags)
        assert_same(got, oracle_all_modes(oracle, blocks, pat, exact=exact, ignore_case=True), f"ignore_case exact={exact} pat={pat!r}")
        if routed is not None and got["count_matches"] * 2048 > sum(b.size for b in blocks):
            assert ("byte-parallel" in gs.shard.scan_kernel_name(xsg.COUNT_MATCHES)) == routed, pat


@pytest.mark.parametrize("exact", [False, True])
def test_bordered_patterns_whose_occurrences_do_not_overlap_in_the_data(gs, oracle, exact):
    """`that` has a border (t...t) and could overlap itself ("thathat") -- in text it does not, and the library finds that
    out once per binding (one count pass per border for the word two overlapping occurrences would spell) and then
    counts and lists it like a pattern without a border.  Both outcomes, several borders, overlaps only in one chunk,
    only across the end-of-chunk zone, under ignore_case; every tag against the oracle, and the async entry point."""
    import torch
    text = [corpus.text_block(31, i, n, needle_rate=0.01) for i, n in enumerate((90000, 16384 * 3 + 5, 40000))]
    glued = np.frombuffer(b"so thathat is that and thathathat too\n" * 3, dtype=np.uint8)
    tail_pair = np.concatenate([text[2][:-1], np.frombuffer(b" thathat", dtype=np.uint8)])  # unterminated, overlap at the very end
    cases = {
        "no overlap": text,
        "overlaps in one chunk": [text[0], np.concatenate([text[1][:30000], glued, text[1][30000:]]), text[2]],
        "overlap in the end zone": [text[0], tail_pair],
    }
    flags = xsg.FLAG_EXACT_TAIL if exact else 0
    dc = torch.zeros(xsg.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    for name, blocks in cases.items():
        gs.bind(blocks)
        for pat in (b"that", b"else", b"stats", b"abcab", b"aXa", b"was w", b"tt", b"hath", b"e the"):
            got = gs.all_modes(pat, flags)
            want = oracle_all_modes(oracle, blocks, pat, exact=exact)
            assert_same(got, want, f"{name} exact={exact} pat={pat!r}")
            gs.shard.count_async(xsg.COUNT_MATCHES, 0, dc.data_ptr())  # uses what the synchronous calls established
            torch.cuda.synchronize()
            assert int(dc[xsg.CTR_MATCHES]) == want["count_matches"], (name, pat)
        got = gs.all_modes(b"ThAt", flags | xsg.FLAG_IGNORE_CASE)
        assert_same(got, oracle_all_modes(oracle, blocks, b"ThAt", exact=exact, ignore_case=True), f"{name} ignore_case")


def test_global_offsets_and_explicit_line_bases(gs, oracle):
    blocks = [corpus.text_block(5, i, 70_000, needle_rate=1e-3) for i in range(3)]
    goffs = [1_000_000, 5_000_000_000, 5_000_070_000]
    lbases = [10, 1_000_000, 1_000_000 + 7]
    gs.bind(blocks, goffs, lbases)
    got = gs.all_modes(b"Sherlock")
    want = oracle_all_modes(oracle, blocks, b"Sherlock", False, goffs, lbases)
    assert_same(got, want)
    # AUTO bases + a shard base (what a rank with preceding shards would set)
    gs.bind(blocks)
    gs.shard.set_line_base(12345)
    gs.ctx.set_pattern(b"Sherlock")
    got = gs.shard.search_u64(xsg.LINE_INDICES).tolist()
    want = oracle_all_modes(oracle, blocks, b"Sherlock")["line_indices"]
    assert got == [x + 12345 for x in want]
    gs.shard.set_line_base(0)


def test_empty_and_degenerate_shards(gs, oracle):
    z = np.zeros(0, dtype=np.uint8)
    for blocks in ([], [z], [z, z], [np.frombuffer(b"x", dtype=np.uint8)], [np.frombuffer(b"Sherlock", dtype=np.uint8)]):
        gs.bind(blocks)
        for p in (b"x", b"Sherlock", b"Sherlock Holmes"):
            assert_same(gs.all_modes(p), oracle_all_modes(oracle, blocks, p), f"{[b.size for b in blocks]} {p!r}")


def test_argument_errors(gs):
    with pytest.raises(xsg.XsgError) as e:
        gs.ctx.set_pattern(b"")
    assert e.value.code == xsg.EINVAL
    with pytest.raises(xsg.XsgError):
        gs.ctx.set_pattern(b"x" * (xsg.MAX_PATTERN + 1))
    # (round 3 refused the line tags for a literal that contains '\n' and patterns over 1 KiB; both are answered now:
    # test_line_tags_of_literals_that_contain_a_newline, test_patterns_up_to_32_kib)
    gs.bind([np.frombuffer(b"a\nb\n", dtype=np.uint8)])
    gs.ctx.set_pattern(b"a\n")
    assert gs.shard.search_u64(xsg.LINE_BYTE_OFFSETS).tolist() == [0]
    gs.ctx.set_pattern(b"a[^x]b", xsg.FLAG_REGEX)  # an EXPRESSION that can match '\n' keeps to the match tags
    with pytest.raises(xsg.XsgError) as e:
        gs.shard.search_u64(xsg.LINE_BYTE_OFFSETS)
    assert e.value.code == xsg.ENOTSUP
    with pytest.raises(xsg.XsgError):
        gs.ctx.set_pattern(b"x" * (xsg.MAX_REGEX + 1), xsg.FLAG_REGEX)
    # misaligned chunk offset
    import torch
    t = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
    bad = xsg.make_chunks([8], [100])
    with pytest.raises(xsg.XsgError) as e:
        xsg.Shard(gs.ctx, t.data_ptr(), t.numel(), bad)
    assert e.value.code == xsg.EINVAL
    bad = xsg.make_chunks([0], [5000])
    with pytest.raises(xsg.XsgError):
        xsg.Shard(gs.ctx, t.data_ptr(), t.numel(), bad)


def test_async_count_on_a_caller_stream(gs, oracle):
    import torch
    blocks = [corpus.text_block(21, i, 1 << 20, needle_rate=2e-4) for i in range(4)]
    gs.bind(blocks)
    gs.ctx.set_pattern(b"Sherlock")
    ctr = torch.zeros(xsg.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gs.shard.count_async(xsg.COUNT_MATCHES | xsg.WITH_NEWLINES, st.cuda_stream, ctr.data_ptr())
        gs.shard.count_async(xsg.COUNT_MATCHES | xsg.WITH_NEWLINES, st.cuda_stream, ctr.data_ptr())
    st.synchronize()
    want = oracle_all_modes(oracle, blocks, b"Sherlock")
    got = ctr.cpu().numpy().astype(np.uint64)
    assert int(got[xsg.CTR_MATCHES]) == want["count_matches"]
    assert int(got[xsg.CTR_NEWLINES]) == want["newlines"]
    assert int(got[xsg.CTR_BYTES]) == want["bytes"]
    # a pattern that can overlap itself: the match count needs the greedy walk -- served on the device too
    # (bounded list route); only the newline count next to it is the synchronous call's
    gs.ctx.set_pattern(b"abab")
    with pytest.raises(xsg.XsgError) as e:
        gs.shard.count_async(xsg.COUNT_MATCHES | xsg.WITH_NEWLINES, 0, ctr.data_ptr())
    assert e.value.code == xsg.ENOTSUP


def test_async_count_of_patterns_that_overlap_themselves(gs, oracle):
    """xsg_count_async(XSG_COUNT_MATCHES) for `aa`, `abab`, `that`, `[ab]{3}`: the greedy non-overlap count
    (simd_search.cpp:333) without a trip to the host == the oracle == the synchronous route; more raw occurrences
    than the arrays hold -> every counter UINT64_MAX, and the synchronous call teaches the shard the size."""
    import torch
    rng = np.random.default_rng(77)
    ab = np.frombuffer(b"ab\n", dtype=np.uint8)
    blocks = [ab[rng.integers(0, 3, size=n)].copy() for n in (70_000, 16384, 16385, 33, 0, 200_001)]
    blocks.append(np.frombuffer(b"that thathat thathathat\nthat\n" * 3000, dtype=np.uint8).copy())
    gs.bind(blocks)
    ctr = torch.zeros(xsg.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    st = torch.cuda.Stream()
    for pat, flags in ((b"aa", 0), (b"abab", 0), (b"aba", xsg.FLAG_EXACT_TAIL), (b"that", 0), (b"THAT", xsg.FLAG_IGNORE_CASE),
                       (b"[ab]{3}", xsg.FLAG_REGEX), (b"a", 0)):
        gs.ctx.set_pattern(pat, flags)
        for _ in range(2):  # twice: the pass must leave the shard as it found it
            gs.shard.count_async(xsg.COUNT_MATCHES, st.cuda_stream, ctr.data_ptr())
            st.synchronize()
            got = ctr.cpu().numpy().astype(np.uint64)
            want = gs.shard.count(xsg.COUNT_MATCHES)
            assert int(got[xsg.CTR_MATCHES]) == int(want[xsg.CTR_MATCHES]), pat
            assert int(got[xsg.CTR_BYTES]) == sum(b.size for b in blocks)
        if not flags:
            assert int(got[xsg.CTR_MATCHES]) == sum(oracle.count(b, pat, False) for b in blocks), pat
    # 2.5 M raw occurrences of `aa` against a capacity of 2^20: refused; after xsg_count the arrays are large enough
    run = np.full(2_500_000, ord("a"), dtype=np.uint8)
    gs.bind([run])
    gs.ctx.set_pattern(b"aa")
    gs.shard.count_async(xsg.COUNT_MATCHES, st.cuda_stream, ctr.data_ptr())
    st.synchronize()
    assert all(int(x) == -1 for x in ctr.cpu())
    want = int(gs.shard.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES])
    assert want == oracle.count(run, b"aa", False)
    gs.shard.count_async(xsg.COUNT_MATCHES, st.cuda_stream, ctr.data_ptr())
    st.synchronize()
    assert int(ctr[xsg.CTR_MATCHES]) == want


@pytest.mark.parametrize("exact", [False, True])
def test_ignore_case(gs, oracle, exact):
    """XSG_FLAG_IGNORE_CASE == search(toLower(chunk), toLower(pattern))
    (simd::toLower, src/utils/string_utils.cpp:11-33); lines keep original bytes."""
    rng = np.random.default_rng(2718 + int(exact))
    flags = xsg.FLAG_IGNORE_CASE | (xsg.FLAG_EXACT_TAIL if exact else 0)
    alph = np.frombuffer(b"aAbB \ncC", dtype=np.uint8)
    blocks = [alph[rng.integers(0, len(alph), size=int(n))].copy() for n in (5000, 16385, 40000, 33)]
    gs.bind(blocks)
    for p in (b"a", b"Ab", b"aBa", b"abAB", b"b A", b"ABCabc", b"aAbBcCaA", b"cab cab ab", b"a" * 20):
        got = gs.all_modes(p, flags)
        want = oracle_all_modes(oracle, blocks, p, exact, ignore_case=True)
        assert_same(got, want, f"icase exact={exact} pat={p!r}")
    # realistic text with upper-cased needles and non-ASCII bytes (must not fold)
    tb = [corpus.text_block(91, i, 500_000, needle_rate=5e-4) for i in range(3)]
    for b in tb:
        pos = rng.integers(0, b.size - 64, size=200)
        for q in pos:
            seg = b[q:q + 40]
            up = (seg >= 97) & (seg <= 122)
            seg[up] -= 32
        b[rng.integers(0, b.size, size=500)] = rng.integers(128, 256, size=500).astype(np.uint8)
        b[-1] = 10
    gs.bind(tb)
    for p in (b"sherlock", b"SHERLOCK", b"Sherlock Holmes", b"tHe", b"E", b"\xc0\xe0"):
        got = gs.all_modes(p, flags)
        want = oracle_all_modes(oracle, tb, p, exact, ignore_case=True)
        assert_same(got, want, f"icase text exact={exact} pat={p!r}")
        assert want["count_matches"] >= oracle_all_modes(oracle, tb, p, exact)["count_matches"]


def test_many_tiny_chunks(gs, oracle):
    """20 000 chunks of 0..600 bytes: every chunk is mostly 'tail zone', tiles hold
    one chunk each, the finish kernels loop over many chunks per wave."""
    rng = np.random.default_rng(606)
    alph = np.frombuffer(b"ab \nSherlock", dtype=np.uint8)
    sizes = rng.integers(0, 600, size=20000)
    blocks = [alph[rng.integers(0, len(alph), size=int(n))].copy() for n in sizes]
    gs.bind(blocks)
    for p in (b"ab", b"Sher", b"ba b", b"lock\nS"):
        for exact in (False, True):
            got = gs.all_modes(p, xsg.FLAG_EXACT_TAIL if exact else 0, lines=(p != b"ba b"))
            want = oracle_all_modes(oracle, blocks, p, exact)
            if p == b"ba b":
                want = {k: want[k] for k in ("count_matches", "newlines", "bytes", "match_byte_offsets")}
            assert_same(got, want, f"tiny chunks pat={p!r} exact={exact}")


def test_all_byte_values_and_long_patterns(gs, oracle):
    rng = np.random.default_rng(77)
    data = rng.integers(0, 256, size=300_000).astype(np.uint8)
    # plant long patterns (incl. the maximum length) across tile / wave / load boundaries
    pats = [bytes(rng.integers(0, 256, size=n).astype(np.uint8)).replace(b"\n", b"\x0b") for n in (9, 16, 17, 64, 1000, 1024)]
    for k, p in enumerate(pats):
        for pos in (16384 * (k + 1) - len(p) // 2, 4096 * (k + 3) - 3, 1024 * (k + 40) - 1, 200_000 + 2000 * k):
            data[pos:pos + len(p)] = np.frombuffer(p, dtype=np.uint8)
    data[-len(pats[0]):] = np.frombuffer(pats[0], dtype=np.uint8)  # a match ending exactly at the chunk end
    blocks = [data[:150_001].copy(), data[150_001:].copy()]
    gs.bind(blocks)
    for p in pats + [b"\x00", b"\xff\xff", bytes([0x80, 0x41, 0xc3])]:
        for exact in (False, True):
            got = gs.all_modes(p, xsg.FLAG_EXACT_TAIL if exact else 0)
            want = oracle_all_modes(oracle, blocks, p, exact)
            assert_same(got, want, f"bytes pat[{len(p)}] exact={exact}")
    assert oracle_all_modes(oracle, blocks, pats[4], True)["count_matches"] >= 3


def test_one_big_chunk_with_long_lines(gs, oracle):
    """A single 40 MB chunk whose lines are up to 3 MB long (line starts far from
    their matches; many tiles without any newline)."""
    rng = np.random.default_rng(5150)
    b = corpus.text_block(808, 0, 40_000_000, needle_rate=2e-4)
    nl = np.flatnonzero(b == 10)
    keep = set(rng.choice(nl[:-1], size=60, replace=False).tolist()) | {int(nl[-1])}
    kill = np.array([i for i in nl if int(i) not in keep])
    b[kill] = 32
    gs.bind([b])
    for p in (b"Sherlock", b"Holmes", b"q"):
        got = gs.all_modes(p)
        want = oracle_all_modes(oracle, [b], p)
        assert_same(got, want, f"long lines pat={p!r}")


def test_one_huge_line_with_many_matches(gs, oracle):
    """8 MB without a single newline and ~60 000 matches in it (then a few normal
    lines): per-match work must stay bounded by the gap to the previous match."""
    rng = np.random.default_rng(99)
    words = [b"alpha ", b"beta ", b"needle ", b"gamma ", b"delta "]
    idx = rng.integers(0, len(words), size=1_400_000)
    big = np.frombuffer(b"".join(words[i] for i in idx), dtype=np.uint8).copy()
    tail = np.frombuffer(b"\nshort needle line\nno match here\nneedle at end", dtype=np.uint8)
    b = np.concatenate([big, tail])
    assert (big == 10).sum() == 0 and b.size > 7_000_000
    gs.bind([b])
    import time
    t0 = time.perf_counter()
    got = gs.all_modes(b"needle")
    dt = time.perf_counter() - t0
    want = oracle_all_modes(oracle, [b], b"needle")
    assert_same(got, want, "huge line")
    assert want["count_matches"] > 50_000 and want["count_lines"] == 3  # the huge line, the short one, the unterminated last one
    assert dt < 5.0, f"huge-line search took {dt:.1f} s"


def test_long_patterns_with_a_shifted_filter_window(gs, oracle):
    """Long patterns are filtered on their rarest 8-byte window, not their first 8 bytes
    (xsg_api.cpp pick_filter_window): a match is then found where its WINDOW lies.  Plant
    matches so that start and window fall into different units / wave-loads / tiles,
    at the very start of a chunk, closer to it than the window offset, and around the
    tail zone; add decoys that hold the window but not the rest."""
    rng = np.random.default_rng(4242)
    pats = [b"detective street", b"aaaaaaaaaaaaaaaaaaaa B aaaaaaaaaaaaaaaaaa", b"the quick brown fox jumps over",
            b"abcabcabcabc abcabcabcabc", b"ZaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaQ z"]
    for p in pats:
        n = len(p)
        words = [b"street ", b"detective ", b"aaaa", b" B ", b"fox ", b"abc", b"the quick ", b"\n", b"ective s", b"a", b"Q z"]
        idx = rng.integers(0, len(words), size=40_000)
        data = np.frombuffer(b"".join(words[i] for i in idx), dtype=np.uint8).copy()[:200_000]
        pa = np.frombuffer(p, dtype=np.uint8)
        starts = [0, 1, 7, 15, 16]
        for t in (16384, 32768, 65536, 4096 * 5, 1024 * 37):
            starts += [t - j for j in (0, 1, 7, 8, 9, 15, 16, 17, n - 1, n, n + 1, n // 2) if t - j > 200]
        ends = [data.size - n - d for d in (0, 1, 5, 30, 31, 32, 33, 40, 64, 100)]  # inside and around the tail zone
        for s in sorted(set(starts)):
            data[s:s + n] = pa
        blocks = [data.copy(), data[3:90_000].copy(), data[:n + 3].copy(), data[:n].copy(), data[1:n].copy()]
        for e in ends:
            blk = data.copy()
            blk[e:e + n] = pa
            blocks.append(blk[100_000:].copy())
        gs.bind(blocks)
        for exact in (False, True):
            for icase in (False, True):
                flags = (xsg.FLAG_EXACT_TAIL if exact else 0) | (xsg.FLAG_IGNORE_CASE if icase else 0)
                got = gs.all_modes(p, flags)
                want = oracle_all_modes(oracle, blocks, p, exact, ignore_case=icase)
                assert_same(got, want, f"shifted window pat={p!r} exact={exact} icase={icase}")
        assert oracle_all_modes(oracle, blocks, p, True)["count_matches"] > 15


def test_a_64_mib_run_of_one_byte_is_resolved_in_parallel(gs, oracle):
    """64 MiB of `a` searched for `aa` / `aaaa`: 67 million raw occurrences in ONE chain.  The chain head's thread
    gives up after its budget and the chain is finished by pointer jumping over next-reported links (k_greedy_links /
    k_greedy_jump, ~25 rounds); the list must equal the reference's sequential walk (simd_search.cpp:324-336,
    search_wrappers.h:29-52), end-of-chunk behaviour included, and come back in well under the seconds one lane
    would need."""
    import time
    n = 64 << 20
    run = np.full(n + 1, ord("a"), dtype=np.uint8)
    run[-1] = 10
    blocks = [run, np.concatenate([np.frombuffer(b"xa", dtype=np.uint8), run[:5_000_001], np.frombuffer(b"b aa\n", dtype=np.uint8)])]
    gs.bind(blocks)
    for pat in (b"aa", b"aaaa"):
        gs.ctx.set_pattern(pat)
        gs.shard.search_u64_view(xsg.MATCH_BYTE_OFFSETS)  # warm: buffers
        t0 = time.perf_counter()
        got = gs.shard.search_u64_view(xsg.MATCH_BYTE_OFFSETS)
        dt = time.perf_counter() - t0
        want = np.concatenate([oracle.byte_offsets_match(blocks[0], pat), oracle.byte_offsets_match(blocks[1], pat) + np.uint64(blocks[0].size)])
        assert got.size == want.size >= n // len(pat)
        assert np.array_equal(got, want), pat
        assert int(gs.shard.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES]) == want.size
        assert dt < 1.0, dt


def test_half_a_gigabyte_without_a_newline(gs, oracle):
    """One 512 MiB chunk that is a single unterminated line, needles at both ends and in the middle, then the
    same with one newline in the middle: the line tags have to find a line start / line end half a gigabyte
    away from the match.  A lane looks 4 KiB far on its own, then the whole wave finishes the search 4 KiB a step
    (newline_query in xsg_kernels.hip): 2 s for all tags here; a byte loop in one thread took minutes."""
    import time
    n = 512 << 20
    b = np.full(n, ord("x"), dtype=np.uint8)
    b[::97] = 32
    for pos in (5, n // 2 + 3, n - 4000):
        b[pos:pos + 8] = np.frombuffer(b"Sherlock", dtype=np.uint8)
    for with_nl in (False, True):
        if with_nl:
            b[n // 2 - 1000] = 10
        gs.bind([b])
        t0 = time.perf_counter()
        got = gs.all_modes(b"Sherlock")
        dt = time.perf_counter() - t0
        want = oracle_all_modes(oracle, [b], b"Sherlock")
        for k in ("count_matches", "count_lines", "match_byte_offsets", "line_byte_offsets", "line_indices", "lines_offsets"):
            assert got[k] == want[k], (with_nl, k)
        assert [len(x) for x in got["lines"]] == [len(x) for x in want["lines"]], with_nl
        assert want["count_matches"] == 3 and want["count_lines"] == (2 if with_nl else 1)
        assert dt < 10.0, f"single-line chunk took {dt:.1f} s"
        print(f"512 MiB single line (newline in the middle: {with_nl}): all tags in {dt:.2f} s")


def test_one_long_chain_of_overlapping_occurrences(gs, oracle):
    """A megabyte-long run of one byte searched for `aa` / `aaa` is ONE chain of overlapping occurrences per chunk:
    the greedy non-overlap walk (simd_search.cpp:333, search_wrappers.h:42) keeps every 2nd / 3rd of a million
    candidates (k_greedy_keep walks it eight entries a fetch)."""
    import time
    run = np.full(1 << 20, ord("a"), dtype=np.uint8)
    blocks = [np.concatenate([run, np.frombuffer(b"\n", dtype=np.uint8)]),
              np.concatenate([np.frombuffer(b"xa", dtype=np.uint8), run[:300_001], np.frombuffer(b"b\naa\n", dtype=np.uint8)])]
    gs.bind(blocks)
    for pat in (b"aa", b"aaa", b"aaaaaaaaa"):
        t0 = time.perf_counter()
        got = gs.all_modes(pat)
        dt = time.perf_counter() - t0
        assert_same(got, oracle_all_modes(oracle, blocks, pat), f"run of a, {pat!r}")
        assert got["count_matches"] >= (1 << 20) // len(pat)
        assert dt < 5.0, dt


def test_line_tags_of_literals_that_contain_a_newline(gs, oracle):
    """search_wrappers.h:29-50,163-207 for a pattern with '\\n' in it: the line walk is a chain from occurrence to
    occurrence (after a match, on to the first newline at or behind its END), lines span several text lines, a pattern
    that begins with '\\n' has its line start behind the match's first byte (:111-123).  All tags, both tail modes,
    ignore_case, chunks with and without a final newline, long chains (every line of a chunk)."""
    rng = np.random.default_rng(99)
    blocks = []
    for i in range(5):
        n = int(rng.choice([0, 1, 50, 4095, 4096, 16384, 40000, 70001]))
        b = np.frombuffer(b"ab\n", dtype=np.uint8)[rng.integers(0, 3, size=n)].copy()
        if n and i % 2:
            b[-1] = 10
        blocks.append(b)
    blocks.append(corpus.text_block(5, 0, 300_000, needle_rate=3e-4))
    blocks.append(np.frombuffer(b"Sherlock\n" * 20000, dtype=np.uint8).copy())  # one chain through 20000 lines
    for exact in (False, True):
        gs.bind(blocks)
        for pat in (b"a\n", b"\na", b"\n", b"\n\n", b"b\nb", b"ab\nab", b"\nab\n", b"k\nS", b"\nSherlock", b"Sherlock\nSherlock", b"e\nthe"):
            got = gs.all_modes(pat, xsg.FLAG_EXACT_TAIL if exact else 0)
            assert_same(got, oracle_all_modes(oracle, blocks, pat, exact), f"{pat!r} exact={exact}")
    gs.bind(blocks)
    for pat in (b"K\ns", b"\nSHERLOCK"):
        assert_same(gs.all_modes(pat, xsg.FLAG_IGNORE_CASE), oracle_all_modes(oracle, blocks, pat, ignore_case=True), f"{pat!r} icase")
    # the split-phase count (what the file pipeline calls) serves count_lines of such a pattern synchronously
    gs.ctx.set_pattern(b"k\nS")
    gs.shard.count_begin(xsg.COUNT_LINES)
    assert int(gs.shard.count_end()[xsg.CTR_LINES]) == oracle_all_modes(oracle, blocks, b"k\nS")["count_lines"]
    import torch
    ctr = torch.zeros(xsg.NUM_COUNTERS + 1, dtype=torch.int64, device="cuda:0")
    with pytest.raises(xsg.XsgError) as e:  # the stream-ordered entry point may not wait for the host: it says so
        gs.shard.count_async(xsg.COUNT_LINES, 0, ctr.data_ptr())
    assert e.value.code == xsg.ENOTSUP


def test_patterns_up_to_32_kib(gs, oracle):
    """XSG_MAX_PATTERN is 32 KiB since round 4 (the reference takes any std::string): the scan kernel keeps the first KiB
    of a long pattern in LDS and verifies the rest of a candidate from the pattern's device copy; the end-of-chunk walk
    reads it from there anyway."""
    rng = np.random.default_rng(5)
    base = corpus.text_block(31, 0, 400_000, needle_rate=1e-4)
    for n in (1025, 2000, 5000, 20000, xsg.MAX_PATTERN):
        o = int(rng.integers(0, base.size - n))
        pat = base[o:o + n].tobytes()
        # the pattern occurs where it was cut from; planted again twice, once in the chunk's last bytes (the lossy tail zone)
        blocks = [base.copy(), np.concatenate([base[:100_000], base[o:o + n], base[100_000:150_000], base[o:o + n], base[:17]]),
                  base[o:o + n].copy(), base[o:o + n - 1].copy()]
        gs.bind(blocks)
        for exact in (False, True):
            got = gs.all_modes(pat, xsg.FLAG_EXACT_TAIL if exact else 0)
            want = oracle_all_modes(oracle, blocks, pat, exact)
            assert want["count_matches"] >= 3
            assert_same(got, want, f"plen={n} exact={exact}")
        near = pat[:-1] + bytes([pat[-1] ^ 1])  # differs in its last byte: every candidate is verified to the end and dropped
        assert_same(gs.all_modes(near), oracle_all_modes(oracle, blocks, near), f"plen={n} near miss")


def test_async_count_with_a_status_word(gs, oracle):
    """xsg_count_async_status: what xsg_count_async reports by poisoning all four counters (UINT64_MAX) arrives as a
    status word, and the counters read zero -- a caller that sums or all-reduces them cannot mistake it for a count."""
    import torch
    blocks = [corpus.text_block(21, i, 1 << 20, needle_rate=2e-4) for i in range(3)]
    gs.bind(blocks)
    buf = torch.full((xsg.NUM_COUNTERS + 1,), 77, dtype=torch.int64, device="cuda:0")
    st = torch.cuda.Stream()
    for pat, flags, mode in ((b"Sherlock", 0, xsg.COUNT_MATCHES | xsg.WITH_NEWLINES), (b"the", 0, xsg.COUNT_LINES), (b"abab", 0, xsg.COUNT_MATCHES)):
        gs.ctx.set_pattern(pat, flags)
        gs.shard.count_async_status(mode, st.cuda_stream, buf.data_ptr(), buf.data_ptr() + 8 * xsg.NUM_COUNTERS)
        st.synchronize()
        got = buf.cpu().numpy().astype(np.uint64)
        assert int(got[xsg.NUM_COUNTERS]) == xsg.STATUS_OK
        want = gs.shard.count(mode)
        assert [int(x) for x in got[:xsg.NUM_COUNTERS]] == [int(x) for x in want], pat
    # more raw occurrences of `aa` than the device-side list holds: status OVERFLOW, counters zero
    run = np.full(2_500_000, ord("a"), dtype=np.uint8)
    gs2 = GpuSearch()
    gs2.bind([run])
    gs2.ctx.set_pattern(b"aa")
    gs2.shard.count_async_status(xsg.COUNT_MATCHES, st.cuda_stream, buf.data_ptr(), buf.data_ptr() + 8 * xsg.NUM_COUNTERS)
    st.synchronize()
    got = buf.cpu().numpy().astype(np.uint64)
    assert int(got[xsg.NUM_COUNTERS]) == xsg.STATUS_OVERFLOW and not got[:xsg.NUM_COUNTERS].any()
    # an ascii-only expression on data with a byte >= 0x80: status NONASCII, counters zero
    gs2.bind([np.frombuffer("caf\u00e9 Sherlock\n".encode() * 50, dtype=np.uint8)])
    gs2.ctx.set_pattern(b"S.erlock", xsg.FLAG_REGEX)
    gs2.shard.count_async_status(xsg.COUNT_MATCHES, st.cuda_stream, buf.data_ptr(), buf.data_ptr() + 8 * xsg.NUM_COUNTERS)
    st.synchronize()
    got = buf.cpu().numpy().astype(np.uint64)
    assert int(got[xsg.NUM_COUNTERS]) == xsg.STATUS_NONASCII and not got[:xsg.NUM_COUNTERS].any()


def test_first_search_of_a_fresh_binding_with_a_long_pattern(oracle, monkeypatch):
    """A long pattern settles its filter window in the probe of its FIRST pass on a binding, on the newline-counting
    instantiation -- whose per-tile array a plain count never asked for (round 4: a null store, found by
    scripts/first_call.py on a 50 GiB shard; gpu_util's all_modes begins with a count that wants newlines and hid it).
    Fresh context and binding per tag, the probe forced on a small shard."""
    from gpu_util import upload
    monkeypatch.setenv("XSG_PROBE_MIN_BYTES", "0")
    blocks = [corpus.text_block(11, i, 400_000 + 31 * i, needle_rate=3e-4) for i in range(3)]
    t, chunks = upload(blocks)
    from gpu_util import oracle_regex_all_modes
    for pat, flags in ((b"detective street", 0), (b"Sherlock Holmes said", 0), (b"Sherlock", 0), (b"the", 0), (b"Holmes", 0),
                       (b"sherlock", xsg.FLAG_IGNORE_CASE), (b"She[r ]lock", xsg.FLAG_REGEX), (b"lock(ed|s)?", xsg.FLAG_REGEX)):
        if flags & xsg.FLAG_REGEX:
            want, _ = oracle_regex_all_modes(oracle, blocks, pat, False)
        else:
            want = oracle_all_modes(oracle, blocks, pat, ignore_case=bool(flags & xsg.FLAG_IGNORE_CASE))
        for key in ("count_matches", "count_lines", "match_byte_offsets", "line_byte_offsets", "line_indices", "lines",
                    "count_async", "count_begin"):
            ctx = xsg.Context(0)  # a fresh context and binding: nothing allocated by an earlier tag
            sh = xsg.Shard(ctx, t.data_ptr(), t.numel(), chunks)
            ctx.set_pattern(pat, flags)
            if key == "count_async":  # the stream-ordered entry point as the first call
                import torch
                d = torch.zeros(xsg.NUM_COUNTERS + 1, dtype=torch.int64, device="cuda:0")
                st = torch.cuda.Stream()
                sh.count_async_status(xsg.COUNT_MATCHES, st.cuda_stream, d.data_ptr(), d.data_ptr() + 8 * xsg.NUM_COUNTERS)
                st.synchronize()
                h = d.cpu().numpy().astype(np.uint64)
                if int(h[xsg.NUM_COUNTERS]) == 0:
                    assert int(h[xsg.CTR_MATCHES]) == want["count_matches"], (pat, key)
                sh.close()
                ctx.close()
                continue
            if key == "count_begin":
                sh.count_begin(xsg.COUNT_LINES)
                assert int(sh.count_end()[xsg.CTR_LINES]) == want["count_lines"], (pat, key)
                sh.close()
                ctx.close()
                continue
            if key == "count_matches":
                got = int(sh.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES])
            elif key == "count_lines":
                got = int(sh.count(xsg.COUNT_LINES)[xsg.CTR_LINES])
            elif key == "lines":
                got = sh.search_lines()[0]
            else:
                mode = {"match_byte_offsets": xsg.MATCH_BYTE_OFFSETS, "line_byte_offsets": xsg.LINE_BYTE_OFFSETS,
                        "line_indices": xsg.LINE_INDICES}[key]
                got = sh.search_u64(mode).tolist()
            assert got == want[key], (pat, key)
            sh.close()
            ctx.close()
