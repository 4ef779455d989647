"""The oracle (oracle/xs_oracle.c) against the reference's own known answers and
against vectors produced by the reference build (oracle/_ref) -- CPU only."""
import numpy as np
import pytest

import golden_util as G


def _b(s):
    return s.encode("latin-1")


def test_reference_simd_search_known_answers(oracle):
    # test/src/string_search/simd_searchTest.cpp:36-99 (27 assertions)
    ka = G.load("ref_simd_search_known_answers.json")
    text = _b(ka["text"])
    assert len(text) == ka["text_len"] == 1240
    n = 0
    for c, want in ka["strchr"]:
        assert oracle.strchr(text, _b(c)) == want; n += 1
    for p, want in ka["strstr"]:
        assert oracle.strstr(text, _b(p)) == want; n += 1
    for p, shift, want in ka["findNext"]:
        assert oracle.find_next(_b(p), text, shift) == want; n += 1
    for shift, want in ka["findNextNewLine"]:
        assert oracle.find_next_newline(text, shift) == want; n += 1
    for p, want in ka["countMatchingLines"]:
        assert oracle.count_matching_lines(_b(p), text) == want; n += 1
    for p, want in ka["countMatches"]:
        assert oracle.count_matches(_b(p), text) == want; n += 1
    assert n == 32  # 4 + 5 + 7 + 4 + 6 + 6


def test_reference_search_wrappers_known_answers(oracle):
    # test/src/string_search/search_wrappersTest.cpp:26,39,52,64-69
    ka = G.load("ref_search_wrappers_known_answers.json")
    text, pat = _b(ka["text"]), _b(ka["pattern"])
    assert oracle.byte_offsets_match(text, pat).tolist() == ka["byte_offsets_match"]
    assert oracle.byte_offsets_line(text, pat).tolist() == ka["byte_offsets_line"]
    assert oracle.count(text, pat) == ka["count"]
    assert oracle.lines(text, pat) == [_b(s) for s in ka["line"]]
    # the stale expectation differs only by the trailing newline
    assert [l + b"\n" for l in oracle.lines(text, pat)] == [_b(s) for s in ka["line_as_asserted_stale"]]
    assert oracle.line_indices(text, pat).tolist() == [0, 2, 3, 8]


def test_generated_vectors(oracle):
    n = 0
    for name, data, e in G.generated_cases():
        p = _b(e["pattern"])
        ctx = f"{name} pat={e['pattern']!r}"
        assert oracle.count_matches(p, data) == e["countMatches"], ctx
        assert oracle.count_matching_lines(p, data) == e["countMatchingLines"], ctx
        assert oracle.count(data, p, True) == e["count_skip"], ctx
        assert oracle.count(data, p, False) == e["count_noskip"], ctx
        assert oracle.byte_offsets_match(data, p).tolist() == e["byte_offsets_match"], ctx
        if "byte_offsets_line" in e:
            assert oracle.byte_offsets_line(data, p).tolist() == e["byte_offsets_line"], ctx
            beg, ln = oracle.lines_spans(data, p)
            assert beg.tolist() == e["lines_begin"] and ln.tolist() == e["lines_len"], ctx
            assert oracle.line_indices(data, p).tolist() == e["line_indices"], ctx
        for shift, want in e["findNext"]:
            assert oracle.find_next(p, data, shift) == want, ctx
        n += 1
    assert n > 100


def test_lossy_tail_is_reproduced(oracle):
    # measured behaviour of the reference
    assert oracle.find_next(b"aab", b"aaab") == -1
    assert oracle.find_next(b"Sherlock", b"x" * 100 + b"SheSherlock") == -1
    assert oracle.find_next(b"Sherlock", b"x" * 100 + b"SheSherlock" + b"y" * 64) == 103
    oracle.set_exact(True)
    try:
        assert oracle.find_next(b"aab", b"aaab") == 1
        assert oracle.find_next(b"Sherlock", b"x" * 100 + b"SheSherlock") == 103
    finally:
        oracle.set_exact(False)


def test_restatement_equals_reference_build_on_random_inputs(oracle, reference):
    """Live cross-check against oracle/_ref (only where the reference build exists)."""
    rng = np.random.default_rng(12345)
    pats = [b"a", b"aa", b"ab", b"aba", b"abab", b"bab", b"aab", b"abc", b"ab\n"]
    alph = np.frombuffer(b"ab\n", dtype=np.uint8)
    for it in range(4000):
        n = int(rng.integers(0, 300))
        data = alph[rng.integers(0, 3 if it % 3 else 2, size=n)].copy()
        p = pats[int(rng.integers(0, len(pats)))]
        shift = int(rng.integers(0, n + 2))
        assert oracle.find_next(p, data, shift) == reference.find_next(p, data, shift)
        assert oracle.find_next_newline(data, shift) == reference.find_next_newline(data, shift)
        assert oracle.count_matches(p, data) == reference.count_matches(p, data)
        assert oracle.count_matching_lines(p, data) == reference.count_matching_lines(p, data)
        assert oracle.scalar_strstr(data, p) == reference.scalar_strstr(data, p)


def test_wrappers_over_reference_primitives_equal_restated(oracle, reference):
    rng = np.random.default_rng(777)
    alph = np.frombuffer(b"ab \n", dtype=np.uint8)
    pats = [b"a", b"aa", b"ab", b"aba", b"abab", b"b a", b"aab"]
    for it in range(1500):
        n = int(rng.integers(0, 500))
        data = alph[rng.integers(0, 4, size=n)].copy()
        p = pats[int(rng.integers(0, len(pats)))]
        got = (oracle.byte_offsets_match(data, p).tolist(), oracle.byte_offsets_line(data, p).tolist(),
               oracle.count(data, p, True), oracle.count(data, p, False),
               [x.tolist() for x in oracle.lines_spans(data, p)], oracle.line_indices(data, p).tolist())
        oracle.use_reference_primitives(reference)
        try:
            want = (oracle.byte_offsets_match(data, p).tolist(), oracle.byte_offsets_line(data, p).tolist(),
                    oracle.count(data, p, True), oracle.count(data, p, False),
                    [x.tolist() for x in oracle.lines_spans(data, p)], oracle.line_indices(data, p).tolist())
        finally:
            oracle.use_reference_primitives(None)
        assert got == want


def test_restatement_equals_reference_build_on_text_with_dense_and_bordered_needles(oracle, reference):
    """The inputs on which round 3's device routes act -- needles that are in most lines (`e`, `the`), dense 4..8-byte
    ones (`that`, `Holmes`), bordered ones with and without overlapping occurrences in the data -- pinned on the CPU
    side: the restatement against the reference's own compiled primitives, and the wrappers over both."""
    import corpus
    texts = [corpus.text_block(91, i, n, needle_rate=0.02) for i, n in enumerate((20011, 4096, 333, 65536))]
    glued = np.frombuffer(b"so thathat is that and thathathat too\nelselse stats statstats\n" * 4, dtype=np.uint8)
    texts.append(np.concatenate([texts[1][:1000], glued, texts[1][1000:]]))
    texts.append(texts[0][:-1].copy())  # unterminated
    for data in texts:
        for p in (b"e", b"th", b"the", b"that", b"else", b"stats", b"Holmes", b"Sherlock", b"was w", b"tt", b"e the", b"aXa"):
            assert oracle.count_matches(p, data) == reference.count_matches(p, data), p
            assert oracle.count_matching_lines(p, data) == reference.count_matching_lines(p, data), p
            got = (oracle.byte_offsets_match(data, p).tolist(), oracle.byte_offsets_line(data, p).tolist(),
                   oracle.count(data, p, True), oracle.count(data, p, False), [x.tolist() for x in oracle.lines_spans(data, p)])
            oracle.use_reference_primitives(reference)
            try:
                want = (oracle.byte_offsets_match(data, p).tolist(), oracle.byte_offsets_line(data, p).tolist(),
                        oracle.count(data, p, True), oracle.count(data, p, False), [x.tolist() for x in oracle.lines_spans(data, p)])
            finally:
                oracle.use_reference_primitives(None)
            assert got == want, p


def test_chunk_driver_matches_single_thread(oracle):
    import corpus
    blocks = [corpus.text_block(3, i, 200_000 + 1000 * i, needle_rate=1e-3) for i in range(7)]
    off, ln, cap = corpus.chunk_table([b.size for b in blocks])
    buf = np.zeros(cap, dtype=np.uint8)
    for o, b in zip(off, blocks):
        buf[int(o):int(o) + b.size] = b
    want = [oracle.count(b, b"Sherlock", False) for b in blocks]
    for nt in (1, 4):
        tot, per = oracle.count_chunks_mt(buf, off, ln, b"Sherlock", False, nt)
        assert per.tolist() == want and tot == sum(want)


def test_ignore_case_semantics_equal_the_references_strcasestr(oracle, reference):
    """ignore_case is served as search(toLower(chunk), toLower(pattern)).  The snapshot has no
    caller of its case-insensitive primitive, but the primitive exists: simd::strcasestr
    (src/string_search/simd_search.cpp:220-287, scalar tail :80-104).  On the compiled reference, strcasestr(data, pat)
    returns exactly what strstr(toLower(data), toLower(pat)) returns -- body, lossy tail and block anchoring included
    -- so the convention is the reference's own (patterns of 2+ bytes; strcasestr is undefined for one byte)."""
    import numpy as np
    rng = np.random.default_rng(4711)
    alphabets = [np.frombuffer(b"abAB", dtype=np.uint8), np.frombuffer(b"abcABC \n", dtype=np.uint8),
                 np.frombuffer(b"Sherlock sHERLOCK\n", dtype=np.uint8), np.arange(256, dtype=np.uint8)]
    n = 0
    for it in range(6000):
        al = alphabets[it % len(alphabets)]
        size = int(rng.integers(0, 300))
        data = al[rng.integers(0, len(al), size=size)].copy()
        plen = int(rng.integers(2, 9))
        if size > plen + 2 and rng.random() < 0.7:
            o = int(rng.integers(0, size - plen))
            pat = data[o:o + plen].copy()
            flip = rng.random(plen) < 0.5  # change the case of some letters of the needle
            pat = np.where(flip & (((pat | 0x20) >= 97) & ((pat | 0x20) <= 122)), pat ^ 0x20, pat).astype(np.uint8)
        else:
            pat = al[rng.integers(0, len(al), size=plen)].copy()
        got = reference.strcasestr(data, pat.tobytes())
        want = reference.strstr(oracle.lower(data), oracle.lower(pat).tobytes())
        assert got == want, (data.tobytes(), pat.tobytes())
        assert oracle.strstr(oracle.lower(data), oracle.lower(pat).tobytes()) == want
        n += got >= 0
    assert n > 1500


def test_generated_ignore_case_vectors(oracle):
    """The oracle's ignore_case convention (search on toLower(data), toLower(pattern)) against the vectors the
    reference's own strcasestr produced."""
    n = 0
    for name, data, e in G.generated_icase_cases():
        low = oracle.lower(data)
        p = oracle.lower(e["pattern"].encode("latin-1")).tobytes()
        ctx = f"{name} pat={e['pattern']!r}"
        assert oracle.count(low, p, True) == e["count_skip"], ctx
        assert oracle.count(low, p, False) == e["count_noskip"], ctx
        assert oracle.byte_offsets_match(low, p).tolist() == e["byte_offsets_match"], ctx
        assert oracle.byte_offsets_line(low, p).tolist() == e["byte_offsets_line"], ctx
        beg, ln = oracle.lines_spans(low, p)
        assert beg.tolist() == e["lines_begin"] and ln.tolist() == e["lines_len"], ctx
        assert oracle.line_indices(low, p, 0).tolist() == e["line_indices"], ctx
        n += 1
    assert n >= 40


def _py_walks(ref_or_oracle, data: bytes, pat: bytes):
    """search_wrappers.h:29-52, 111-123, 149-154, 163-207 typed out in Python on top of findNext / findNextNewLine
    (the reference's compiled primitives when `ref_or_oracle` is the Reference): what the reference's walks return for
    ANY pattern, one that contains '\\n' included -- count(skip_to_nl), byte_offsets_line, line."""
    n = len(data)

    def prev_nl_rel(v):  # :111-123 (uint64 arithmetic: a match whose first byte is '\n' returns 2^64 - 1)
        rel = 0
        while True:
            if data[v - rel] == 10:
                return (rel - 1) & 0xFFFFFFFFFFFFFFFF
            if rel >= v:
                return v
            rel += 1

    count, line_offs, lines = 0, [], []
    shift = 0
    while shift < n:  # :163-185 with skip_to_nl and :29-52 with the line-start functor (:149-154)
        m = ref_or_oracle.find_next(pat, data, shift)
        if m == -1:
            break
        count += 1
        line_offs.append((m - prev_nl_rel(m)) & 0xFFFFFFFFFFFFFFFF)
        shift = m + len(pat)
        nl = ref_or_oracle.find_next_newline(data, shift)
        if nl == -1:
            break
        shift = nl + 1
    shift = 0
    while shift < n:  # :187-207
        m = ref_or_oracle.find_next(pat, data, shift)
        if m == -1:
            break
        begin = (m - prev_nl_rel(m)) & 0xFFFFFFFFFFFFFFFF
        shift = m + len(pat)
        end = ref_or_oracle.find_next_newline(data, shift)
        if end == -1:
            break
        shift = end + 1
        lines.append(data[begin:end])
    return count, line_offs, lines


def test_line_walks_of_patterns_that_contain_a_newline(oracle, reference):
    """The reference's line walks are defined for any std::string (search_wrappers.h:29-50, 163-207): after a match, on to
    the first '\\n' at or behind its END.  With a '\\n' inside the pattern an occurrence reaches into the next line and the
    reported "line" spans several; a pattern that BEGINS with '\\n' gets its line start one byte behind the match start
    (previous_new_line_offset_relative_to_match looks at the match's first byte first and wraps, :111-123).  The oracle's
    C walks must equal the walks typed out above on the reference's own compiled findNext / findNextNewLine."""
    rng = np.random.default_rng(4242)
    pats = [b"a\n", b"\na", b"\n", b"\n\n", b"b\nb", b"ab\nab", b"a\nb\na", b"\nab\n", b"b\n\nb"]
    cases = 0
    for it in range(600):
        n = int(rng.integers(0, 400))
        data = bytes(np.frombuffer(b"ab\n", dtype=np.uint8)[rng.integers(0, 3, size=n)]) + b"a" * int(rng.integers(0, 80)) * int(rng.integers(0, 2))
        if rng.random() < 0.5 and data:
            data = data[:-1] + b"\n"
        arr = np.frombuffer(data, dtype=np.uint8)
        for p in pats:
            want = _py_walks(reference, data, p)
            assert oracle.count(arr, p, True) == want[0], (data, p)
            assert [int(x) for x in oracle.byte_offsets_line(arr, p)] == want[1], (data, p)
            assert oracle.lines(arr, p) == want[2], (data, p)
            # xs::line_indices has no reference implementation: the newlines before the line start
            assert [int(x) for x in oracle.line_indices(arr, p, 7)] == [7 + data[:b].count(b"\n") for b in want[1]], (data, p)
            cases += 1
    assert cases == 600 * len(pats)
