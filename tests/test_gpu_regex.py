"""XSG_FLAG_REGEX on the GPU (class-sequence expressions decided inside k_scan) against the oracle's
restatement of the reference's regex walks (search_wrappers.h:63-103,209-271), all six tags, bit-exact."""
import json
from pathlib import Path

import numpy as np
import pytest

import corpus
import xsg
from gpu_util import GpuSearch, oracle_all_modes, oracle_regex_all_modes
from test_oracle_regex import ACCEPTED, REFUSED, rand_expr, rand_expr2
from test_regex_dfa import MULTILINE, VARIABLE, rand_var_expr
from xs_oracle import UnsupportedRegex

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def gs():
    return GpuSearch()


def check(gs, oracle, blocks, expr, icase=False, ctx="", **kw):
    flags = xsg.FLAG_REGEX | (xsg.FLAG_IGNORE_CASE if icase else 0)
    try:
        want, with_lines = oracle_regex_all_modes(oracle, blocks, expr, icase, **kw)
    except UnsupportedRegex:
        # an ascii_only expression ('.', negated classes) on data with bytes >= 0x80: every entry point refuses
        gs.ctx.set_pattern(expr, flags)
        s = gs.shard
        calls = [lambda: s.count(xsg.COUNT_MATCHES), lambda: s.search_u64(xsg.MATCH_BYTE_OFFSETS)]
        n, sets = xsg.regex_check(expr, flags & xsg.FLAG_IGNORE_CASE)  # n == 0: the automaton route, never matches '\n'
        if not any((int(sets[k][0]) >> 10) & 1 for k in range(n)):  # line modes too, unless the expression can match '\n'
            calls += [lambda: s.count(xsg.COUNT_LINES), lambda: s.search_u64(xsg.LINE_INDICES), lambda: s.search_lines()]
        for call in calls:
            with pytest.raises(xsg.XsgError) as ei:
                call()
            assert ei.value.code == xsg.ENOTSUP and "0x80" in str(ei.value), ctx
        return None
    got = gs.all_modes(expr, flags, lines=with_lines)
    for k in want:
        assert got[k] == want[k], f"{ctx} expr={expr!r} icase={icase}: {k}: got {str(got[k])[:160]} want {str(want[k])[:160]}"
    return want


def test_reference_known_answers(gs, oracle):
    ka = json.loads((GOLD / "ref_search_wrappers_known_answers.json").read_text())
    text = np.frombuffer(ka["text"].encode("latin-1"), dtype=np.uint8)
    gs.bind([text])
    r = ka["regex"]
    got = gs.all_modes(r["pattern"].encode(), xsg.FLAG_REGEX)
    assert got["match_byte_offsets"] == r["byte_offsets_match"]   # search_wrappersTest.cpp:77-83
    assert got["line_byte_offsets"] == r["byte_offsets_line"]     # :90-96
    assert got["count_lines"] == r["count"]                       # :103
    assert [x.decode() for x in got["lines"]] == ka["line"]
    check(gs, oracle, [text], r["pattern"].encode())


def test_the_integration_tests_expression_on_text(gs, oracle):
    """`She[r ]lock` (test/src/xsearchTest.cpp:9) on generated text with both spellings planted, several chunks."""
    rng = np.random.default_rng(31)
    blocks = []
    for i in range(4):
        b = corpus.text_block(900 + i, 0, 3_000_000 + 777 * i, needle_rate=3e-5)
        nl = np.flatnonzero(b == 10)
        for pos in rng.choice(nl[:-2], size=40, replace=False):
            w = [b"She lock", b"SHE LOCK", b"sherlock", b"She\tlock", b"Sherlocc"][int(rng.integers(0, 5))]
            b[pos + 1:pos + 1 + len(w)] = np.frombuffer(w, dtype=np.uint8)
        blocks.append(b)
    gs.bind(blocks)
    for icase in (False, True):
        want = check(gs, oracle, blocks, b"She[r ]lock", icase, "text")
        lit = oracle_all_modes(oracle, blocks, b"Sherlock", True, ignore_case=icase)
        assert want["count_matches"] > lit["count_matches"] > 50   # the class adds the `She lock` spellings
    assert check(gs, oracle, blocks, b"She[r ]lock", True)["count_matches"] > check(gs, oracle, blocks, b"She[r ]lock")["count_matches"]


def test_accepted_expressions_and_literal_only_ones(gs, oracle):
    rng = np.random.default_rng(8)
    alphabet = np.frombuffer(b"abcABC xyz019_-.]\n\tgreyant[|m\xc3\xa9", dtype=np.uint8)
    blocks = [alphabet[rng.integers(0, len(alphabet), size=n)].copy() for n in (70_000, 16384, 16385, 33, 5, 0, 40_001)]
    for b in blocks:
        if b.size:
            b[-1] = 10
    gs.bind(blocks)
    for expr in ACCEPTED + [b"a\\.b", b"\\]\\.", b"A", b"\\n", b"[ab]\\n", b"\\s\\s"]:
        for icase in (False, True):
            check(gs, oracle, blocks, expr, icase, "accepted")
    # an expression of singletons is an ordinary literal without the reference's scalar-tail quirk
    gs.ctx.set_pattern(b"ab\\.", xsg.FLAG_REGEX)
    a = gs.shard.search_u64(xsg.MATCH_BYTE_OFFSETS).tolist()
    gs.ctx.set_pattern(b"ab.", xsg.FLAG_EXACT_TAIL)
    assert a == gs.shard.search_u64(xsg.MATCH_BYTE_OFFSETS).tolist()


def test_alternation_dot_and_negation_on_ascii_text(gs, oracle):
    """round 2 of the regex row: equal-length alternations, '.', negated classes -- on ASCII text, where the
    byte-per-position reading IS RE2's"""
    blocks = [corpus.text_block(4242, i, 1_500_000 + 333 * i, needle_rate=1e-4) for i in range(3)]
    gs.bind(blocks)
    for expr in (b"Sherlock|She lock", b"Holmes|Watson", b"(Sher|padd)lock", b"She.lock", b"S.{6}k", b"[^a-z]he ",
                 b"(the|and) (she|was|had)", b"street|locked|Watson", b"\\D\\d\\D", b"lock[^e]", b"(ab|cd){2}",
                 b"S(her|HER)lock", b"[[:upper:]][[:lower:]]{5} "):
        for icase in (False, True):
            want = check(gs, oracle, blocks, expr, icase, "ascii")
            assert want is not None
    # an alternation whose alternatives differ in one position is the class sequence it merges into
    a = check(gs, oracle, blocks, b"She(r| )lock")
    b = check(gs, oracle, blocks, b"She[r ]lock")
    assert a == b and a["count_matches"] > 0


def test_ascii_only_expressions_refuse_non_ascii_data(gs, oracle):
    """'.' and negated classes match whole code points in RE2: on data with a byte >= 0x80 the search is refused
    (XSG_ENOTSUP), never decided byte by byte; the same data is fine for expressions without them, and the same
    expressions are fine again on the next ASCII shard (the refusal leaves no state behind)."""
    import torch
    text = corpus.text_block(99, 0, 400_000)
    dirty = text.copy()
    dirty[123_457:123_459] = (0xc3, 0xa9)  # one 'é' somewhere in the middle
    gs.bind([text, dirty])
    assert check(gs, oracle, [text, dirty], b"She.lock") is None
    assert check(gs, oracle, [text, dirty], b"[^x]he ", True) is None
    assert check(gs, oracle, [text, dirty], b"She[r ]lock") is not None       # byte-exact on any data
    assert check(gs, oracle, [text, dirty], b"caf\xc3\xa9|She l") is not None
    # the asynchronous count has no return code to refuse with: it poisons all four counters
    gs.ctx.set_pattern(b"t.e", xsg.FLAG_REGEX)
    c = torch.zeros(xsg.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    gs.shard.count_async(xsg.COUNT_LINES, 0, c.data_ptr())  # (`t.e` may overlap itself: its match count is xsg_count's)
    torch.cuda.synchronize()
    assert all(int(x) == -1 for x in c.cpu())
    gs.bind([text])
    assert check(gs, oracle, [text], b"t.e") is not None


@pytest.mark.parametrize("expr", REFUSED[:12])
def test_refused_expressions_fail_loudly(gs, expr):
    with pytest.raises(xsg.XsgError) as ei:
        gs.ctx.set_pattern(expr, xsg.FLAG_REGEX)
    assert ei.value.code in (xsg.ENOTSUP, xsg.EINVAL)


def regex_rounds(seed, oracle, gs, rounds=20):
    """`rounds` random shards x up to 6 random class-sequence expressions; AssertionError on the first difference."""
    rng = np.random.default_rng(5000 + seed)
    sizes = [0, 1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 1023, 1024, 1025, 4096, 4097, 16383, 16384, 16385, 32769]
    alphabets = [np.frombuffer(b"abcABC xyz019_-.]\n\n\t", dtype=np.uint8), np.frombuffer(b"ab\n", dtype=np.uint8),
                 np.frombuffer(b"abc019 \n", dtype=np.uint8)]
    for it in range(rounds):
        alphabet = alphabets[int(rng.integers(0, len(alphabets)))]
        blocks = []
        for _ in range(int(rng.integers(1, 6))):
            n = int(rng.choice(sizes)) if rng.random() < 0.7 else int(rng.integers(0, 80_000))
            b = alphabet[rng.integers(0, len(alphabet), size=n)].copy()
            if n and rng.random() < 0.6:
                b[-1] = 10
            blocks.append(b)
        gs.bind(blocks)
        for k in range(6):
            expr = rand_expr(rng) if k % 2 == 0 else rand_expr2(rng)
            try:
                xsg.regex_check(expr)
            except xsg.XsgError:
                continue
            check(gs, oracle, blocks, expr, bool(rng.integers(0, 2)), f"seed={seed} it={it} sizes={[b.size for b in blocks]}")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_expressions(gs, oracle, seed):
    regex_rounds(seed, oracle, gs)


@pytest.mark.parametrize("seed", [21, 22])
def test_random_expressions_with_the_aligned_dword_trigger(oracle, seed):
    regex_rounds(seed, oracle, GpuSearch(hot=1))


def test_global_offsets_line_bases_and_job_api(gs, oracle, tmp_path):
    blocks = [corpus.text_block(77, i, 200_000) for i in range(3)]
    goffs = [10**9, 5 * 10**9, 2**40]
    bases = [7, 1000, 123456789]
    gs.bind(blocks, goffs, bases)
    check(gs, oracle, blocks, b"[Tt]he [a-z]{3} ", False, "global", global_offsets=goffs, line_bases=bases)
    # through the file pipeline (xsg_job_*): pattern_flags carries XSG_FLAG_REGEX
    data = np.concatenate(blocks)
    path = tmp_path / "t.txt"
    data.tofile(path)
    want, _ = oracle_regex_all_modes(oracle, [data], b"[Tt]he [a-z]{3} ")
    j = xsg.Job(b"[Tt]he [a-z]{3} ", str(path), mode=xsg.MATCH_BYTE_OFFSETS, flags=xsg.FLAG_REGEX, chunk_bytes=65536,
                num_threads=2, num_max_readers=2)
    got = j.result().tolist()
    # chunks are newline-aligned, so per-chunk walks concatenate to the whole-file walk
    assert got == want["match_byte_offsets"]


# ---- the variable-length half: k_rx_scan (csrc/xsg_rx_kernels.hip), one line per lane, two automata -----------------
def test_variable_length_expressions_on_text(gs, oracle):
    """operators, lazy forms and alternatives of different lengths on generated text, several chunks, all six tags"""
    blocks = [corpus.text_block(5151, i, 1_200_000 + 4321 * i, needle_rate=1e-4) for i in range(3)]
    blocks[2] = np.concatenate([blocks[2][:-1], np.frombuffer(b" Sherlock Holmes", dtype=np.uint8)])  # no final newline
    gs.bind(blocks)
    for expr in (b"Sherlock|Holmes|Dr\\. Watson", b"Sher.*mes", b"Sher.*?k", b"coul?d", b"[A-Z][a-z]+ [A-Z][a-z]+",
                 b"lock(ed|s)?", b"\\w+ere", b"(the|The) +\\w{5,}", b"S[a-z]{3,9}k", b"e{2,}", b"[0-9]+", b"a.{0,4}?y",
                 b"(?:st|pad)lock|str+eet"):
        for icase in (False, True):
            want = check(gs, oracle, blocks, expr, icase, "variable")
            assert want is not None
            if expr != b"[0-9]+":  # the generated text has no digits
                assert want["count_matches"] > 0 and want["count_lines"] > 0


def test_variable_length_expressions_on_awkward_shards(gs, oracle):
    """chunks of 0, 1, tile and tile+-1 bytes, lines longer than a tile, lines that straddle tiles, last lines
    without a newline, data of newlines only"""
    rng = np.random.default_rng(77)
    alphabet = np.frombuffer(b"aabbccxyz01 _\n", dtype=np.uint8)
    long_line = alphabet[rng.integers(0, len(alphabet) - 1, size=50_000)].copy()  # no newline in 50 KB
    blocks = [alphabet[rng.integers(0, len(alphabet), size=n)].copy() for n in (0, 1, 16383, 16384, 16385, 40_000)]
    blocks += [long_line, np.full(20_000, 10, dtype=np.uint8), np.concatenate([long_line[:20_000], np.array([10], dtype=np.uint8), long_line[:30_000]])]
    gs.bind(blocks)
    for expr in VARIABLE:
        if b"\xc3" in expr:
            continue
        for icase in (False, True):
            check(gs, oracle, blocks, expr, icase, "awkward")


def test_variable_length_ascii_only_refusal_and_job_api(gs, oracle, tmp_path):
    import torch
    text = corpus.text_block(98, 0, 300_000)
    dirty = text.copy()
    dirty[200_001:200_003] = (0xc3, 0xa9)
    gs.bind([text, dirty])
    assert check(gs, oracle, [text, dirty], b"Sher.*k") is None          # '.' on non-ASCII data: refused
    assert check(gs, oracle, [text, dirty], b"Sherlock|Holmes!*") is not None  # explicit bytes only: served
    gs.ctx.set_pattern(b"Sher.*k", xsg.FLAG_REGEX)
    c = torch.zeros(xsg.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    gs.shard.count_async(xsg.COUNT_MATCHES, 0, c.data_ptr())
    torch.cuda.synchronize()
    assert all(int(x) == -1 for x in c.cpu())
    gs.bind([text])
    want = check(gs, oracle, [text], b"Sher.*k")
    # asynchronous counts of both kinds, on the clean shard
    gs.ctx.set_pattern(b"Sher.*k", xsg.FLAG_REGEX)
    for mode, key, ctr in ((xsg.COUNT_MATCHES, "count_matches", xsg.CTR_MATCHES), (xsg.COUNT_LINES, "count_lines", xsg.CTR_LINES)):
        gs.shard.count_async(mode | xsg.WITH_NEWLINES, 0, c.data_ptr())
        torch.cuda.synchronize()
        got = c.cpu().tolist()
        assert got[ctr] == want[key] and got[xsg.CTR_NEWLINES] == want["newlines"]
    # through the file pipeline
    path = tmp_path / "t.txt"
    text.tofile(path)
    j = xsg.Job(b"[Tt]he +[a-z]+ly ", str(path), mode=xsg.MATCH_BYTE_OFFSETS, flags=xsg.FLAG_REGEX, chunk_bytes=65536,
                num_threads=2, num_max_readers=2)
    want2, _ = oracle_regex_all_modes(oracle, [text], b"[Tt]he +[a-z]+ly ")
    assert j.result().tolist() == want2["match_byte_offsets"]


def oracle_is_quick(oracle, blocks, expr, icase, seconds=3):
    """CPython's `re` -- the oracle's stand-in for RE2 -- backtracks; a random expression with ambiguous alternatives
    under a loop (`([ab]|a)+c`) can take exponential time where the product's automata are linear.  Such an
    expression is no use as a test case: found out here, under an alarm, before the GPU is asked."""
    import signal
    from xs_oracle import RegexProgram, UnsupportedRegex, compile_class_sequence

    class TooSlow(Exception):
        pass

    def on_alarm(*_):
        raise TooSlow()

    try:
        compile_class_sequence(expr, icase)
        return True  # a class sequence: the C oracle, no backtracking
    except UnsupportedRegex:
        pass
    prog = RegexProgram(expr, icase)
    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(seconds)
    try:
        for b in blocks:
            oracle.rx_byte_offsets(b, prog, False)
        return True
    except TooSlow:
        return False
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)


def rx_rounds(seed, oracle, gs, rounds=12):
    rng = np.random.default_rng(9000 + seed)
    sizes = [0, 1, 63, 64, 65, 1023, 1024, 1025, 16383, 16384, 16385, 32769]
    alphabets = [np.frombuffer(b"aabbccxyz01 _\n\n", dtype=np.uint8), np.frombuffer(b"ab\n", dtype=np.uint8),
                 np.frombuffer(b"abcxyz01 _ abc\n", dtype=np.uint8)]
    done = 0
    for it in range(rounds):
        alphabet = alphabets[int(rng.integers(0, len(alphabets)))]
        blocks = []
        for _ in range(int(rng.integers(1, 6))):
            n = int(rng.choice(sizes)) if rng.random() < 0.6 else int(rng.integers(0, 60_000))
            b = alphabet[rng.integers(0, len(alphabet), size=n)].copy()
            if n and rng.random() < 0.6:
                b[-1] = 10
            blocks.append(b)
        gs.bind(blocks)
        for k in range(6):
            expr = rand_var_expr(rng)
            icase = bool(rng.integers(0, 3) == 0)
            try:
                xsg.regex_check(expr, xsg.FLAG_IGNORE_CASE if icase else 0)
            except xsg.XsgError:
                continue
            if not oracle_is_quick(oracle, blocks, expr, icase):
                continue
            check(gs, oracle, blocks, expr, icase, f"seed={seed} it={it} sizes={[b.size for b in blocks]}")
            done += 1
    return done


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_variable_length_expressions(gs, oracle, seed):
    assert rx_rounds(seed, oracle, gs) > 20


def test_expressions_that_can_match_a_newline(gs, oracle):
    """a set of the expression accepts '\\n' (\\s+, [^a]+): matches may span lines, the chunk is walked by one lane
    (k_rx_chunk); the match tags only -- the line tags refuse, as for a literal that contains a newline"""
    rng = np.random.default_rng(4)
    alphabet = np.frombuffer(b"aabbxy \n\n,Z", dtype=np.uint8)
    blocks = [alphabet[rng.integers(0, len(alphabet), size=n)].copy() for n in (0, 1, 5000, 16384, 16385, 40_000)]
    blocks.append(corpus.text_block(12, 0, 300_000))
    gs.bind(blocks)
    for expr in MULTILINE + [b"Sherlock\\s+Holmes", b"[a-z]+\\s+[A-Z][a-z]+"]:
        for icase in (False, True):
            want = check(gs, oracle, blocks, expr, icase, "multiline")
            assert want is not None and "count_lines" not in want
    gs.ctx.set_pattern(b"x\\s*y", xsg.FLAG_REGEX)
    for call in (lambda: gs.shard.count(xsg.COUNT_LINES), lambda: gs.shard.search_u64(xsg.LINE_INDICES), lambda: gs.shard.search_lines()):
        with pytest.raises(xsg.XsgError) as ei:
            call()
        assert ei.value.code == xsg.ENOTSUP


def test_both_routes_of_the_automaton_family_agree(gs, oracle, monkeypatch, tmp_path):
    """an expression with a selective start is searched two ways: candidates by the scan kernel's class-sequence
    matcher + the anchored automaton at candidates (xsg_count / xsg_search / the jobs), or every line walked by
    k_rx_scan (xsg_count_async; XSG_RX_PRE=0 forces it everywhere).  Same oracle, same answers."""
    blocks = [corpus.text_block(777, i, 900_000 + 1111 * i, needle_rate=2e-4) for i in range(3)]
    blocks.append(np.frombuffer(b"Sherlock Holmes locked the lock\nSher", dtype=np.uint8).copy())  # ends inside a prefix
    gs.bind(blocks)
    for expr in (b"Sher.*mes", b"Sherlock|Holmes", b"lock(ed|s)?", b"(the|The) +\\w{5,}", b"[Tt]he +[a-z]+ly ",
                 b"She\\s+lock", b"(?:st|pad)lock|str+eet", b"Sher.*?k"):
        assert xsg.regex_prefix(expr)[0] >= 3
        for icase in (False, True):
            res = []
            for pre in ("1", "0"):
                monkeypatch.setenv("XSG_RX_PRE", pre)
                res.append(check(gs, oracle, blocks, expr, icase, f"pre={pre}"))
            assert res[0] == res[1] and res[0] is not None
    # the file pipeline on the prefilter route (xsg_count_begin / _end run it synchronously; list tags through xsg_search)
    data = np.concatenate(blocks[:3])
    path = tmp_path / "routes.txt"
    data.tofile(path)
    want, _ = oracle_regex_all_modes(oracle, [data], b"Sher.*mes")
    for pre in ("1", "0"):
        monkeypatch.setenv("XSG_RX_PRE", pre)
        for mode, key in ((xsg.COUNT_MATCHES, "count_matches"), (xsg.COUNT_LINES, "count_lines"),
                          (xsg.LINE_BYTE_OFFSETS, "line_byte_offsets")):
            j = xsg.Job(b"Sher.*mes", str(path), mode=mode, flags=xsg.FLAG_REGEX, chunk_bytes=1 << 20, num_threads=2,
                        num_max_readers=2)
            r = j.result()
            assert (r if isinstance(r, int) else r.tolist()) == want[key], (pre, key)
            j.close()
    monkeypatch.delenv("XSG_RX_PRE")


def test_prefilter_gives_way_when_candidates_run_far(gs, oracle, monkeypatch):
    """`aaa+` on a long run of `a`: every position of the run is a candidate and would scan to its end -- the
    verification pass has a budget per candidate, and when one outruns it the search is redone by walking the text
    once (k_rx_scan).  Same answers either way."""
    run = np.full(300_000, ord("a"), dtype=np.uint8)
    blocks = [np.concatenate([run, np.frombuffer(b"\nxaaay aaaa b\n", dtype=np.uint8), run[:5000], np.frombuffer(b"\n", dtype=np.uint8)]),
              corpus.text_block(3, 0, 100_000)]
    gs.bind(blocks)
    monkeypatch.setenv("XSG_RX_PRE", "1")  # by default shards this small are walked by k_rx_scan in the first place
    for expr in (b"aaa+", b"aaa+b?", b"xaa+y"):
        assert xsg.regex_prefix(expr)[0] >= 3
        want = check(gs, oracle, blocks, expr, False, "budget")
        assert want is not None and want["count_matches"] >= 1


def test_factor_prefilter_of_expressions_without_a_selective_start(gs, oracle, monkeypatch):
    """`\\w+ere`: no definite start, but every match contains `\\were` -- the synchronous entry points mark the tiles in
    which a line with an occurrence starts (once per binding and pattern) and k_rx_scan leaves every other tile at
    once.  XSG_RX_FAC=1 forces the mask on these small shards, =0 walks every tile; same answers, also from the
    stream-ordered count that follows and reuses the mask."""
    import torch
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"aabbccxyz01 _\n", dtype=np.uint8)
    long_line = alphabet[rng.integers(0, len(alphabet) - 1, size=50_000)].copy()  # no newline in 50 KB: crosses three tiles
    sets = {
        "text": ([corpus.text_block(606, i, 700_000 + 999 * i, needle_rate=2e-4) for i in range(3)],
                 (b"\\w+ere", b"[a-z]*ould", b"\\w+ock(ed|s)?", b"[a-z]+ +Holmes", b"\\w*zzzq\\w*")),
        "awkward": ([alphabet[rng.integers(0, len(alphabet), size=n)].copy() for n in (0, 1, 16383, 16384, 16385, 40_000)] +
                    [long_line, np.concatenate([long_line[:20_000], np.array([10], dtype=np.uint8), long_line[:30_000]])],
                    (b"[a-c]+xyz", b"\\w+_01", b"[ab]*cc[xy]+", b"\\S*01_\\S*")),
    }
    c = torch.zeros(xsg.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    for name, (blocks, exprs) in sets.items():
        gs.bind(blocks)
        for expr in exprs:
            assert xsg.regex_prefix(expr)[0] == 0 and xsg.regex_factor(expr)[0] >= 3, expr
            res = []
            for fac in ("1", "0"):
                monkeypatch.setenv("XSG_RX_FAC", fac)
                res.append(check(gs, oracle, blocks, expr, False, f"{name} fac={fac}"))
                gs.shard.count_async(xsg.COUNT_MATCHES, 0, c.data_ptr())  # stream-ordered, behind the mask of the calls above
                torch.cuda.synchronize()
                assert int(c[xsg.CTR_MATCHES]) == res[-1]["count_matches"], (name, expr, fac)
            assert res[0] == res[1] and res[0] is not None
    monkeypatch.delenv("XSG_RX_FAC")


def test_rewritten_bytes_need_a_rebind_or_invalidate(gs, oracle, monkeypatch):
    """include/xsg.h: the bytes of a binding are immutable; a caller that refills the buffer in place re-binds (or calls
    xsg_shard_invalidate) and every per-binding result derived from the old bytes -- here the factor prefilter's tile
    marks, which make k_rx_scan skip tiles -- is rebuilt.  (Without that call the old marks would hide the new matches:
    that is the documented contract, not tested as behaviour.)"""
    import torch
    monkeypatch.setenv("XSG_RX_FAC", "1")
    a = corpus.text_block(71, 0, 400_000, needle_rate=0.0)
    b = a.copy()
    b[300_000:300_009] = np.frombuffer(b" somewere", dtype=np.uint8)  # a match of \w+ere in a tile that had none
    b[100_000:100_008] = np.frombuffer(b" nowhere", dtype=np.uint8)
    expr = b"\\w+ere"
    gs.bind([a])
    gs.ctx.set_pattern(expr, xsg.FLAG_REGEX)
    from xs_oracle import RegexProgram
    prog = RegexProgram(expr)
    n_a = int(gs.shard.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES])
    assert n_a == oracle.rx_count(a, prog, False)
    assert "factor prefilter" in gs.shard.scan_kernel_name(xsg.COUNT_MATCHES)
    want_b = oracle.rx_count(b, prog, False)
    assert want_b == n_a + 2
    c = torch.zeros(xsg.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    for how in ("invalidate", "rebind"):
        gs.keep[:a.size].copy_(torch.from_numpy(a))          # back to the old bytes, marks rebuilt for them
        torch.cuda.synchronize()
        gs.shard.invalidate()
        assert int(gs.shard.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES]) == n_a
        gs.keep[:b.size].copy_(torch.from_numpy(b))          # the caller rewrites the buffer in place ...
        torch.cuda.synchronize()
        if how == "invalidate":                               # ... and says so
            gs.shard.invalidate()
        else:
            chunks = xsg.make_chunks([0], [b.size])
            gs.shard.rebind(gs.keep.data_ptr(), gs.keep.numel(), chunks)
        assert int(gs.shard.count(xsg.COUNT_MATCHES)[xsg.CTR_MATCHES]) == want_b, how
        gs.shard.count_async(xsg.COUNT_MATCHES, 0, c.data_ptr())
        torch.cuda.synchronize()
        assert int(c[xsg.CTR_MATCHES]) == want_b, how
        assert gs.shard.search_u64(xsg.MATCH_BYTE_OFFSETS).tolist() == oracle.rx_byte_offsets(b, prog, False).tolist()


def test_tiles_without_a_trigger_byte(gs, oracle):
    """k_rx_scan leaves a tile that holds none of the expression's (few) trigger bytes without staging it; the one
    line that starts in such a tile and runs on behind it is followed from the tile's end.  Text without capital
    S / H, needles planted so that matches begin just before, at and just behind tile borders (16 KiB), in lines that
    cross one or several tiles, at the very end of a chunk without a final newline -- all tags, both count routes."""
    import torch
    rng = np.random.default_rng(77)
    words = [w for w in corpus.LEXICON_NOSH]
    def text(n, nl_every):
        out = bytearray()
        while len(out) < n:
            out += words[int(rng.integers(0, len(words)))]
            out += b"\n" if int(rng.integers(0, nl_every)) == 0 else b" "
        return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()
    T = 16384
    b0 = text(5 * T + 1000, 6)
    for pos, w in ((T - 3, b"Sherlock"), (2 * T, b"Holmes"), (3 * T + 1, b"Sherlock Holmes"), (4 * T - 8, b"Sherlock"), (4 * T + 600, b"Holmes")):
        b0[pos:pos + len(w)] = np.frombuffer(w, dtype=np.uint8)
    b1 = text(4 * T, 4000)                         # lines of tens of KB: a line start is followed over several tiles
    b1[100] = 10
    for pos, w in ((T + 5000, b"Sherlock"), (3 * T - 2, b"Holmes"), (3 * T + 9000, b"Sher and then mes")):
        b1[pos:pos + len(w)] = np.frombuffer(w, dtype=np.uint8)
    b2 = text(2 * T + 77, 6)
    b2[-8:] = np.frombuffer(b"Sherlock", dtype=np.uint8)  # the chunk ends in a match, no final newline
    b3 = text(3 * T, 6)                            # no needle at all
    blocks = [b0, b1, b2, b3]
    gs.bind(blocks)
    c = torch.zeros(xsg.NUM_COUNTERS, dtype=torch.int64, device="cuda:0")
    for expr, icase in ((b"Sherlock|Holmes", False), (b"Sher.*mes", False), (b"Holmes?", False), (b"sherlock|holmes", True), (b"Sherlocks?( Holmes)?", False)):
        assert xsg.regex_dfa(expr, xsg.FLAG_IGNORE_CASE if icase else 0)[0].ncls > 0
        res = check(gs, oracle, blocks, expr, icase, "no-trigger tiles")
        assert res["count_matches"] >= 3
        for mode, key in ((xsg.COUNT_MATCHES, "count_matches"), (xsg.COUNT_LINES, "count_lines")):
            gs.shard.count_async(mode | xsg.WITH_NEWLINES, 0, c.data_ptr())  # k_rx_scan itself, newline counts included
            torch.cuda.synchronize()
            assert int(c[xsg.CTR_MATCHES if mode == xsg.COUNT_MATCHES else xsg.CTR_LINES]) == res[key], (expr, key)
            assert int(c[xsg.CTR_NEWLINES]) == res["newlines"], expr
