"""The regex row, class-sequence family: the oracle's restatement of the reference's
regex walks (include/xsearch/string_search/search_wrappers.h:63-103,209-271) against the reference's own
known answers, and both expression parsers -- the oracle's (oracle/xs_oracle.py) and the product's
(x-search_amd/csrc/xsg_classseq.cpp through the C ABI, no GPU needed) -- against each other and against
CPython's `re` as an independent reading of the syntax."""
import json
import re
from pathlib import Path

import numpy as np
import pytest

import xs_oracle
import xsg
from xs_oracle import UnsupportedRegex, compile_class_sequence

GOLD = Path(__file__).parent / "golden"

REFUSED = [b"a*", b"^ab", b"ab$", b"\\bab", b"(?i)ab", b"(?P<n>ab)", b"(ab)*",
           b"(ab)?", b"ab\\", b"[ab", b"(ab", b"ab)", b"\\pLab", b"\\Qab\\E", b"\\1", b"[\xc3\xa9]", b"\xff", b"a{0}", b"{2}",
           b"\\x{100}", b"\\xzz", b"()", b"a||b", b"|a", b"a{2}{3}", b"[[:nope:]]"]
# not class sequences (the oracle's class-sequence reader refuses them), served by the product's automaton route
# (tests/test_regex_dfa.py): variable length, or more positions than the scan kernel's 32
VARIABLE = [b"ab+", b"colou?r", b"x{2,3}", b"x{2,}", b"ab|c", b"(a|bc)d", b"(ab|c)(d|ef)", b"a" * 33]
ACCEPTED = [b"She[r ]lock", b"(a[n|m]t)", b"[0-9]{4}-\\d\\d", b"a\\.b", b"\\w{3} \\w", b"[]a]x", b"[a\\]]x", b"[a-]x", b"[-a]x",
            b"gr[ae]y", b"((a)[bc])d", b"\\x41\\x{42}[\\x43-\\x45]", b"caf\xc3\xa9 [ab]", b"a]b}", b"\\t[ \\t]x", b"[\\d_]x",
            b"\\[a\\]", b"a{3}b{1}", b"[a-c]{32}",
            # round 2: alternation of equal-length sequences, groups with {n}, (?: ), posix classes, and -- ASCII
            # data only -- '.', negated classes, \D \W \S
            b"Sherlock|She lock", b"foo|bar|baz", b"(ab|cd)e", b"x(ab|cd){2}y", b"(?:ab)c", b"(ab){2}", b"\xc3\xa9{2}",
            b"a|b", b"[[:alpha:]]b", b"[[:digit:][:upper:]_]x", b"a.b", b"[^a]b", b"\\Dab", b"\\Wab", b"\\Sab",
            b"[^[:space:]]x", b"(gr[ae]y|blue) ", b"(a|b)(c|d)(e|f)(g|h)", b"[^\\n]x", b"a(.|x)c"]


def product_expr(expr, icase=False):
    """-> (positions, ascii_only, sets[alt][pos][8])"""
    n, na, ao, sets = xsg.regex_info(expr, xsg.FLAG_IGNORE_CASE if icase else 0)
    return n, ao, sets


def oracle_expr(expr, icase=False):
    cs = compile_class_sequence(expr, icase)
    sets = np.array([[[cs.sets[a * cs.plen + k][q] for q in range(8)] for k in range(cs.plen)] for a in range(cs.nalt)],
                    dtype=np.uint32)
    return cs.plen, cs.ascii_only, sets


def accepts(sets, s: bytes) -> bool:
    return any(all((int(alt[k][b >> 5]) >> (b & 31)) & 1 for k, b in enumerate(s)) for alt in sets)


def assert_same_language(expr, icase, rng):
    """The product merges alternatives (She(r| )lock -> She[r ]lock), the oracle keeps the cross product: the two
    are compared by what they accept -- every string drawn from either side's alternatives, mutations of those,
    and noise."""
    no, ao, so = oracle_expr(expr, icase)
    npr, ap, sp = product_expr(expr, icase)
    assert no == npr, expr
    assert ao == ap, expr
    samples = []
    for sets in (so, sp):
        for alt in sets:
            for _ in range(6):
                s = bytearray()
                for k in range(no):
                    members = [b for b in range(256) if (int(alt[k][b >> 5]) >> (b & 31)) & 1]
                    s.append(members[int(rng.integers(0, len(members)))])
                samples.append(bytes(s))
    for s in list(samples):
        for _ in range(3):
            t = bytearray(s)
            t[int(rng.integers(0, len(t)))] = int(rng.integers(0, 256))
            samples.append(bytes(t))
    samples += [bytes(rng.integers(0, 256, size=no, dtype=np.uint8)) for _ in range(20)]
    for s in samples:
        assert accepts(so, s) == accepts(sp, s), (expr, icase, s)


def test_reference_known_answers_for_the_regex_wrappers(oracle):
    ka = json.loads((GOLD / "ref_search_wrappers_known_answers.json").read_text())
    text = ka["text"].encode("latin-1")
    r = ka["regex"]
    cs = compile_class_sequence(r["pattern"].encode())
    assert oracle.regex_byte_offsets_match(text, cs).tolist() == r["byte_offsets_match"]   # search_wrappersTest.cpp:77-83
    assert oracle.regex_byte_offsets_line(text, cs).tolist() == r["byte_offsets_line"]     # :90-96
    assert oracle.regex_count(text, cs) == r["count"]                                      # :103
    # the same text and the literal `ant`: the regex answers equal the literal ones in the reference's tests
    assert r["byte_offsets_match"] == ka["byte_offsets_match"] and r["byte_offsets_line"] == ka["byte_offsets_line"]
    beg, ln = oracle.regex_lines_spans(text, cs)
    assert [text[int(b):int(b + l)].decode() for b, l in zip(beg, ln)] == ka["line"]


@pytest.mark.parametrize("expr", REFUSED)
def test_both_parsers_refuse(expr):
    with pytest.raises(UnsupportedRegex):
        compile_class_sequence(expr)
    with pytest.raises(xsg.XsgError) as ei:
        xsg.regex_check(expr)
    assert ei.value.code in (xsg.ENOTSUP, xsg.EINVAL)
    assert "not supported" in str(ei.value)


@pytest.mark.parametrize("expr", VARIABLE)
def test_not_a_class_sequence_but_served(expr):
    with pytest.raises(UnsupportedRegex):
        compile_class_sequence(expr)
    assert xsg.regex_check(expr)[0] == 0  # 0 positions: "variable length, the automaton route"
    xs_oracle.RegexProgram(expr)


@pytest.mark.parametrize("expr", ACCEPTED)
def test_both_parsers_agree(expr):
    rng = np.random.default_rng(len(expr) * 7919 + expr[0])
    for icase in (False, True):
        assert_same_language(expr, icase, rng)


def test_product_limits_on_alternatives():
    """more than 8 alternatives that cannot be merged, or more than 64 sets: too many for the scan kernel's
    position-wise matcher (the oracle's class-sequence reader, which keeps up to 256 sets, takes them) -- the product
    serves them through the automaton route instead (0 positions reported)"""
    for expr in (b"aa|bb|cc|dd|ee|ff|gg|hh|ii", b"|".join(bytes([97 + i]) * 9 for i in range(8))):
        compile_class_sequence(expr)
        assert xsg.regex_check(expr)[0] == 0
        xsg.regex_dfa(expr)
    # eight that merge down are fine: (a|b)(c|d)(e|f) is one sequence of three classes
    assert xsg.regex_info(b"(a|b)(c|d)(e|f)")[1] == 1


def rand_expr(rng):
    """A random class-sequence expression in the syntax both RE2 and CPython read the same way."""
    letters = b"abcABC xyz019_-.]"
    out = []
    npos = 0
    while npos < int(rng.integers(1, 9)):
        k = int(rng.integers(0, 10))
        if k <= 3:
            c = letters[int(rng.integers(0, len(letters)))]
            out.append(b"\\" + bytes([c]) if c in b".-]" else bytes([c]))
        elif k <= 6:
            members = bytes(letters[int(i)] for i in rng.integers(0, 13, size=int(rng.integers(1, 4))))
            body = b"".join(b"\\" + bytes([m]) if m in b"]-\\^" else bytes([m]) for m in members)
            if rng.random() < 0.3:
                body += b"a-c" if rng.random() < 0.5 else b"0-9"
            out.append(b"[" + body + b"]")
        elif k == 7:
            out.append([b"\\d", b"\\w", b"\\s"][int(rng.integers(0, 3))])
        elif k == 8 and out and not out[-1].endswith(b"}") and out[-1] != b"(" and out[-1] != b")":
            r = int(rng.integers(1, 4))
            out.append(b"{%d}" % r)
            npos += r - 1
            continue
        else:
            c = letters[int(rng.integers(0, 12))]
            out.append(bytes([c]))
        npos += 1
    e = b"".join(out)
    return b"(" + e + b")" if rng.random() < 0.3 else e


def rand_expr2(rng):
    """round 2 syntax: '.', negated classes, and alternations whose alternatives all have the same length"""
    def seq(n):
        out = []
        for _ in range(n):
            k = int(rng.integers(0, 8))
            if k <= 3:
                out.append(bytes([b"abcxyz01 _"[int(rng.integers(0, 10))]]))
            elif k == 4:
                out.append(b".")
            elif k == 5:
                out.append(b"[^" + bytes(b"abc0 \n"[int(i)] for i in rng.integers(0, 6, size=int(rng.integers(1, 3)))).replace(b"\n", b"\\n") + b"]")
            elif k == 6:
                out.append(b"[" + bytes(b"abcxyz"[int(i)] for i in rng.integers(0, 6, size=2)) + b"]")
            else:
                out.append([b"\\D", b"\\W", b"\\S", b"\\d"][int(rng.integers(0, 4))])
        return b"".join(out)
    n = int(rng.integers(1, 5))
    nalt = int(rng.integers(1, 4))
    body = b"|".join(seq(n) for _ in range(nalt))
    shape = int(rng.integers(0, 3))
    if shape == 0:
        return body
    if shape == 1:
        return seq(int(rng.integers(0, 3))) + b"(" + body + b")" + seq(int(rng.integers(0, 3)))
    return b"(?:" + body + b"){2}" if n * 2 * nalt * nalt <= 32 and n <= 2 else b"(" + body + b")"


def test_parsers_and_walk_against_cpython_re(oracle):
    rng = np.random.default_rng(20231)
    alphabet = np.frombuffer(b"abcABC xyz019_-.]\n\n\t", dtype=np.uint8)
    done = 0
    for it in range(600):
        expr = rand_expr(rng) if it % 3 else rand_expr2(rng)
        icase = bool(rng.integers(0, 2))
        try:
            pyre = re.compile(expr, re.IGNORECASE if icase else 0)
        except re.error:
            continue
        try:
            assert_same_language(expr, icase, rng)
        except xsg.XsgError:  # more alternatives than the product takes: not this test's subject
            continue
        data = alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 3000)))].tobytes()
        cs = compile_class_sequence(expr, icase)
        hay = oracle.lower(data).tobytes() if icase else data
        got = oracle.regex_byte_offsets_match(hay, cs).tolist()
        want = [m.start() for m in pyre.finditer(data)]  # leftmost, non-overlapping: the walk of :63-87
        assert got == want, (expr, icase)
        assert oracle.regex_count(hay, cs, False) == len(want)
        # one match per line (skip_to_nl): first match of every line that has one
        per_line = []
        pos = 0
        while True:
            m = pyre.search(data, pos)
            if not m:
                break
            per_line.append(m.start())
            nl = data.find(b"\n", m.end())
            if nl < 0:
                break
            pos = nl + 1
        if not any(cs.accepts(k, 10) for k in range(cs.plen)):
            assert oracle.regex_byte_offsets_match(hay, cs, True).tolist() == per_line, (expr, icase)
            assert oracle.regex_count(hay, cs, True) == len(per_line)
        done += 1
    assert done > 450


def test_the_whole_match_is_group_one_of_the_wrapped_expression(oracle):
    """The reference's walk is RE2::PartialMatch(input, pattern, &match) (search_wrappers.h:71): `match` receives
    capture group ONE, and its own tests hand over an expression that is one group -- re2::RE2("(a[n|m]t)")
    (search_wrappersTest.cpp:78).  XSG_FLAG_REGEX(expr) is defined as that call with "(" + expr + ")": the reported
    offset and the resume point are those of the WHOLE match, and groups inside `expr` group only.  Pinned here
    against an independent engine: CPython's `re` on the wrapped expression, group 1, under the same walk."""
    from xs_oracle import RegexProgram
    text = (b"Sherlock and Sherwood locked the lock, locks and colours; the ant, the amt, ants\n"
            b"Sherwoodlock Sherlocks relocked unlock lockeds\nan amt and an ant\n") * 3
    for expr in (b"Sher(lock|wood)", b"lock(ed|s)?", b"(a[n|m]t)", b"((a[n|m]t))", b"(Sher)(lock|wood)s?", b"col(ou?)r(s)?"):
        wrapped = re.compile(b"(" + expr + b")", re.DOTALL)
        for skip in (False, True):
            want, pos = [], 0
            while True:
                m = wrapped.search(text, pos)
                if m is None:
                    break
                want.append(m.start(1))          # :72 shift = match.data() - input.data()
                pos = m.end(1)                   # :74 shift += match.size()
                if skip:
                    nl = text.find(b"\n", pos)
                    if nl < 0:
                        break
                    pos = nl + 1
            try:
                got = oracle.regex_byte_offsets_match(text, compile_class_sequence(expr)) if not skip else None
            except UnsupportedRegex:
                got = None
            prog_got = oracle.rx_byte_offsets(text, RegexProgram(expr), skip)
            assert prog_got.tolist() == want, (expr, skip)
            if got is not None:
                assert got.tolist() == want, (expr, "class sequence")
    # ... and NOT the reference's answer for an unwrapped expression with an inner group: there group 1 is the inner
    # group, e.g. `Sher(lock|wood)` reports where `lock` / `wood` starts.  Callers wrap (as the reference's tests do).
    inner = re.compile(b"Sher(lock|wood)")
    m = inner.search(text)
    assert m.start(1) == m.start() + 4
    assert oracle.rx_byte_offsets(text, RegexProgram(b"Sher(lock|wood)"), False)[0] == m.start()
