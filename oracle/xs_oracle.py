"""ctypes front-end of the CPU oracle (oracle/xs_oracle.c) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product under x-search_amd/.

`Oracle`     wraps libxs_oracle.so (the restatement; semantics cited per function
             in xs_oracle.h).
`Reference`  wraps oracle/_ref/libxsref.so = the reference's own
             src/string_search/simd_search.cpp compiled unmodified (C++-mangled
             names); only usable on hosts with AVX2.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_LIB = HERE / "libxs_oracle.so"
_REF = HERE / "_ref" / "libxsref.so"

_u64p = C.POINTER(C.c_uint64)


def _stale() -> bool:
    return not _LIB.exists() or _LIB.stat().st_mtime < (HERE / "xs_oracle.c").stat().st_mtime


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is mounted).  Several processes may get here at
    once (bench.py under torch.distributed.run starts one per GPU): the build runs under a file lock and the
    others find the library up to date when they get the lock."""
    need_ref = not _REF.exists() and Path("/root/reference/src/string_search/simd_search.cpp").exists()
    if not (force or _stale() or need_ref):
        return
    import fcntl
    with open(HERE / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or _stale():
                subprocess.check_call(["make", "-C", str(HERE), "--no-print-directory"], stdout=subprocess.DEVNULL)
            elif not _REF.exists() and need_ref:
                subprocess.check_call(["make", "-C", str(HERE), "--no-print-directory", "ref"], stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def host_has_avx2() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            return " avx2 " in f.read().replace("\n", " ")
    except OSError:
        return False


def _as_bytes(x) -> bytes:
    if isinstance(x, bytes):
        return x
    if isinstance(x, str):
        return x.encode("latin-1")
    if isinstance(x, np.ndarray):
        return x.tobytes()
    return bytes(x)


class _Buf:
    """Keeps a bytes/ndarray alive and exposes (char*, len) without copying arrays."""

    def __init__(self, data):
        if isinstance(data, np.ndarray):
            assert data.dtype == np.uint8 and data.flags.c_contiguous
            self.keep = data
            self.ptr = C.cast(data.ctypes.data, C.c_char_p) if data.size else C.c_char_p(b"")
            self.addr = data.ctypes.data
            self.len = int(data.size)
        else:
            b = _as_bytes(data)
            self.keep = b
            self.ptr = C.c_char_p(b)
            self.addr = C.cast(self.ptr, C.c_void_p).value
            self.len = len(b)


class UnsupportedRegex(ValueError):
    """The expression is not a fixed-length sequence of ASCII byte classes."""


class ClassSeq(C.Structure):
    """xso_classseq: bit b of sets[a * plen + k] set <=> alternative a accepts byte b at position k."""
    _fields_ = [("plen", C.c_uint32), ("nalt", C.c_uint32), ("sets", (C.c_uint32 * 8) * 256)]
    ascii_only = False  # the expression used '.', a negated class or \D \W \S: meaningful on ASCII data only

    def accepts(self, k: int, b: int) -> bool:
        """does ANY alternative accept byte b at position k?"""
        return any((self.sets[a * self.plen + k][b >> 5] >> (b & 31)) & 1 for a in range(max(self.nalt, 1)))


_PUNCT = set(range(0x21, 0x7f)) - set(range(0x30, 0x3a)) - set(range(0x41, 0x5b)) - set(range(0x61, 0x7b))
_ASCII = frozenset(range(0x80))
_ESC_SETS = {ord("d"): set(range(0x30, 0x3a)),
             ord("w"): set(range(0x30, 0x3a)) | set(range(0x41, 0x5b)) | set(range(0x61, 0x7b)) | {0x5f},
             ord("s"): {9, 10, 12, 13, 32},  # RE2: \s == [\t\n\f\r ]
             ord("a"): {7}, ord("f"): {12}, ord("n"): {10}, ord("r"): {13}, ord("t"): {9}, ord("v"): {11}}
_POSIX = {
    b"alnum": _ESC_SETS[ord("w")] - {0x5f}, b"alpha": set(range(0x41, 0x5b)) | set(range(0x61, 0x7b)),
    b"ascii": set(range(0x80)), b"blank": {9, 32}, b"cntrl": set(range(0x20)) | {0x7f},
    b"digit": set(range(0x30, 0x3a)), b"graph": set(range(0x21, 0x7f)), b"lower": set(range(0x61, 0x7b)),
    b"print": set(range(0x20, 0x7f)), b"punct": set(_PUNCT), b"space": {9, 10, 11, 12, 13, 32},
    b"upper": set(range(0x41, 0x5b)), b"word": set(_ESC_SETS[ord("w")]),
    b"xdigit": set(range(0x30, 0x3a)) | set(range(0x41, 0x47)) | set(range(0x61, 0x67)),
}


class _RegexReader:
    """The oracle's own reading of the RE2 syntax subset (written independently of
    x-search_amd/csrc/xsg_classseq.cpp; tests/test_oracle_regex.py checks both against CPython's `re`).
    Grammar:  alt := cat ('|' cat)* ;  cat := piece+ ;  piece := atom ['{' n '}'] ;
              atom := literal | escape | '[' class ']' | '.' | '(' ['?:'] alt ')'
    Every function returns the LIST of alternatives it can stand for, each a tuple of frozensets (one per byte
    position); concatenation is the cross product, alternation the concatenation of the lists."""

    def __init__(self, e: bytes, ignore_case: bool = False):
        self.e, self.i, self.ascii_only, self.icase = e, 0, False, ignore_case

    def negate(self, members) -> frozenset:
        """ASCII complement; under ignore_case the members are closed under case first, as RE2's (?i) does
        ((?i)[^a] excludes 'a' and 'A')"""
        m = set(members)
        if self.icase:
            m |= {b ^ 0x20 for b in m if 0x41 <= (b & ~0x20) <= 0x5a and b < 0x80}
        self.ascii_only = True
        return _ASCII - m

    def peek(self):
        return self.e[self.i] if self.i < len(self.e) else None

    def escape(self):  # self.i is at the character after the backslash
        e = self.e
        if self.i >= len(e):
            raise UnsupportedRegex("trailing backslash")
        c = e[self.i]
        self.i += 1
        if c in _ESC_SETS:
            return frozenset(_ESC_SETS[c])
        if c in (ord("D"), ord("W"), ord("S")):
            return self.negate(_ESC_SETS[c | 0x20])
        if c == ord("x"):
            if e[self.i:self.i + 1] == b"{":
                k = e.find(b"}", self.i + 1)
                body = e[self.i + 1:k] if k > 0 else b""
                if not body or len(body) > 8 or any(ch not in b"0123456789abcdefABCDEF" for ch in body):
                    raise UnsupportedRegex("malformed \\x{...}")
                v, self.i = int(body, 16), k + 1
            else:
                body = e[self.i:self.i + 2]
                if len(body) != 2 or any(ch not in b"0123456789abcdefABCDEF" for ch in body):
                    raise UnsupportedRegex("malformed \\xHH")
                v, self.i = int(body, 16), self.i + 2
            if v > 0x7f:
                raise UnsupportedRegex("code point above 0x7f")
            return frozenset({v})
        if c in _PUNCT:
            return frozenset({c})
        raise UnsupportedRegex(f"escape \\{chr(c)!r}")

    def bracket(self):  # self.i is just behind '['
        e = self.e
        neg = e[self.i:self.i + 1] == b"^"
        if neg:
            self.i += 1
        members, first = set(), True
        while True:
            if self.i >= len(e):
                raise UnsupportedRegex("missing ]")
            m = e[self.i]
            if m == ord("]") and not first:
                self.i += 1
                break
            first = False
            if m == ord("[") and e[self.i + 1:self.i + 2] == b":":
                k = e.find(b":]", self.i + 2)
                if k < 0:
                    raise UnsupportedRegex("malformed [:class:]")
                name = e[self.i + 2:k]
                pneg = name[:1] == b"^"
                name = name[1:] if pneg else name
                if name not in _POSIX:
                    raise UnsupportedRegex("unknown posix class")
                members |= self.negate(_POSIX[name]) if pneg else _POSIX[name]
                self.i = k + 2
                continue
            if m >= 0x80:
                raise UnsupportedRegex("non-ASCII class member")
            self.i += 1
            lo = self.escape() if m == ord("\\") else frozenset({m})
            if len(lo) == 1 and e[self.i:self.i + 1] == b"-" and self.i + 1 < len(e) and e[self.i + 1] != ord("]"):
                h = e[self.i + 1]
                self.i += 2
                if h >= 0x80:
                    raise UnsupportedRegex("non-ASCII class member")
                hi = self.escape() if h == ord("\\") else frozenset({h})
                if len(hi) != 1 or min(hi) < min(lo):
                    raise UnsupportedRegex("bad range")
                members |= set(range(min(lo), min(hi) + 1))
            else:
                members |= lo
        if neg:
            members = set(self.negate(members))
        if not members:
            raise UnsupportedRegex("empty class")
        return frozenset(members)

    def atom(self):
        e, c = self.e, self.peek()
        if c == ord("("):
            self.i += 1
            if e[self.i:self.i + 1] == b"?":
                if e[self.i:self.i + 2] != b"?:":
                    raise UnsupportedRegex("(?")
                self.i += 2
            alts = self.alt()
            if self.peek() != ord(")"):
                raise UnsupportedRegex("missing )")
            self.i += 1
            return alts
        if c == ord("["):
            self.i += 1
            return [(self.bracket(),)]
        if c == ord("\\"):
            self.i += 1
            return [(self.escape(),)]
        if c == ord("."):
            self.i += 1
            self.ascii_only = True
            return [(_ASCII - {10},)]
        if c in b"*+?^${":
            raise UnsupportedRegex(f"operator {chr(c)!r}")
        if c >= 0x80:
            n = 4 if c >= 0xf0 else 3 if c >= 0xe0 else 2
            try:
                e[self.i:self.i + n].decode("utf-8")
            except UnicodeDecodeError:
                raise UnsupportedRegex("pattern is not valid UTF-8")
            seq = tuple(frozenset({b}) for b in e[self.i:self.i + n])
            self.i += n
            return [seq]
        self.i += 1
        return [(frozenset({c}),)]

    @staticmethod
    def cross(left, right):
        out = [a + b for a in left for b in right]
        if len(out) > 64 or any(len(s) > 32 for s in out):
            raise UnsupportedRegex("too many alternatives or positions")
        return out

    def piece(self):
        alts = self.atom()
        if self.peek() is not None and self.peek() in b"*+?":
            raise UnsupportedRegex("repetition operator")
        if self.peek() == ord("{"):
            k = self.e.find(b"}", self.i)
            body = self.e[self.i + 1:k] if k > 0 else b""
            if not body.isdigit() or len(body) > 4 or int(body) == 0:
                raise UnsupportedRegex("only x{n}, n >= 1")
            self.i = k + 1
            if self.peek() is not None and self.peek() in b"*+?{":
                raise UnsupportedRegex("stacked quantifiers")
            base = alts
            for _ in range(int(body) - 1):
                alts = self.cross(alts, base)
        return alts

    def cat(self):
        alts, any_piece = [()], False
        while self.peek() is not None and self.peek() not in b"|)":
            alts = self.cross(alts, self.piece())
            any_piece = True
        if not any_piece:
            raise UnsupportedRegex("empty expression or alternative")
        return alts

    def alt(self):
        alts = self.cat()
        while self.peek() == ord("|"):
            self.i += 1
            alts = alts + self.cat()
            if len(alts) > 64:
                raise UnsupportedRegex("too many alternatives")
        return alts


def compile_class_sequence(expr: bytes, ignore_case: bool = False) -> ClassSeq:
    """expression -> ClassSeq (all alternatives, every one `plen` positions long); raises UnsupportedRegex for
    anything else.  ignore_case: as for literals -- toLower on the data and on every set
    (src/utils/string_utils.cpp:11-33).  `.ascii_only` is set when the expression used '.', a negated class or
    \\D \\W \\S: the oracle's walks then refuse non-ASCII data, like the product."""
    rd = _RegexReader(bytes(expr), ignore_case)
    if not rd.e:
        raise UnsupportedRegex("empty expression")
    alts = rd.alt()
    if rd.i < len(rd.e):
        raise UnsupportedRegex("unmatched )" if rd.peek() == ord(")") else "trailing garbage")
    plen = len(alts[0])
    if any(len(a) != plen for a in alts):
        raise UnsupportedRegex("alternatives of different lengths")
    alts = list(dict.fromkeys(alts))  # duplicates change nothing
    if plen == 0 or plen > 32 or len(alts) * plen > 256:
        raise UnsupportedRegex("empty, longer than 32 positions, or more than 256 sets")
    cs = ClassSeq()
    cs.plen, cs.nalt = plen, len(alts)
    cs.ascii_only = rd.ascii_only
    for a, seq in enumerate(alts):
        for k, st in enumerate(seq):
            if ignore_case:
                st = {b + 32 if 0x41 <= b <= 0x5a else b for b in st}
            for b in st:
                cs.sets[a * plen + k][b >> 5] |= 1 << (b & 31)
    return cs


class _TreeReader(_RegexReader):
    """The variable-length half of the syntax: x* x+ x? x{n} x{n,} x{n,m}, their lazy
    forms, alternatives of any lengths.  RE2 is absent from the snapshot (un-vendored, unpinned submodule), so
    `RE2::PartialMatch` is stood in for by CPython's backtracking `re`, which implements the same published
    semantics for this syntax (leftmost match; among the matches at that offset the first in priority order:
    greedy prefers more, lazy less, `a|b` prefers a -- RE2's "leftmost-first", re2/re2.h) by a different algorithm
    than the product's automata.  The expression is NOT handed to `re` as written: this reader re-emits it with every
    atom as an explicit byte class (RE2's ASCII definitions of \\d \\w \\s [:classes:], its case folding, '.' as any
    ASCII byte but \\n), so only the operators' semantics are CPython's.
    Every method returns (python_source, minlen, sets)."""

    def __init__(self, e: bytes, ignore_case: bool = False):
        super().__init__(e, ignore_case)
        self.sets = []

    def emit_set(self, members) -> tuple:
        m = set(members)
        if self.icase:  # the sets are closed under ASCII case; the data is searched as it is
            m |= {b ^ 0x20 for b in m if 0x41 <= (b & ~0x20) <= 0x5a and b < 0x80}
        self.sets.append(frozenset(m))
        return b"[" + b"".join(b"\\x%02x" % b for b in sorted(m)) + b"]", 1

    def t_atom(self):
        e, c = self.e, self.peek()
        if c == ord("("):
            self.i += 1
            if e[self.i:self.i + 1] == b"?":
                if e[self.i:self.i + 2] != b"?:":
                    raise UnsupportedRegex("(?")
                self.i += 2
            src, mn = self.t_alt()
            if self.peek() != ord(")"):
                raise UnsupportedRegex("missing )")
            self.i += 1
            return b"(?:" + src + b")", mn
        if c == ord("["):
            self.i += 1
            return self.emit_set(self.bracket())
        if c == ord("\\"):
            self.i += 1
            return self.emit_set(self.escape())
        if c == ord("."):
            self.i += 1
            self.ascii_only = True
            return self.emit_set(_ASCII - {10})
        if c in b"*+?^${":
            raise UnsupportedRegex(f"operator {chr(c)!r}")
        if c >= 0x80:
            n = 4 if c >= 0xf0 else 3 if c >= 0xe0 else 2
            try:
                e[self.i:self.i + n].decode("utf-8")
            except UnicodeDecodeError:
                raise UnsupportedRegex("pattern is not valid UTF-8")
            parts = [self.emit_set({b})[0] for b in e[self.i:self.i + n]]
            self.i += n
            return b"(?:" + b"".join(parts) + b")", n
        self.i += 1
        return self.emit_set({c})

    def t_piece(self):
        src, mn = self.t_atom()
        c = self.peek()
        lo = hi = None
        if c is not None and c in b"*+?":
            self.i += 1
            lo, hi = (0, None) if c == ord("*") else (1, None) if c == ord("+") else (0, 1)
            q = bytes([c])
        elif c == ord("{"):
            k = self.e.find(b"}", self.i)
            body = self.e[self.i + 1:k] if k > 0 else b""
            parts = body.split(b",")
            if len(parts) > 2 or not parts[0].isdigit() or len(parts[0]) > 4 or (len(parts) == 2 and parts[1] and
                                                                                 (not parts[1].isdigit() or len(parts[1]) > 4)):
                raise UnsupportedRegex("malformed repetition")
            lo = int(parts[0])
            hi = lo if len(parts) == 1 else (int(parts[1]) if parts[1] else None)
            if (hi is not None and (hi < lo or hi == 0)) or lo > 1000 or (hi or 0) > 1000:
                raise UnsupportedRegex("bad repetition count")
            self.i = k + 1
            q = b"{" + body + b"}"
        else:
            return src, mn
        if self.peek() == ord("?"):
            self.i += 1
            q += b"?"
        if self.peek() is not None and self.peek() in b"*+?{":
            raise UnsupportedRegex("stacked quantifiers")
        if mn == 0:
            raise UnsupportedRegex("repetition of something that can match the empty string")
        return b"(?:" + src + b")" + q, mn * lo

    def t_cat(self):
        parts, total = [], 0
        while self.peek() is not None and self.peek() not in b"|)":
            src, mn = self.t_piece()
            parts.append(src)
            total += mn
        if not parts:
            raise UnsupportedRegex("empty expression or alternative")
        return b"".join(parts), total

    def t_alt(self):
        alts = [self.t_cat()]
        while self.peek() == ord("|"):
            self.i += 1
            alts.append(self.t_cat())
        if len(alts) == 1:
            return alts[0]
        return b"(?:" + b"|".join(a[0] for a in alts) + b")", min(a[1] for a in alts)


class RegexProgram:
    """A variable-length expression as the oracle searches it: `.re` (compiled CPython pattern over explicit byte
    classes), `.ascii_only`, `.minlen`, `.multiline`.  Raises UnsupportedRegex for what the product refuses too:
    expressions that can match the empty string, anchors, flags."""

    def __init__(self, expr: bytes, ignore_case: bool = False):
        import re as _re
        rd = _TreeReader(bytes(expr), ignore_case)
        if not rd.e:
            raise UnsupportedRegex("empty expression")
        src, mn = rd.t_alt()
        if rd.i < len(rd.e):
            raise UnsupportedRegex("unmatched )" if rd.peek() == ord(")") else "trailing garbage")
        if mn == 0:
            raise UnsupportedRegex("the expression can match the empty string")
        # a set that accepts '\n' lets a match span lines: the match walks only (the line walks are refused, as for a
        # literal pattern that contains '\n')
        self.multiline = any(10 in st for st in rd.sets)
        self.source = src
        self.re = _re.compile(src, _re.DOTALL)
        self.ascii_only = rd.ascii_only
        self.minlen = mn


class Oracle:
    def __init__(self):
        build()
        self.lib = lib = C.CDLL(str(_LIB))
        cp, sz, i64, u64, ci = C.c_char_p, C.c_size_t, C.c_int64, C.c_uint64, C.c_int
        vp = C.c_void_p
        lib.xso_set_exact.argtypes = [ci]
        lib.xso_get_exact.restype = ci
        lib.xso_use_primitives.argtypes = [vp, vp]
        for name in ("xso_scalar_strstr", "xso_strstr"):
            getattr(lib, name).argtypes = [vp, sz, cp, sz]
            getattr(lib, name).restype = vp
        lib.xso_strchr.argtypes = [vp, sz, C.c_char]
        lib.xso_strchr.restype = vp
        lib.xso_find_next.argtypes = [cp, sz, vp, sz, sz]
        lib.xso_find_next.restype = i64
        lib.xso_find_next_newline.argtypes = [vp, sz, sz]
        lib.xso_find_next_newline.restype = i64
        for name in ("xso_count_matching_lines", "xso_count_matches"):
            getattr(lib, name).argtypes = [cp, sz, vp, sz]
            getattr(lib, name).restype = u64
        lib.xso_byte_offsets_match.argtypes = [vp, sz, cp, sz, ci, _u64p, u64]
        lib.xso_byte_offsets_match.restype = u64
        lib.xso_byte_offsets_line.argtypes = [vp, sz, cp, sz, _u64p, u64]
        lib.xso_byte_offsets_line.restype = u64
        lib.xso_count.argtypes = [vp, sz, cp, sz, ci]
        lib.xso_count.restype = u64
        lib.xso_lines.argtypes = [vp, sz, cp, sz, _u64p, _u64p, u64]
        lib.xso_lines.restype = u64
        lib.xso_line_indices.argtypes = [vp, sz, cp, sz, u64, _u64p, u64]
        lib.xso_line_indices.restype = u64
        lib.xso_count_newlines.argtypes = [vp, sz]
        lib.xso_count_newlines.restype = u64
        lib.xso_to_lower.argtypes = [vp, sz]
        csp = C.POINTER(ClassSeq)
        lib.xso_regex_byte_offsets_match.argtypes = [vp, sz, csp, ci, _u64p, u64]
        lib.xso_regex_byte_offsets_line.argtypes = [vp, sz, csp, _u64p, u64]
        lib.xso_regex_count.argtypes = [vp, sz, csp, ci]
        lib.xso_regex_lines.argtypes = [vp, sz, csp, _u64p, _u64p, u64]
        lib.xso_regex_line_indices.argtypes = [vp, sz, csp, u64, _u64p, u64]
        for name in ("xso_regex_byte_offsets_match", "xso_regex_byte_offsets_line", "xso_regex_count",
                     "xso_regex_lines", "xso_regex_line_indices"):
            getattr(lib, name).restype = u64
        lib.xso_count_chunks_mt.argtypes = [vp, _u64p, _u64p, u64, cp, sz, ci, ci, _u64p]
        lib.xso_count_chunks_mt.restype = u64
        lib.xso_pool_create.argtypes = [ci, ci]
        lib.xso_pool_create.restype = vp
        lib.xso_pool_destroy.argtypes = [vp]
        lib.xso_corpus_alloc.argtypes = [u64]
        lib.xso_corpus_alloc.restype = vp
        lib.xso_corpus_free.argtypes = [vp, u64]
        lib.xso_pool_replicate.argtypes = [vp, vp, _u64p, _u64p, u64, _u64p, C.POINTER(vp)]
        lib.xso_pool_count_chunks.argtypes = [vp, vp, _u64p, _u64p, u64, cp, sz, ci, ci, _u64p, C.POINTER(C.c_double)]
        lib.xso_pool_count_chunks.restype = u64

    # -- configuration ------------------------------------------------------
    def set_exact(self, exact: bool) -> None:
        self.lib.xso_set_exact(1 if exact else 0)

    def use_reference_primitives(self, ref: "Reference | None") -> None:
        """Route the restated wrappers through oracle/_ref's findNext/findNextNewLine."""
        if ref is None:
            self.lib.xso_use_primitives(None, None)
        else:
            self.lib.xso_use_primitives(C.cast(ref.fn_findnext, C.c_void_p), C.cast(ref.fn_findnl, C.c_void_p))

    def lower(self, data) -> np.ndarray:
        """simd::toLower on a copy (string_utils.cpp:11-33)."""
        a = np.frombuffer(_as_bytes(data), dtype=np.uint8).copy() if not isinstance(data, np.ndarray) else data.copy()
        if a.size:
            self.lib.xso_to_lower(a.ctypes.data, a.size)
        return a

    # -- primitives ---------------------------------------------------------
    def strstr(self, data, pat) -> int:
        """offset of the match or -1 (the C function returns a pointer)."""
        b, p = _Buf(data), _as_bytes(pat)
        r = self.lib.xso_strstr(b.addr, b.len, p, len(p))
        return -1 if not r else r - b.addr

    def scalar_strstr(self, data, pat) -> int:
        b, p = _Buf(data), _as_bytes(pat)
        r = self.lib.xso_scalar_strstr(b.addr, b.len, p, len(p))
        return -1 if not r else r - b.addr

    def strchr(self, data, c) -> int:
        b = _Buf(data)
        r = self.lib.xso_strchr(b.addr, b.len, _as_bytes(c))
        return -1 if not r else r - b.addr

    def find_next(self, pat, data, shift=0) -> int:
        b, p = _Buf(data), _as_bytes(pat)
        return self.lib.xso_find_next(p, len(p), b.addr, b.len, shift)

    def find_next_newline(self, data, shift=0) -> int:
        b = _Buf(data)
        return self.lib.xso_find_next_newline(b.addr, b.len, shift)

    def count_matches(self, pat, data) -> int:
        b, p = _Buf(data), _as_bytes(pat)
        return self.lib.xso_count_matches(p, len(p), b.addr, b.len)

    def count_matching_lines(self, pat, data) -> int:
        b, p = _Buf(data), _as_bytes(pat)
        return self.lib.xso_count_matching_lines(p, len(p), b.addr, b.len)

    # -- wrappers -----------------------------------------------------------
    def _list(self, fn, *args) -> np.ndarray:
        n = fn(*args, None, 0)
        out = np.empty(n, dtype=np.uint64)
        if n:
            fn(*args, out.ctypes.data_as(_u64p), n)
        return out

    def byte_offsets_match(self, data, pat, skip_to_nl=False) -> np.ndarray:
        b, p = _Buf(data), _as_bytes(pat)
        return self._list(self.lib.xso_byte_offsets_match, b.addr, b.len, p, len(p), 1 if skip_to_nl else 0)

    def byte_offsets_line(self, data, pat) -> np.ndarray:
        b, p = _Buf(data), _as_bytes(pat)
        return self._list(self.lib.xso_byte_offsets_line, b.addr, b.len, p, len(p))

    def count(self, data, pat, skip_to_nl=True) -> int:
        b, p = _Buf(data), _as_bytes(pat)
        return self.lib.xso_count(b.addr, b.len, p, len(p), 1 if skip_to_nl else 0)

    def lines_spans(self, data, pat):
        b, p = _Buf(data), _as_bytes(pat)
        n = self.lib.xso_lines(b.addr, b.len, p, len(p), None, None, 0)
        beg = np.empty(n, dtype=np.uint64)
        ln = np.empty(n, dtype=np.uint64)
        if n:
            self.lib.xso_lines(b.addr, b.len, p, len(p), beg.ctypes.data_as(_u64p), ln.ctypes.data_as(_u64p), n)
        return beg, ln

    def lines(self, data, pat) -> list[bytes]:
        raw = _as_bytes(data) if not isinstance(data, np.ndarray) else data
        beg, ln = self.lines_spans(data, pat)
        if isinstance(raw, np.ndarray):
            return [raw[int(s):int(s + l)].tobytes() for s, l in zip(beg, ln)]
        return [raw[int(s):int(s + l)] for s, l in zip(beg, ln)]

    def line_indices(self, data, pat, line_base=0) -> np.ndarray:
        b, p = _Buf(data), _as_bytes(pat)
        return self._list(self.lib.xso_line_indices, b.addr, b.len, p, len(p), int(line_base))

    # -- regex wrappers (class sequences; cs = compile_class_sequence(expr)) ----
    @staticmethod
    def _ascii_guard(data, cs):
        """an ascii_only expression ('.', negated classes) has no byte-per-position meaning on non-ASCII data:
        refused, like the product does"""
        if getattr(cs, "ascii_only", False):
            a = data if isinstance(data, np.ndarray) else np.frombuffer(_as_bytes(data), dtype=np.uint8)
            if a.size and int(a.max()) >= 0x80:
                raise UnsupportedRegex("ascii-only expression on non-ASCII data")

    def regex_byte_offsets_match(self, data, cs, skip_to_nl=False) -> np.ndarray:
        self._ascii_guard(data, cs)
        b = _Buf(data)
        return self._list(self.lib.xso_regex_byte_offsets_match, b.addr, b.len, C.byref(cs), 1 if skip_to_nl else 0)

    def regex_byte_offsets_line(self, data, cs) -> np.ndarray:
        self._ascii_guard(data, cs)
        b = _Buf(data)
        return self._list(self.lib.xso_regex_byte_offsets_line, b.addr, b.len, C.byref(cs))

    def regex_count(self, data, cs, skip_to_nl=True) -> int:
        self._ascii_guard(data, cs)
        b = _Buf(data)
        return self.lib.xso_regex_count(b.addr, b.len, C.byref(cs), 1 if skip_to_nl else 0)

    def regex_lines_spans(self, data, cs):
        self._ascii_guard(data, cs)
        b = _Buf(data)
        n = self.lib.xso_regex_lines(b.addr, b.len, C.byref(cs), None, None, 0)
        beg = np.empty(n, dtype=np.uint64)
        ln = np.empty(n, dtype=np.uint64)
        if n:
            self.lib.xso_regex_lines(b.addr, b.len, C.byref(cs), beg.ctypes.data_as(_u64p), ln.ctypes.data_as(_u64p), n)
        return beg, ln

    def regex_line_indices(self, data, cs, line_base=0) -> np.ndarray:
        self._ascii_guard(data, cs)
        b = _Buf(data)
        return self._list(self.lib.xso_regex_line_indices, b.addr, b.len, C.byref(cs), int(line_base))

    # -- regex wrappers, variable-length expressions (prog = RegexProgram(expr)): the reference's walks
    #    (search_wrappers.h:63-87, 209-271) line by line, `prog.re.search` standing in for RE2::PartialMatch ----------
    def rx_byte_offsets(self, data, prog, skip_to_nl=False, as_line_start=False) -> np.ndarray:
        """_regex_byte_offsets (:63-87); as_line_start = the func of regex::byte_offsets_line (:220-225)"""
        self._ascii_guard(data, prog)
        d = _as_bytes(data)
        out, pos = [], 0
        while True:
            m = prog.re.search(d, pos)  # :71 PartialMatch(input, pattern, &match) with input = data[pos:]
            if m is None:
                break
            v = m.start()  # :72-73
            out.append(d.rfind(b"\n", 0, v) + 1 if as_line_start else v)  # previous_new_line_offset_relative_to_match (:111-123)
            pos = m.end()  # :74-75
            if skip_to_nl:  # :76-84
                nl = d.find(b"\n", pos)
                if nl < 0:
                    break
                pos = nl + 1
        return np.asarray(out, dtype=np.uint64)

    def rx_count(self, data, prog, skip_to_nl=True) -> int:
        """regex::count (:250-269): the same walk, counting"""
        return int(self.rx_byte_offsets(data, prog, skip_to_nl).size)

    def rx_lines_spans(self, data, prog):
        """xs::lines with a regex (no wrapper in the snapshot: the literal `line` walk, :187-207, with the regex find)"""
        self._ascii_guard(data, prog)
        d = _as_bytes(data)
        beg, ln, pos = [], [], 0
        while pos < len(d):
            m = prog.re.search(d, pos)
            if m is None:
                break
            b = d.rfind(b"\n", 0, m.start()) + 1
            e = d.find(b"\n", m.end())
            if e < 0:
                break
            pos = e + 1
            beg.append(b)
            ln.append(e - b)
        return np.asarray(beg, dtype=np.uint64), np.asarray(ln, dtype=np.uint64)

    def rx_line_indices(self, data, prog, line_base=0) -> np.ndarray:
        """xs::line_indices with a regex: number of newlines before the line start (inferred, as for literals)"""
        self._ascii_guard(data, prog)
        d = _as_bytes(data)
        out, pos, counted_to, seen = [], 0, 0, 0
        while pos < len(d):
            m = prog.re.search(d, pos)
            if m is None:
                break
            b = d.rfind(b"\n", 0, m.start()) + 1
            seen += d.count(b"\n", counted_to, b)
            counted_to = b
            out.append(line_base + seen)
            e = d.find(b"\n", m.end())
            if e < 0:
                break
            pos = e + 1
        return np.asarray(out, dtype=np.uint64)

    def count_newlines(self, data) -> int:
        b = _Buf(data)
        return self.lib.xso_count_newlines(b.addr, b.len)

    def count_chunks_mt(self, data: np.ndarray, offsets, lengths, pat, skip_to_nl=False, nthreads=1):
        b, p = _Buf(data), _as_bytes(pat)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        per = np.zeros(len(off), dtype=np.uint64)
        tot = self.lib.xso_count_chunks_mt(b.addr, off.ctypes.data_as(_u64p), ln.ctypes.data_as(_u64p), len(off), p,
                                           len(p), 1 if skip_to_nl else 0, int(nthreads), per.ctypes.data_as(_u64p))
        return int(tot), per


class CpuPoolCorpus:
    """bench.py's cpu_baseline leg: `nchunks` chunks replicated from template blocks into untouched anonymous
    memory by the pool's own workers (first touch), searched by persistent worker threads
    (xs_oracle.c: xso_pool_*; the worker loop of include/xsearch/Searcher.h:100-120)."""

    def __init__(self, oracle: "Oracle", blocks, plan, nthreads_touch: int):
        self.o = oracle
        lib = oracle.lib
        self.blocks = [np.ascontiguousarray(b) for b in blocks]
        lens = np.array([self.blocks[int(t)].size for t in plan], dtype=np.uint64)
        padded = (lens + np.uint64(4095)) // np.uint64(4096) * np.uint64(4096)
        self.off = np.concatenate([[np.uint64(0)], np.cumsum(padded)[:-1]]).astype(np.uint64)
        self.len = lens
        self.cap = int(padded.sum())
        self.n = len(lens)
        self.base = lib.xso_corpus_alloc(self.cap)
        if not self.base:
            raise MemoryError(f"cannot map {self.cap} bytes")
        src_idx = np.ascontiguousarray(plan, dtype=np.uint64)
        ptrs = (C.c_void_p * len(self.blocks))(*[b.ctypes.data for b in self.blocks])
        pool = lib.xso_pool_create(int(nthreads_touch), 1)
        lib.xso_pool_replicate(pool, self.base, self.off.ctypes.data_as(_u64p), self.len.ctypes.data_as(_u64p), self.n,
                               src_idx.ctypes.data_as(_u64p), ptrs)
        lib.xso_pool_destroy(pool)

    def count(self, pat: bytes, nthreads: int, passes: int, first_chunks: int | None = None, skip_to_nl=False):
        """-> (total of one pass, seconds for all passes, bytes per pass)"""
        lib = self.o.lib
        n = self.n if first_chunks is None else min(self.n, int(first_chunks))
        pool = lib.xso_pool_create(int(nthreads), 1)
        sec = C.c_double(0)
        p = _as_bytes(pat)
        # one untimed pass first: thread start-up and page-table warm-up are not the search
        lib.xso_pool_count_chunks(pool, self.base, self.off.ctypes.data_as(_u64p), self.len.ctypes.data_as(_u64p), n, p,
                                  len(p), 1 if skip_to_nl else 0, 1, None, C.byref(sec))
        tot = lib.xso_pool_count_chunks(pool, self.base, self.off.ctypes.data_as(_u64p), self.len.ctypes.data_as(_u64p),
                                        n, p, len(p), 1 if skip_to_nl else 0, int(passes), None, C.byref(sec))
        lib.xso_pool_destroy(pool)
        if tot == (1 << 64) - 1:
            raise RuntimeError("CPU passes disagree")
        return int(tot), float(sec.value), int(self.len[:n].sum())

    def close(self):
        if self.base:
            self.o.lib.xso_corpus_free(self.base, self.cap)
            self.base = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Reference:
    """The reference's own simd_search.cpp, compiled unmodified (oracle/Makefile)."""

    SYM = {
        "findNext": "_ZN2xs6search4simd8findNextEPKcmS3_mm",
        "findNextNewLine": "_ZN2xs6search4simd15findNextNewLineEPKcmm",
        "countMatches": "_ZN2xs6search4simd12countMatchesEPKcmS3_m",
        "countMatchingLines": "_ZN2xs6search4simd18countMatchingLinesEPKcmS3_m",
        "strstr": "_ZN2xs6search4simd6strstrEPKcmS3_m",
        "strchr": "_ZN2xs6search4simd6strchrEPKcmc",
        "scalar_strstr": "_ZN2xs6search4simd13scalar_strstrEPKcmS3_m",
        "strcasestr": "_ZN2xs6search4simd10strcasestrEPKcmS3_m",  # simd_search.cpp:220-287 (no caller in the snapshot)
    }

    @staticmethod
    def available() -> bool:
        build()
        return _REF.exists() and host_has_avx2()

    def __init__(self):
        if not Reference.available():
            raise RuntimeError("oracle/_ref/libxsref.so missing or host lacks AVX2")
        self.lib = lib = C.CDLL(str(_REF))
        cp, sz, i64, u64, vp = C.c_char_p, C.c_size_t, C.c_int64, C.c_uint64, C.c_void_p
        self.fn_findnext = getattr(lib, self.SYM["findNext"])
        self.fn_findnext.argtypes = [cp, sz, vp, sz, sz]
        self.fn_findnext.restype = i64
        self.fn_findnl = getattr(lib, self.SYM["findNextNewLine"])
        self.fn_findnl.argtypes = [vp, sz, sz]
        self.fn_findnl.restype = i64
        self.fn_count = getattr(lib, self.SYM["countMatches"])
        self.fn_count.argtypes = [cp, sz, vp, sz]
        self.fn_count.restype = u64
        self.fn_count_lines = getattr(lib, self.SYM["countMatchingLines"])
        self.fn_count_lines.argtypes = [cp, sz, vp, sz]
        self.fn_count_lines.restype = u64
        self.fn_strstr = getattr(lib, self.SYM["strstr"])
        self.fn_strstr.argtypes = [vp, sz, cp, sz]
        self.fn_strstr.restype = vp
        self.fn_strchr = getattr(lib, self.SYM["strchr"])
        self.fn_strchr.argtypes = [vp, sz, C.c_char]
        self.fn_strchr.restype = vp
        self.fn_scalar_strstr = getattr(lib, self.SYM["scalar_strstr"])
        self.fn_scalar_strstr.argtypes = [vp, sz, cp, sz]
        self.fn_scalar_strstr.restype = vp
        self.fn_strcasestr = getattr(lib, self.SYM["strcasestr"])
        self.fn_strcasestr.argtypes = [vp, sz, cp, sz]
        self.fn_strcasestr.restype = vp

    def strcasestr(self, data, pat) -> int:
        """Offset of simd::strcasestr's result or -1.  Patterns of 2+ bytes only: with one byte the reference
        passes pattern_len - 2 == SIZE_MAX to compare_case_insensitive (simd_search.cpp:211)."""
        b, p = _Buf(data), _as_bytes(pat)
        assert len(p) >= 2
        r = self.fn_strcasestr(b.addr, b.len, p, len(p))
        return -1 if not r else r - b.addr

    def find_next(self, pat, data, shift=0) -> int:
        b, p = _Buf(data), _as_bytes(pat)
        return self.fn_findnext(p, len(p), b.addr, b.len, shift)

    def find_next_newline(self, data, shift=0) -> int:
        b = _Buf(data)
        return self.fn_findnl(b.addr, b.len, shift)

    def count_matches(self, pat, data) -> int:
        b, p = _Buf(data), _as_bytes(pat)
        return self.fn_count(p, len(p), b.addr, b.len)

    def count_matching_lines(self, pat, data) -> int:
        b, p = _Buf(data), _as_bytes(pat)
        return self.fn_count_lines(p, len(p), b.addr, b.len)

    def strstr(self, data, pat) -> int:
        b, p = _Buf(data), _as_bytes(pat)
        r = self.fn_strstr(b.addr, b.len, p, len(p))
        return -1 if not r else r - b.addr

    def strchr(self, data, c) -> int:
        b = _Buf(data)
        r = self.fn_strchr(b.addr, b.len, _as_bytes(c))
        return -1 if not r else r - b.addr

    def scalar_strstr(self, data, pat) -> int:
        b, p = _Buf(data), _as_bytes(pat)
        r = self.fn_scalar_strstr(b.addr, b.len, p, len(p))
        return -1 if not r else r - b.addr
