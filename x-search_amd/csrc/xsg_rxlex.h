// xsg_rxlex.h -- the atom level of the RE2 syntax the GPU matchers read: escapes, bracket classes, POSIX classes,
// shared by the class-sequence parser (xsg_classseq.cpp) and the automaton compiler (xsg_regex.cpp).  Internal.
#pragma once
#include <string>

#include "xsg_classseq.h"

namespace xsg {

struct AtomLexer {
  const uint8_t* re = nullptr;
  size_t n = 0, i = 0;
  std::string* err = nullptr;
  bool ascii_only = false;  // '.', a negated class or \D \W \S was used: exact on ASCII data only
  bool icase = false;

  bool fail(const std::string& m) {
    *err = m + " (at byte " + std::to_string(i) + " of the expression)";
    return false;
  }
  static ByteSet single(uint32_t b) {
    ByteSet s{};
    set_add(s, b);
    return s;
  }
  static void add_range(ByteSet& s, uint32_t lo, uint32_t hi) {
    for (uint32_t b = lo; b <= hi; ++b) set_add(s, b);
  }
  static bool is_punct(uint8_t c) { return c < 0x80 && c > 0x20 && !((c | 0x20) >= 'a' && (c | 0x20) <= 'z') && !(c >= '0' && c <= '9'); }
  static int hexval(uint8_t c) {
    if (c >= '0' && c <= '9') return c - '0';
    if ((c | 0x20) >= 'a' && (c | 0x20) <= 'f') return (c | 0x20) - 'a' + 10;
    return -1;
  }

  // Negation under ignore_case: RE2 folds the listed members first ((?i)[^a] excludes 'a' AND 'A'), so they are
  // closed under ASCII case before the complement is taken.  (Folding the complement afterwards instead would map
  // the surviving 'A' onto 'a' and let [^a] accept 'a'.)
  ByteSet ascii_complement(ByteSet s) const {
    if (icase)
      for (uint32_t b = 'a'; b <= 'z'; ++b)
        if (set_has(s, b) || set_has(s, b - 32)) set_add(s, b), set_add(s, b - 32);
    ByteSet r{};
    for (int q = 0; q < 4; ++q) r[q] = ~s[q];  // bytes 0x00..0x7f only
    return r;
  }

  // after a backslash (i points at the escaped character): a byte set
  bool escape(ByteSet* out) {
    if (i >= n) return fail("trailing backslash");
    const uint8_t c = re[i++];
    ByteSet s{};
    switch (c) {
      case 'd': case 'D': add_range(s, '0', '9'); break;
      case 'w': case 'W': add_range(s, '0', '9'); add_range(s, 'A', 'Z'); add_range(s, 'a', 'z'); set_add(s, '_'); break;
      case 's': case 'S': set_add(s, '\t'); set_add(s, '\n'); set_add(s, '\f'); set_add(s, '\r'); set_add(s, ' '); break;
      case 'a': set_add(s, 7); break;
      case 'f': set_add(s, '\f'); break;
      case 'n': set_add(s, '\n'); break;
      case 'r': set_add(s, '\r'); break;
      case 't': set_add(s, '\t'); break;
      case 'v': set_add(s, 11); break;
      case 'x': {
        uint32_t v = 0;
        if (i < n && re[i] == '{') {
          ++i;
          size_t digits = 0;
          while (i < n && hexval(re[i]) >= 0 && digits < 8) v = v * 16 + (uint32_t)hexval(re[i++]), ++digits;
          if (digits == 0 || i >= n || re[i] != '}') return fail("malformed \\x{...}");
          ++i;
        } else {
          if (i + 2 > n || hexval(re[i]) < 0 || hexval(re[i + 1]) < 0) return fail("malformed \\xHH");
          v = (uint32_t)(hexval(re[i]) * 16 + hexval(re[i + 1]));
          i += 2;
        }
        if (v > 0x7f) return fail("\\x escape above 0x7f is a multi-byte code point in RE2: not supported");
        set_add(s, v);
        break;
      }
      default:
        if (!is_punct(c)) return fail(std::string("escape \\") + (char)c + " is not supported by the GPU matcher");
        set_add(s, c);
    }
    if (c == 'D' || c == 'W' || c == 'S') {
      s = ascii_complement(s);
      ascii_only = true;
    }
    *out = s;
    return true;
  }

  // "[:name:]" / "[:^name:]" at i (pointing at '['): RE2's ASCII classes
  bool posix_class(ByteSet* out) {
    const size_t end = [&] {
      for (size_t j = i + 2; j + 1 < n; ++j)
        if (re[j] == ':' && re[j + 1] == ']') return j;
      return (size_t)0;
    }();
    if (!end) return fail("malformed [:class:]");
    std::string name((const char*)re + i + 2, end - (i + 2));
    bool neg = false;
    if (!name.empty() && name[0] == '^') neg = true, name.erase(0, 1);
    ByteSet s{};
    if (name == "alnum") add_range(s, '0', '9'), add_range(s, 'A', 'Z'), add_range(s, 'a', 'z');
    else if (name == "alpha") add_range(s, 'A', 'Z'), add_range(s, 'a', 'z');
    else if (name == "ascii") add_range(s, 0, 0x7f);
    else if (name == "blank") set_add(s, '\t'), set_add(s, ' ');
    else if (name == "cntrl") add_range(s, 0, 0x1f), set_add(s, 0x7f);
    else if (name == "digit") add_range(s, '0', '9');
    else if (name == "graph") add_range(s, '!', '~');
    else if (name == "lower") add_range(s, 'a', 'z');
    else if (name == "print") add_range(s, ' ', '~');
    else if (name == "punct") add_range(s, '!', '/'), add_range(s, ':', '@'), add_range(s, '[', '`'), add_range(s, '{', '~');
    else if (name == "space") add_range(s, '\t', '\r'), set_add(s, ' ');
    else if (name == "upper") add_range(s, 'A', 'Z');
    else if (name == "word") add_range(s, '0', '9'), add_range(s, 'A', 'Z'), add_range(s, 'a', 'z'), set_add(s, '_');
    else if (name == "xdigit") add_range(s, '0', '9'), add_range(s, 'A', 'F'), add_range(s, 'a', 'f');
    else return fail("unknown [:" + name + ":] class");
    if (neg) {
      s = ascii_complement(s);
      ascii_only = true;
    }
    i = end + 2;
    *out = s;
    return true;
  }

  // i points just behind '['
  bool char_class(ByteSet* out) {
    ByteSet s{};
    bool negated = false;
    if (i < n && re[i] == '^') negated = true, ++i;
    bool first = true;
    for (;;) {
      if (i >= n) return fail("missing ]");
      uint8_t c = re[i];
      if (c == ']' && !first) {
        ++i;
        break;
      }
      first = false;
      if (c == '[' && i + 1 < n && re[i + 1] == ':') {
        ByteSet ps{};
        if (!posix_class(&ps)) return false;
        for (int q = 0; q < 8; ++q) s[q] |= ps[q];
        continue;
      }
      if (c >= 0x80) return fail("non-ASCII class members are not supported");
      ByteSet lo_set{};
      bool lo_is_set = false;  // \d \w \s inside a class
      ++i;
      if (c == '\\') {
        if (!escape(&lo_set)) return false;
        lo_is_set = set_size(lo_set) != 1;
      } else {
        lo_set = single(c);
      }
      // a range?  "x-y" with y not the closing bracket
      if (!lo_is_set && i + 1 < n && re[i] == '-' && re[i + 1] != ']') {
        ++i;
        uint8_t h = re[i++];
        ByteSet hi_set{};
        if (h >= 0x80) return fail("non-ASCII class members are not supported");
        if (h == '\\') {
          if (!escape(&hi_set)) return false;
          if (set_size(hi_set) != 1) return fail("bad class range");
        } else {
          hi_set = single(h);
        }
        const int lo = set_single(lo_set), hi = set_single(hi_set);
        if (hi < lo) return fail("bad class range");
        add_range(s, (uint32_t)lo, (uint32_t)hi);
      } else {
        for (int q = 0; q < 8; ++q) s[q] |= lo_set[q];
      }
    }
    if (negated) {
      s = ascii_complement(s);
      ascii_only = true;
    }
    if (set_size(s) == 0) return fail("empty class");
    *out = s;
    return true;
  }
};

}  // namespace xsg
