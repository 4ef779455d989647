// xsg_classseq.cpp -- parser for the fixed-length class-sequence subset of RE2 syntax.
//
// Accepted (each item stands for exactly one byte position unless it is a multi-byte literal):
//   literal bytes; non-ASCII literals must be well-formed UTF-8 (RE2 parses the pattern as UTF-8) and
//   stand for their bytes
//   \  + ASCII punctuation -> that character;  \a \f \n \r \t \v;  \xHH and \x{H..} up to 0x7f
//   \d = [0-9]   \w = [0-9A-Za-z_]   \s = [\t\n\f\r ]        (RE2's ASCII definitions)
//   [ ... ]  positive class of ASCII members: literals, a-z ranges, the escapes above, a leading ']'
//   atom{n}  n >= 1 copies of a one-byte atom
//   ( ... )  capture groups: transparent (the reference's walk reads group 1, and the API layer wraps the whole
//            expression in one -- search_wrappers.h:70-75 only works with at least one group, the unit tests pass
//            "(a[n|m]t)", the integration tests `She[r ]lock`)
// Refused: . * + ? | ^ $ {n,m} (?...) \b \B \A \z \D \W \S \p \P \Q \C, backslash + letter/digit otherwise,
//   negated classes, [:posix:] classes, class members >= 0x80 ('.', negation and wide members match multi-byte
//   code points in RE2: not one byte per position), more than kMaxClassSeq positions, an empty expression.
#include "xsg_classseq.h"

#include <string.h>

namespace xsg {

uint32_t set_size(const ByteSet& s) {
  uint32_t n = 0;
  for (uint32_t w : s) n += (uint32_t)__builtin_popcount(w);
  return n;
}

int set_single(const ByteSet& s) {
  if (set_size(s) != 1) return -1;
  for (int q = 0; q < 8; ++q)
    if (s[q]) return q * 32 + __builtin_ctz(s[q]);
  return -1;
}

namespace {

struct Parser {
  const uint8_t* re;
  size_t n, i = 0;
  std::vector<ByteSet>* seq;
  std::string* err;
  int depth = 0;

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

  // after a backslash (i points at the escaped character): a byte set
  bool escape(ByteSet* out) {
    if (i >= n) return fail("trailing backslash");
    const uint8_t c = re[i++];
    ByteSet s{};
    switch (c) {
      case 'd': add_range(s, '0', '9'); break;
      case 'w': add_range(s, '0', '9'); add_range(s, 'A', 'Z'); add_range(s, 'a', 'z'); set_add(s, '_'); break;
      case 's': set_add(s, '\t'); set_add(s, '\n'); set_add(s, '\f'); set_add(s, '\r'); set_add(s, ' '); break;
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
    *out = s;
    return true;
  }

  // i points just behind '['
  bool char_class(ByteSet* out) {
    ByteSet s{};
    if (i < n && re[i] == '^') return fail("negated classes match multi-byte code points in RE2: not supported");
    bool first = true;
    for (;;) {
      if (i >= n) return fail("missing ]");
      uint8_t c = re[i];
      if (c == ']' && !first) {
        ++i;
        break;
      }
      first = false;
      if (c == '[' && i + 1 < n && re[i + 1] == ':') return fail("[:posix:] classes are not supported");
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
    if (set_size(s) == 0) return fail("empty class");
    *out = s;
    return true;
  }

  bool push(const ByteSet& s) {
    if (seq->size() >= kMaxClassSeq) return fail("expression longer than " + std::to_string(kMaxClassSeq) + " positions");
    seq->push_back(s);
    return true;
  }

  bool run() {
    bool last_is_atom = false;  // the previous item was a one-byte atom a {n} may follow
    while (i < n) {
      const uint8_t c = re[i];
      if (c == '(') {
        if (i + 1 < n && re[i + 1] == '?') return fail("(?...) groups and flags are not supported");
        ++i, ++depth;
        last_is_atom = false;
        continue;
      }
      if (c == ')') {
        if (depth == 0) return fail("unmatched )");
        ++i, --depth;
        last_is_atom = false;  // a quantifier on a group is refused below
        if (i < n && (re[i] == '{' || re[i] == '*' || re[i] == '+' || re[i] == '?')) return fail("quantified groups are not supported");
        continue;
      }
      if (c == '{') {
        if (!last_is_atom) return fail("{ without a one-byte atom before it");
        size_t j = i + 1;
        uint32_t cnt = 0, digits = 0;
        while (j < n && re[j] >= '0' && re[j] <= '9' && digits < 4) cnt = cnt * 10 + (re[j++] - '0'), ++digits;
        if (digits == 0 || j >= n || re[j] != '}') return fail("only the fixed repetition {n} is supported");
        if (cnt == 0) return fail("{0} is not supported");
        i = j + 1;
        const ByteSet s = seq->back();
        for (uint32_t k = 1; k < cnt; ++k)
          if (!push(s)) return false;
        last_is_atom = false;
        continue;
      }
      if (c == '.' ) return fail("'.' matches multi-byte code points in RE2: not supported");
      if (c == '*' || c == '+' || c == '?' || c == '|' || c == '^' || c == '$')
        return fail(std::string("operator '") + (char)c + "' is not a fixed-length class sequence");
      ByteSet s{};
      if (c == '[') {
        ++i;
        if (!char_class(&s)) return false;
        if (!push(s)) return false;
        last_is_atom = true;
        continue;
      }
      if (c == '\\') {
        ++i;
        if (!escape(&s)) return false;
        if (!push(s)) return false;
        last_is_atom = true;
        continue;
      }
      if (c >= 0x80) {  // one well-formed UTF-8 sequence: its bytes, in order
        const int len = c >= 0xf0 ? 4 : c >= 0xe0 ? 3 : 2;
        if (c < 0xc2 || c > 0xf4 || i + len > n) return fail("pattern is not valid UTF-8");
        for (int k = 1; k < len; ++k)
          if ((re[i + k] & 0xc0) != 0x80) return fail("pattern is not valid UTF-8");
        for (int k = 0; k < len; ++k)
          if (!push(single(re[i + k]))) return false;
        i += len;
        last_is_atom = false;  // {n} would repeat the whole code point
        if (i < n && re[i] == '{') return fail("repetition of a multi-byte character is not supported");
        continue;
      }
      ++i;
      if (!push(single(c))) return false;  // ']' and '}' on their own are literals in RE2 too
      last_is_atom = true;
    }
    if (depth != 0) return fail("missing )");
    if (seq->empty()) return fail("empty expression");
    return true;
  }
};

}  // namespace

bool compile_class_sequence(const uint8_t* re, size_t n, std::vector<ByteSet>* seq, std::string* err) {
  seq->clear();
  Parser p{re, n, 0, seq, err};
  if (!p.run()) {
    seq->clear();
    return false;
  }
  return true;
}

void fold_sets(std::vector<ByteSet>* seq) {
  for (ByteSet& s : *seq)
    for (uint32_t b = 'A'; b <= 'Z'; ++b)
      if (set_has(s, b)) {
        s[b >> 5] &= ~(1u << (b & 31u));
        set_add(s, b + 32);
      }
}

bool sequence_can_overlap(const std::vector<ByteSet>& seq) {
  const size_t n = seq.size();
  for (size_t sh = 1; sh < n; ++sh) {
    bool all = true;
    for (size_t k = 0; k + sh < n && all; ++k) {
      bool meet = false;
      for (int q = 0; q < 8; ++q) meet |= (seq[k][q] & seq[k + sh][q]) != 0;
      all = meet;
    }
    if (all) return true;
  }
  return false;
}

}  // namespace xsg
