// xsg_classseq.cpp -- parser for the regular expressions the scan kernel decides itself: alternations of
// fixed-length class sequences, in RE2 syntax.
//
// Accepted (each item stands for exactly one byte position unless it is a multi-byte literal or a group):
//   literal bytes; non-ASCII literals must be well-formed UTF-8 (RE2 parses the pattern as UTF-8) and
//   stand for their bytes
//   \  + ASCII punctuation -> that character;  \a \f \n \r \t \v;  \xHH and \x{H..} up to 0x7f
//   \d = [0-9]   \w = [0-9A-Za-z_]   \s = [\t\n\f\r ]        (RE2's ASCII definitions)
//   [ ... ]  class of ASCII members: literals, a-z ranges, the escapes above, [:alpha:] & co, a leading ']'
//   .  [^ ... ]  \D \W \S   -> the ASCII bytes they accept ('.' excludes '\n', a negated class does not); they
//            mark the expression `ascii_only`: RE2 matches whole code points there, so data with a byte >= 0x80
//            is refused at search time rather than decided differently
//   x{n}     n >= 1 copies of an atom or group
//   ( ... ) (?: ... )  groups: transparent (the reference's walk reads group 1, and the API layer wraps the whole
//            expression in one -- search_wrappers.h:70-75 only works with at least one group, the unit tests pass
//            "(a[n|m]t)", the integration tests `She[r ]lock`)
//   a|b      alternation at any depth, as long as every alternative of the whole expression ends up with the
//            same length (then leftmost-first has nothing to choose: see xsg_classseq.h)
// Refused HERE (the caller then tries the automaton route, xsg_regex.cpp, which serves the variable-length operators):
//   * + ? {n,m} {n,} ^ $ (?flags) (?P<..>) \b \B \A \z \p \P \Q \C, backslash + letter/digit otherwise,
//   class members >= 0x80, alternatives of different lengths, more than kMaxClassSeq positions, more than
//   kMaxAlt alternatives (after merging those that differ in one position), an empty expression or alternative.
#include "xsg_classseq.h"
#include "xsg_rxlex.h"

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

struct Parser : AtomLexer {
  int depth = 0;

  // ---- recursive descent over  alt := concat ('|' concat)* ;  concat := item* ;  item := atom ['{' n '}'] -------
  using Seq = std::vector<ByteSet>;
  using SeqSet = std::vector<Seq>;  // alternatives (any lengths while parsing; checked at the end)
  static constexpr size_t kWorkAlts = 64;  // before merging

  bool cross(SeqSet* acc, const SeqSet& rhs) {
    SeqSet out;
    if (acc->size() * rhs.size() > kWorkAlts) return fail("too many alternatives");
    for (const Seq& a : *acc)
      for (const Seq& b : rhs) {
        if (a.size() + b.size() > kMaxClassSeq)
          return fail("expression longer than " + std::to_string(kMaxClassSeq) + " positions");
        Seq c = a;
        c.insert(c.end(), b.begin(), b.end());
        out.push_back(std::move(c));
      }
    acc->swap(out);
    return true;
  }

  bool parse_atom(SeqSet* out) {
    const uint8_t c = re[i];
    ByteSet s{};
    if (c == '(') {
      ++i;
      if (i < n && re[i] == '?') {
        if (i + 1 < n && re[i + 1] == ':') i += 2;  // (?: ... ) is as transparent as ( ... )
        else return fail("(?...) flags and named groups are not supported");
      }
      ++depth;
      if (!parse_alt(out)) return false;
      if (i >= n || re[i] != ')') return fail("missing )");
      ++i, --depth;
      return true;
    }
    if (c == '[') {
      ++i;
      if (!char_class(&s)) return false;
      out->assign(1, Seq(1, s));
      return true;
    }
    if (c == '\\') {
      ++i;
      if (!escape(&s)) return false;
      out->assign(1, Seq(1, s));
      return true;
    }
    if (c == '.') {  // any character but '\n' (RE2 without (?s)); ASCII only, see the header
      ++i;
      add_range(s, 0, 0x7f);
      s['\n' >> 5] &= ~(1u << ('\n' & 31));
      ascii_only = true;
      out->assign(1, Seq(1, s));
      return true;
    }
    if (c == '*' || c == '+' || c == '?' || c == '^' || c == '$')
      return fail(std::string("operator '") + (char)c + "' is not a fixed-length class sequence");
    if (c == '{') return fail("{ without an atom before it");
    if (c >= 0x80) {  // one well-formed UTF-8 sequence: its bytes, in order
      const int len = c >= 0xf0 ? 4 : c >= 0xe0 ? 3 : 2;
      if (c < 0xc2 || c > 0xf4 || i + len > n) return fail("pattern is not valid UTF-8");
      for (int k = 1; k < len; ++k)
        if ((re[i + k] & 0xc0) != 0x80) return fail("pattern is not valid UTF-8");
      Seq q;
      for (int k = 0; k < len; ++k) q.push_back(single(re[i + k]));
      i += len;
      out->assign(1, q);
      return true;
    }
    ++i;
    out->assign(1, Seq(1, single(c)));  // ']' and '}' on their own are literals in RE2 too
    return true;
  }

  bool parse_item(SeqSet* out) {
    if (!parse_atom(out)) return false;
    if (i < n && (re[i] == '*' || re[i] == '+' || re[i] == '?'))
      return fail(std::string("operator '") + (char)re[i] + "' is not a fixed-length class sequence");
    if (i < n && re[i] == '{') {
      size_t j = i + 1;
      uint32_t cnt = 0, digits = 0;
      while (j < n && re[j] >= '0' && re[j] <= '9' && digits < 4) cnt = cnt * 10 + (re[j++] - '0'), ++digits;
      if (digits == 0 || j >= n || re[j] != '}') return fail("only the fixed repetition {n} is supported");
      if (cnt == 0) return fail("{0} is not supported");
      i = j + 1;
      if (i < n && (re[i] == '?' || re[i] == '*' || re[i] == '+' || re[i] == '{')) return fail("stacked quantifiers");
      const SeqSet base = *out;
      for (uint32_t k = 1; k < cnt; ++k)
        if (!cross(out, base)) return false;
    }
    return true;
  }

  bool parse_concat(SeqSet* out) {
    out->assign(1, Seq());
    bool any = false;
    while (i < n && re[i] != '|' && re[i] != ')') {
      SeqSet item;
      if (!parse_item(&item)) return false;
      if (!cross(out, item)) return false;
      any = true;
    }
    if (!any) return fail("empty expression or alternative");
    return true;
  }

  bool parse_alt(SeqSet* out) {
    if (!parse_concat(out)) return false;
    while (i < n && re[i] == '|') {
      ++i;
      SeqSet rhs;
      if (!parse_concat(&rhs)) return false;
      out->insert(out->end(), rhs.begin(), rhs.end());
      if (out->size() > kWorkAlts) return fail("too many alternatives");
    }
    return true;
  }

  bool run(SeqSet* out) {
    if (n == 0) return fail("empty expression");
    if (!parse_alt(out)) return false;
    if (i < n) return fail(re[i] == ')' ? "unmatched )" : "trailing garbage");
    return true;
  }
};

bool subset_of(const ByteSet& a, const ByteSet& b) {
  for (int q = 0; q < 8; ++q)
    if (a[q] & ~b[q]) return false;
  return true;
}

// fewer alternatives, same language: drop an alternative another one covers; fuse two that differ in one position
void merge_alternatives(std::vector<std::vector<ByteSet>>* alts) {
  bool changed = true;
  while (changed) {
    changed = false;
    for (size_t a = 0; a < alts->size() && !changed; ++a)
      for (size_t b = 0; b < alts->size() && !changed; ++b) {
        if (a == b) continue;
        const auto &A = (*alts)[a], &B = (*alts)[b];
        size_t differ = 0, at = 0;
        bool a_in_b = true;
        for (size_t k = 0; k < A.size(); ++k) {
          if (A[k] != B[k]) ++differ, at = k;
          a_in_b &= subset_of(A[k], B[k]);
        }
        if (a_in_b) {  // (covers A == B)
          alts->erase(alts->begin() + (ptrdiff_t)a);
          changed = true;
        } else if (differ == 1) {
          for (int q = 0; q < 8; ++q) (*alts)[b][at][q] |= A[at][q];
          alts->erase(alts->begin() + (ptrdiff_t)a);
          changed = true;
        }
      }
  }
}

}  // namespace

bool class_expr_from_alternatives(std::vector<std::vector<ByteSet>> alts, ClassExpr* out) {
  *out = ClassExpr{};
  if (alts.empty() || alts[0].empty()) return false;
  const size_t len = alts[0].size();
  for (const auto& a : alts)
    if (a.size() != len) return false;
  merge_alternatives(&alts);
  if (len > kMaxClassSeq || alts.size() > kMaxAlt || alts.size() * len > kMaxAltSets) return false;
  out->npos = (uint32_t)len;
  out->alts = std::move(alts);
  return true;
}

bool compile_class_expr(const uint8_t* re, size_t n, bool ignore_case, ClassExpr* out, std::string* err) {
  *out = ClassExpr{};
  Parser p;
  p.re = re, p.n = n, p.err = err;
  p.icase = ignore_case;
  Parser::SeqSet alts;
  if (!p.run(&alts)) return false;
  const size_t len = alts[0].size();
  for (const auto& a : alts)
    if (a.size() != len) {
      *err = "the alternatives have different lengths (" + std::to_string(len) + " and " + std::to_string(a.size()) +
             " positions): not a fixed-length expression";
      return false;
    }
  merge_alternatives(&alts);
  if (alts.size() > kMaxAlt || alts.size() * len > kMaxAltSets) {
    *err = "too many alternatives for the GPU matcher (" + std::to_string(alts.size()) + " x " + std::to_string(len) +
           " positions; at most " + std::to_string(kMaxAlt) + " alternatives and " + std::to_string(kMaxAltSets) +
           " sets)";
    return false;
  }
  if (ignore_case)  // as for literals: the kernel lowers the data bytes, so every set holds the lowered members
    for (auto& a : alts) fold_sets(&a);
  merge_alternatives(&alts);  // folding may have made alternatives equal
  out->npos = (uint32_t)len;
  out->alts = std::move(alts);
  out->ascii_only = p.ascii_only;
  return true;
}

std::vector<ByteSet> union_sets(const ClassExpr& e) {
  std::vector<ByteSet> u(e.npos, ByteSet{});
  for (const auto& a : e.alts)
    for (uint32_t k = 0; k < e.npos; ++k)
      for (int q = 0; q < 8; ++q) u[k][q] |= a[k][q];
  return u;
}

// the round-1 entry point: one sequence, byte-exact on any data
bool compile_class_sequence(const uint8_t* re, size_t n, std::vector<ByteSet>* seq, std::string* err) {
  seq->clear();
  ClassExpr e;
  if (!compile_class_expr(re, n, false, &e, err)) return false;
  if (e.alts.size() != 1 || e.ascii_only) {
    *err = "not a single class sequence";
    return false;
  }
  *seq = e.alts[0];
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
