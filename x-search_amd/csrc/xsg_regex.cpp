// xsg_regex.cpp -- RE2-syntax expression -> priority NFA -> the two byte-class DFAs of xsg_regex.h.
// Host code, no device needed; tests/test_regex_dfa.py drives the tables from Python against CPython's `re`.
#include "xsg_regex.h"

#include <string.h>

#include <algorithm>
#include <map>

#include "xsg_rxlex.h"

namespace xsg {
namespace {

constexpr uint32_t kInf = 0xffffffffu;

struct Node {
  enum Kind { kSet, kCat, kAlt, kRep } kind = kSet;
  ByteSet set{};
  std::vector<int> kids;
  uint32_t lo = 1, hi = 1;  // kRep; hi == kInf: unbounded
  bool lazy = false;
  uint64_t minlen = 0;
};

// ---- syntax -> tree -------------------------------------------------------------------------------------------
// alt := concat ('|' concat)* ; concat := item+ ; item := atom [ ('*' | '+' | '?' | '{' n [',' [m]] '}') ['?'] ]
struct TreeParser : AtomLexer {
  std::vector<Node> pool;

  int make(Node::Kind k) {
    pool.emplace_back();
    pool.back().kind = k;
    return (int)pool.size() - 1;
  }
  int make_set(const ByteSet& s) {
    const int id = make(Node::kSet);
    pool[id].set = s;
    pool[id].minlen = 1;
    return id;
  }

  bool parse_atom(int* out) {
    const uint8_t c = re[i];
    ByteSet s{};
    if (c == '(') {
      ++i;
      if (i < n && re[i] == '?') {
        if (i + 1 < n && re[i + 1] == ':') i += 2;  // (?: ... ) is as transparent as ( ... ): the walk reads the whole match
        else return fail("(?...) flags and named groups are not supported");
      }
      if (!parse_alt(out)) return false;
      if (i >= n || re[i] != ')') return fail("missing )");
      ++i;
      return true;
    }
    if (c == '[') {
      ++i;
      if (!char_class(&s)) return false;
      *out = make_set(s);
      return true;
    }
    if (c == '\\') {
      ++i;
      if (!escape(&s)) return false;
      *out = make_set(s);
      return true;
    }
    if (c == '.') {  // any character but '\n' (RE2 without (?s)); ASCII only, see xsg_classseq.h
      ++i;
      add_range(s, 0, 0x7f);
      s['\n' >> 5] &= ~(1u << ('\n' & 31));
      ascii_only = true;
      *out = make_set(s);
      return true;
    }
    if (c == '*' || c == '+' || c == '?') return fail(std::string("operator '") + (char)c + "' without an atom before it");
    if (c == '^' || c == '$') return fail(std::string("anchor '") + (char)c + "' is not supported");
    if (c == '{') return fail("{ without an atom before it");
    if (c >= 0x80) {  // one well-formed UTF-8 sequence: its bytes, in order
      const int len = c >= 0xf0 ? 4 : c >= 0xe0 ? 3 : 2;
      if (c < 0xc2 || c > 0xf4 || i + len > n) return fail("pattern is not valid UTF-8");
      for (int k = 1; k < len; ++k)
        if ((re[i + k] & 0xc0) != 0x80) return fail("pattern is not valid UTF-8");
      const int cat = make(Node::kCat);
      for (int k = 0; k < len; ++k) {
        const int b = make_set(single(re[i + k]));
        pool[cat].kids.push_back(b);
      }
      pool[cat].minlen = (uint64_t)len;
      i += len;
      *out = cat;
      return true;
    }
    ++i;
    *out = make_set(single(c));  // ']' and '}' on their own are literals in RE2 too
    return true;
  }

  bool number(size_t* j, uint32_t* v) {
    uint32_t digits = 0;
    *v = 0;
    while (*j < n && re[*j] >= '0' && re[*j] <= '9' && digits < 5) *v = *v * 10 + (re[(*j)++] - '0'), ++digits;
    return digits > 0 && digits < 5;
  }

  bool parse_item(int* out) {
    int atom = -1;
    if (!parse_atom(&atom)) return false;
    uint32_t lo = 1, hi = 1;
    bool quantified = false;
    if (i < n && (re[i] == '*' || re[i] == '+' || re[i] == '?')) {
      lo = re[i] == '+' ? 1u : 0u;
      hi = re[i] == '?' ? 1u : kInf;
      quantified = true;
      ++i;
    } else if (i < n && re[i] == '{') {
      size_t j = i + 1;
      if (!number(&j, &lo) || j >= n) return fail("malformed repetition {n} / {n,} / {n,m}");
      hi = lo;
      if (re[j] == ',') {
        ++j;
        if (j < n && re[j] == '}') hi = kInf;
        else if (!number(&j, &hi)) return fail("malformed repetition {n} / {n,} / {n,m}");
      }
      if (j >= n || re[j] != '}') return fail("malformed repetition {n} / {n,} / {n,m}");
      if (hi != kInf && hi < lo) return fail("bad repetition count");
      if (hi == 0) return fail("{0} is not supported");
      if (lo > 1000 || (hi != kInf && hi > 1000)) return fail("repetition count above 1000 (RE2's limit)");
      i = j + 1;
      quantified = true;
    }
    if (!quantified) {
      *out = atom;
      return true;
    }
    bool lazy = false;
    if (i < n && re[i] == '?') lazy = true, ++i;
    if (i < n && (re[i] == '?' || re[i] == '*' || re[i] == '+' || re[i] == '{')) return fail("stacked quantifiers");
    if (pool[atom].minlen == 0)
      return fail("repetition of a sub-expression that can match the empty string is not supported");
    if (lo == 1 && hi == 1) {
      *out = atom;
      return true;
    }
    const int r = make(Node::kRep);
    pool[r].kids.push_back(atom);
    pool[r].lo = lo, pool[r].hi = hi, pool[r].lazy = lazy;
    pool[r].minlen = pool[atom].minlen * lo;
    *out = r;
    return true;
  }

  bool parse_concat(int* out) {
    std::vector<int> items;
    while (i < n && re[i] != '|' && re[i] != ')') {
      int it = -1;
      if (!parse_item(&it)) return false;
      items.push_back(it);
    }
    if (items.empty()) return fail("empty expression or alternative");
    if (items.size() == 1) {
      *out = items[0];
      return true;
    }
    const int cat = make(Node::kCat);
    uint64_t m = 0;
    for (int it : items) m += pool[it].minlen;
    pool[cat].kids = items;
    pool[cat].minlen = m;
    *out = cat;
    return true;
  }

  bool parse_alt(int* out) {
    std::vector<int> alts;
    int first = -1;
    if (!parse_concat(&first)) return false;
    alts.push_back(first);
    while (i < n && re[i] == '|') {
      ++i;
      int nxt = -1;
      if (!parse_concat(&nxt)) return false;
      alts.push_back(nxt);
    }
    if (alts.size() == 1) {
      *out = alts[0];
      return true;
    }
    const int alt = make(Node::kAlt);
    uint64_t m = UINT64_MAX;
    for (int a : alts) m = std::min(m, pool[a].minlen);
    pool[alt].kids = alts;
    pool[alt].minlen = m;
    *out = alt;
    return true;
  }

  bool run(int* root) {
    if (n == 0) return fail("empty expression");
    if (!parse_alt(root)) return false;
    if (i < n) return fail(re[i] == ')' ? "unmatched )" : "trailing garbage");
    return true;
  }
};

// ---- tree -> NFA ------------------------------------------------------------------------------------------------
struct Inst {
  enum Kind { kSet, kSplit, kMatch } kind = kMatch;
  int set = -1;  // index into Nfa::sets
  int out = -1;  // kSet: next; kSplit: the PREFERRED branch
  int out1 = -1;
};

struct Nfa {
  std::vector<Inst> prog;
  std::vector<ByteSet> sets;  // distinct
  bool too_big = false;

  int set_id(const ByteSet& s) {
    for (size_t k = 0; k < sets.size(); ++k)
      if (sets[k] == s) return (int)k;
    sets.push_back(s);
    return (int)sets.size() - 1;
  }
  int add(Inst::Kind k, int set, int out, int out1) {
    if (prog.size() >= kRxMaxNfa) {
      too_big = true;
      return 0;
    }
    Inst in;
    in.kind = k, in.set = set, in.out = out, in.out1 = out1;
    prog.push_back(in);
    return (int)prog.size() - 1;
  }
};

// entry pc of `node` followed by the continuation `next`; mirror = build the reversed expression
int emit(const std::vector<Node>& pool, int node, int next, bool mirror, Nfa* nfa) {
  if (nfa->too_big) return 0;
  const Node& nd = pool[node];
  switch (nd.kind) {
    case Node::kSet:
      return nfa->add(Inst::kSet, nfa->set_id(nd.set), next, -1);
    case Node::kCat: {
      // built back to front: the last factor (in matching order) first
      if (!mirror)
        for (size_t k = nd.kids.size(); k-- > 0;) next = emit(pool, nd.kids[k], next, mirror, nfa);
      else
        for (size_t k = 0; k < nd.kids.size(); ++k) next = emit(pool, nd.kids[k], next, mirror, nfa);
      return next;
    }
    case Node::kAlt: {
      std::vector<int> entry;
      for (int kid : nd.kids) entry.push_back(emit(pool, kid, next, mirror, nfa));
      int s = entry.back();
      for (size_t k = entry.size() - 1; k-- > 0;) s = nfa->add(Inst::kSplit, -1, entry[k], s);  // earlier alternative preferred
      return s;
    }
    case Node::kRep: {
      const int kid = nd.kids[0];
      int entry;
      if (nd.hi == kInf) {
        const int loop = nfa->add(Inst::kSplit, -1, -1, -1);
        const int body = emit(pool, kid, loop, mirror, nfa);
        if (nfa->too_big) return 0;
        nfa->prog[loop].out = nd.lazy ? next : body;
        nfa->prog[loop].out1 = nd.lazy ? body : next;
        entry = nd.lo == 0 ? loop : body;  // x{lo,} = x^(lo-1) x+
        for (uint32_t k = 1; k < nd.lo; ++k) entry = emit(pool, kid, entry, mirror, nfa);
      } else {
        entry = next;  // x{lo,hi} = x^lo (x (x ...)?)?  -- hi - lo nested optionals, innermost first
        for (uint32_t k = nd.lo; k < nd.hi; ++k) {
          const int body = emit(pool, kid, entry, mirror, nfa);
          entry = nfa->add(Inst::kSplit, -1, nd.lazy ? next : body, nd.lazy ? body : next);
        }
        for (uint32_t k = 0; k < nd.lo; ++k) entry = emit(pool, kid, entry, mirror, nfa);
      }
      return entry;
    }
  }
  return 0;
}

// ---- NFA -> DFA -------------------------------------------------------------------------------------------------
struct Closure {
  const Nfa& nfa;
  std::vector<uint32_t> mark;  // generation stamps
  uint32_t gen = 0;
  explicit Closure(const Nfa& n) : nfa(n), mark(n.prog.size(), 0) {}
  void begin() { ++gen; }
  // follows splits in priority order; consuming instructions and the match land in `list` in that order, once
  void add(int pc, std::vector<int>* list) {
    // explicit stack: expressions like (a|b|c|...){1000} nest deeply
    std::vector<int> stack{pc};
    while (!stack.empty()) {
      const int p = stack.back();
      stack.pop_back();
      if (mark[p] == gen) continue;
      mark[p] = gen;
      const Inst& in = nfa.prog[p];
      if (in.kind == Inst::kSplit) {
        stack.push_back(in.out1);  // popped second
        stack.push_back(in.out);
      } else {
        list->push_back(p);
      }
    }
  }
};

struct Dfa {
  std::vector<std::vector<uint32_t>> rows;  // [state][class] -> state; state 0 = dead
  std::vector<bool> accepting;
  uint32_t start = 0;
};

// ordered = leftmost-first (priority lists, cut behind the first match); else longest match (plain subset construction)
bool determinise(const Nfa& nfa, int entry, const std::vector<uint8_t>& rep, bool ordered, Dfa* out, std::string* err) {
  const uint32_t ncls = (uint32_t)rep.size();
  std::map<std::vector<int>, uint32_t> ids;
  std::vector<std::vector<int>> states;
  Closure cl(nfa);
  auto canon = [&](std::vector<int>* list) {
    if (ordered) {
      for (size_t k = 0; k < list->size(); ++k)
        if (nfa.prog[(*list)[k]].kind == Inst::kMatch) {
          list->resize(k + 1);  // whatever ranks behind a match can no longer win
          break;
        }
    } else {
      std::sort(list->begin(), list->end());
    }
  };
  auto intern = [&](std::vector<int>&& list, uint32_t* id) {
    auto it = ids.find(list);
    if (it != ids.end()) {
      *id = it->second;
      return true;
    }
    if (states.size() >= kRxMaxStates) {
      *err = "the expression needs an automaton of more than " + std::to_string(kRxMaxStates) + " states";
      return false;
    }
    *id = (uint32_t)states.size();
    ids.emplace(list, *id);
    states.push_back(std::move(list));
    return true;
  };
  uint32_t id = 0;
  intern({}, &id);  // dead = 0
  std::vector<int> first;
  cl.begin();
  cl.add(entry, &first);
  canon(&first);
  if (!intern(std::move(first), &out->start)) return false;
  out->rows.clear();
  for (uint32_t s = 0; s < states.size(); ++s) {
    out->rows.emplace_back(ncls, 0u);
    if (s == 0) continue;
    for (uint32_t c = 0; c < ncls; ++c) {
      const uint32_t b = rep[c];
      std::vector<int> next;
      cl.begin();
      const std::vector<int> cur = states[s];  // copy: `states` may grow below
      for (int pc : cur) {
        const Inst& in = nfa.prog[pc];
        if (in.kind == Inst::kSet && set_has(nfa.sets[in.set], b)) cl.add(in.out, &next);
      }
      canon(&next);
      uint32_t t = 0;
      if (!intern(std::move(next), &t)) return false;
      out->rows[s][c] = t;
    }
  }
  out->accepting.assign(states.size(), false);
  for (uint32_t s = 0; s < states.size(); ++s)
    for (int pc : states[s]) out->accepting[s] = out->accepting[s] || nfa.prog[pc].kind == Inst::kMatch;
  return true;
}

// renumber (dead, plain states, accepting states) and flatten to pre-multiplied uint16 rows
bool flatten(const Dfa& d, uint32_t ncls, std::vector<uint16_t>* table, uint32_t* nstates, uint32_t* start,
             uint32_t* first_acc, std::string* err) {
  const uint32_t n = (uint32_t)d.rows.size();
  if ((uint64_t)n * ncls > kRxMaxEntries) {
    *err = "the expression needs an automaton of " + std::to_string(n) + " states x " + std::to_string(ncls) +
           " byte classes; at most " + std::to_string(kRxMaxEntries) + " table entries fit the kernel's LDS";
    return false;
  }
  std::vector<uint32_t> to(n, 0);
  uint32_t k = 1;
  for (uint32_t s = 1; s < n; ++s)
    if (!d.accepting[s]) to[s] = k++;
  *first_acc = k;
  for (uint32_t s = 1; s < n; ++s)
    if (d.accepting[s]) to[s] = k++;
  table->assign((size_t)n * ncls, 0);
  for (uint32_t s = 0; s < n; ++s)
    for (uint32_t c = 0; c < ncls; ++c) (*table)[(size_t)to[s] * ncls + c] = (uint16_t)(to[d.rows[s][c]] * ncls);
  *nstates = n;
  *start = to[d.start];
  return true;
}

// The first k bytes of every match, as alternatives of class sequences: the paths of length k through the anchored
// automaton, the edges out of a state grouped by their target (k <= the shortest match, so no path ends early).
// Among k = 3 .. kmax the one whose alternatives look rarest in text wins (a position that accepts s byte values is
// priced at s/64; a longer prefix must at least halve the estimate to be worth its extra alternatives: `Sher.*mes`
// keeps `Sher`, `colou?r` takes `colo[ur]` + ...).  Gives up (npos = 0) if the alternatives do not fit the scan
// kernel's matcher or do not look selective: every alternative must have at least three positions that accept at
// most two byte values (`Sher`, `[Ss][Hh][Ee]`; not `\\w\\w\\w`), or a hot filter on them would fire everywhere.
void find_prefix(const Dfa& ad, const std::vector<uint8_t>& rep, const uint8_t (&class_of)[256], uint32_t kmax, ClassExpr* out) {
  *out = ClassExpr{};
  const uint32_t ncls = (uint32_t)rep.size();
  std::vector<ByteSet> class_bytes(ncls, ByteSet{});
  for (uint32_t b = 0; b < 256; ++b) set_add(class_bytes[class_of[b]], b);
  double best = 1e300;
  for (uint32_t k = 3; k <= kmax; ++k) {
    struct Path {
      uint32_t state;
      std::vector<ByteSet> seq;
    };
    std::vector<Path> paths{{ad.start, {}}};
    bool ok = true;
    for (uint32_t depth = 0; depth < k && ok; ++depth) {
      std::vector<Path> next;
      for (const Path& pa : paths) {
        std::map<uint32_t, ByteSet> by_target;
        for (uint32_t c = 0; c < ncls; ++c) {
          const uint32_t t = ad.rows[pa.state][c];
          if (t == 0) continue;
          ByteSet& bs = by_target[t];
          for (int q = 0; q < 8; ++q) bs[q] |= class_bytes[c][q];
        }
        for (const auto& kv : by_target) {
          Path np{kv.first, pa.seq};
          np.seq.push_back(kv.second);
          next.push_back(std::move(np));
        }
        if (next.size() > 64) {
          ok = false;
          break;
        }
      }
      paths.swap(next);
    }
    if (!ok || paths.empty()) break;  // longer prefixes only have more paths
    std::vector<std::vector<ByteSet>> alts;
    for (const Path& pa : paths) alts.push_back(pa.seq);
    ClassExpr ex;
    if (!class_expr_from_alternatives(std::move(alts), &ex)) continue;
    bool selective = true;
    double rate = 0;
    for (const auto& a : ex.alts) {
      uint32_t narrow = 0;
      double pr = 1;
      for (const ByteSet& st : a) {
        narrow += set_size(st) <= 2;
        pr *= std::min(1.0, set_size(st) / 64.0);
      }
      selective = selective && narrow >= 3;
      rate += pr;
    }
    if (!selective || rate > 0.5 * best) continue;
    best = rate;
    *out = std::move(ex);
  }
}

// ---- a factor every match contains ------------------------------------------------------------------------------
// The class sequence a node matches if it matches exactly one (a set, a concatenation of such, x{n} of such)
bool fixed_sequence(const std::vector<Node>& pool, int node, std::vector<ByteSet>* out) {
  const Node& nd = pool[node];
  switch (nd.kind) {
    case Node::kSet:
      out->push_back(nd.set);
      return out->size() <= 64;
    case Node::kCat:
      for (int kid : nd.kids)
        if (!fixed_sequence(pool, kid, out)) return false;
      return true;
    case Node::kRep: {
      if (nd.lo != nd.hi || nd.lo > 16) return false;
      for (uint32_t k = 0; k < nd.lo; ++k)
        if (!fixed_sequence(pool, nd.kids[0], out)) return false;
      return true;
    }
    default:
      return false;
  }
}

// Runs of positions that are adjacent in every match: `cur` grows along a concatenation and is closed (`done`) wherever
// the text between two pieces can vary.  x{lo,hi} of a fixed x with hi > lo: the first lo copies are adjacent to what
// precedes, the last lo copies to what follows.
void collect_runs(const std::vector<Node>& pool, int node, std::vector<ByteSet>* cur, std::vector<std::vector<ByteSet>>* done) {
  auto flush = [&] {
    if (!cur->empty()) done->push_back(*cur);
    cur->clear();
  };
  const Node& nd = pool[node];
  std::vector<ByteSet> fx;
  if (fixed_sequence(pool, node, &fx)) {
    cur->insert(cur->end(), fx.begin(), fx.end());
    return;
  }
  switch (nd.kind) {
    case Node::kCat:
      for (int kid : nd.kids) collect_runs(pool, kid, cur, done);
      return;
    case Node::kRep: {
      std::vector<ByteSet> one;
      const bool fixed = fixed_sequence(pool, nd.kids[0], &one) && one.size() <= 8;
      if (fixed && nd.lo >= 1) {
        const uint32_t copies = std::min<uint32_t>(nd.lo, 4);
        for (uint32_t k = 0; k < copies; ++k) cur->insert(cur->end(), one.begin(), one.end());
        flush();
        for (uint32_t k = 0; k < copies; ++k) cur->insert(cur->end(), one.begin(), one.end());
      } else if (nd.lo >= 1) {  // at least one copy of something variable: what it must contain, on its own
        flush();
        collect_runs(pool, nd.kids[0], cur, done);
        flush();
      } else {
        flush();
      }
      return;
    }
    default:  // an alternation: nothing is common to all alternatives for sure
      flush();
      return;
  }
}

void find_factor(const std::vector<Node>& pool, int root, ClassExpr* out) {
  *out = ClassExpr{};
  std::vector<ByteSet> cur;
  std::vector<std::vector<ByteSet>> runs;
  collect_runs(pool, root, &cur, &runs);
  if (!cur.empty()) runs.push_back(cur);
  size_t best_narrow = 0;
  const std::vector<ByteSet>* best = nullptr;
  for (const auto& r : runs) {
    size_t narrow = 0;
    for (const ByteSet& st : r) narrow += set_size(st) <= 2;
    if (narrow > best_narrow) best_narrow = narrow, best = &r;
  }
  if (!best || best_narrow < 3) return;
  std::vector<ByteSet> seq(best->begin(), best->begin() + (ptrdiff_t)std::min<size_t>(best->size(), kMaxClassSeq));
  size_t narrow = 0;
  for (const ByteSet& st : seq) narrow += set_size(st) <= 2;
  if (narrow < 3) return;
  std::vector<std::vector<ByteSet>> alts{seq};
  if (!class_expr_from_alternatives(std::move(alts), out)) *out = ClassExpr{};
}

}  // namespace

bool compile_regex_dfa(const uint8_t* re, size_t n, bool ignore_case, RegexDfa* out, std::string* err) {
  *out = RegexDfa{};
  TreeParser p;
  p.re = re, p.n = n, p.err = err;
  p.icase = ignore_case;
  int root = -1;
  if (!p.run(&root)) return false;
  if (p.pool[root].minlen == 0) {
    *err = "the expression can match the empty string (the reference's walk would not advance): not supported";
    return false;
  }
  if (ignore_case)  // the data is not folded on this route: every set accepts both cases of its letters
    for (Node& nd : p.pool)
      if (nd.kind == Node::kSet)
        for (uint32_t b = 'a'; b <= 'z'; ++b)
          if (set_has(nd.set, b) || set_has(nd.set, b - 32)) set_add(nd.set, b), set_add(nd.set, b - 32);
  // A set that accepts '\n' (\s, a negated class) lets a match span lines: then the unit of the walk is the chunk,
  // not the line (k_rx_chunk), and the line tags do not apply -- as for a literal that contains '\n'.
  for (const Node& nd : p.pool)
    if (nd.kind == Node::kSet && set_has(nd.set, '\n')) out->multiline = true;

  // forward: any-byte loop of lowest priority in front of the expression (RE2's unanchored search)
  Nfa f;
  const int fmatch = f.add(Inst::kMatch, -1, -1, -1);
  const int fentry = emit(p.pool, root, fmatch, false, &f);
  ByteSet any{};
  for (uint32_t b = 0; b < 256; ++b)
    if (b != '\n' || out->multiline) set_add(any, b);
  const int floop_set = f.add(Inst::kSet, f.set_id(any), -1, -1);
  const int fstart = f.add(Inst::kSplit, -1, fentry, floop_set);
  Nfa r;
  const int rmatch = r.add(Inst::kMatch, -1, -1, -1);
  const int rentry = emit(p.pool, root, rmatch, true, &r);
  if (f.too_big || r.too_big) {
    *err = "the expression expands to more than " + std::to_string(kRxMaxNfa) + " NFA positions";
    return false;
  }
  f.prog[floop_set].out = fstart;

  // byte classes: bytes that every set of the expression treats alike ('\n' is in none: a class of its own)
  std::map<std::vector<bool>, uint32_t> sig_ids;
  std::vector<uint8_t> rep;
  for (uint32_t b = 0; b < 256; ++b) {
    std::vector<bool> sig;
    for (const ByteSet& s : f.sets) sig.push_back(set_has(s, b));
    sig.push_back(b == '\n');
    auto it = sig_ids.find(sig);
    if (it == sig_ids.end()) {
      it = sig_ids.emplace(sig, (uint32_t)rep.size()).first;
      rep.push_back((uint8_t)b);
    }
    out->class_of[b] = (uint8_t)it->second;
  }
  out->ncls = (uint32_t)rep.size();

  Dfa fd, rd, ad;
  if (!determinise(f, fstart, rep, true, &fd, err)) return false;
  if (!determinise(r, rentry, rep, false, &rd, err)) return false;
  if (!determinise(f, fentry, rep, true, &ad, err)) return false;  // anchored: the expression without the loop in front
  if (!flatten(fd, out->ncls, &out->fwd, &out->fwd_states, &out->fwd_start, &out->fwd_first_acc, err)) return false;
  if (!flatten(rd, out->ncls, &out->rev, &out->rev_states, &out->rev_start, &out->rev_first_acc, err)) return false;
  if (!flatten(ad, out->ncls, &out->anc, &out->anc_states, &out->anc_start, &out->anc_first_acc, err)) return false;
  find_prefix(ad, rep, out->class_of, (uint32_t)std::min<uint64_t>(p.pool[root].minlen, 8), &out->prefix);
  if (out->prefix.npos == 0 && !out->multiline) find_factor(p.pool, root, &out->factor);
  out->minlen = (uint32_t)std::min<uint64_t>(p.pool[root].minlen, 0xffffffffu);
  out->ascii_only = p.ascii_only;
  return true;
}

}  // namespace xsg
