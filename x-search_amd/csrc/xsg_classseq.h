// xsg_classseq.h -- the regular expressions the scan kernel can decide itself.
//
// The reference sends a pattern to RE2 when it "does not match itself as a
// regex" (include/xsearch/utils/utils.h:17-25) and then walks the chunk with
// RE2::PartialMatch (include/xsearch/string_search/search_wrappers.h:63-87,
// 209-271).  The only regexes its tests use are fixed-length sequences of byte
// classes -- `She[r ]lock` (test/src/xsearchTest.cpp:9), `(a[n|m]t)`
// (test/src/string_search/search_wrappersTest.cpp:78).  For exactly that family
// a match is a position where every class accepts its byte: the same
// position-wise decision as a literal, so it runs in k_scan.  Expressions of
// variable length (repetition operators, alternatives of different lengths) are
// not this family: xsg_regex.h compiles them to automata for a second kernel.
// Anchors and flags are refused by both, never approximated.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <array>
#include <string>
#include <vector>

namespace xsg {

constexpr uint32_t kMaxClassSeq = 32;  // positions; 32 x 256 bits = 1 KiB, the LDS the long-pattern path already has

using ByteSet = std::array<uint32_t, 8>;  // bit b set <=> byte b is accepted

inline bool set_has(const ByteSet& s, uint32_t b) { return (s[b >> 5] >> (b & 31u)) & 1u; }
inline void set_add(ByteSet& s, uint32_t b) { s[b >> 5] |= 1u << (b & 31u); }
uint32_t set_size(const ByteSet& s);
// the only member of a singleton set, else -1
int set_single(const ByteSet& s);

// Parses `re` (RE2 syntax subset, see the .cpp) into one byte set per position.
// Returns false with a message in `err` if the expression is not a fixed-length
// class sequence of 1..kMaxClassSeq positions.
bool compile_class_sequence(const uint8_t* re, size_t n, std::vector<ByteSet>* seq, std::string* err);

// The wider family: an ALTERNATION of class sequences of one common length ( `|` at any depth, groups, atom{n},
// group{n} ), still decided position by position: the expression matches at offset o iff one alternative accepts
// every byte of data[o, o + npos).  All alternatives having the same length, RE2's leftmost-first rule has nothing
// to choose between them: the match is [o, o + npos) for the smallest such o.
// `ascii_only`: the expression used '.', a negated class or \D \W \S.  Those match whole code points in RE2; their
// sets here hold the ASCII bytes only, which is the same thing on ASCII data and nothing else -- a search with
// such an expression REFUSES data that holds a byte >= 0x80 (k_scan checks while it scans) instead of deciding it
// differently from RE2.
constexpr uint32_t kMaxAlt = 8;       // alternatives after merging
constexpr uint32_t kMaxAltSets = 64;  // alternatives x positions: 64 x 32 bytes = 2 KiB of LDS
struct ClassExpr {
  uint32_t npos = 0;
  std::vector<std::vector<ByteSet>> alts;  // each npos long
  bool ascii_only = false;
};
// ignore_case: the sets come back folded (every 'A'..'Z' member replaced by its lower-case letter: the kernel lowers
// the data), negated classes having been closed under case BEFORE the complement, as RE2's (?i) does.
bool compile_class_expr(const uint8_t* re, size_t n, bool ignore_case, ClassExpr* out, std::string* err);
// the same structure from alternatives given as sets (all of one length): merged, bounded by kMaxAlt / kMaxAltSets;
// false if they do not fit.  (The automaton route builds its prefilter this way, xsg_regex.cpp.)
bool class_expr_from_alternatives(std::vector<std::vector<ByteSet>> alts, ClassExpr* out);
// position-wise union of the alternatives (what a filter or a '\n' / overlap test may look at: a superset)
std::vector<ByteSet> union_sets(const ClassExpr& e);

// ignore_case as for literals (toLower on data and pattern, src/utils/string_utils.cpp:11-33):
// the kernel folds the data bytes, so every set is replaced by the fold of its members.
void fold_sets(std::vector<ByteSet>* seq);

// true if two occurrences can overlap: some shift 0 < s < n with seq[k] and seq[k+s] intersecting for all k
bool sequence_can_overlap(const std::vector<ByteSet>& seq);

}  // namespace xsg
