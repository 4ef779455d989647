// xsg_regex.h -- regular expressions of VARIABLE length, compiled to a pair of byte-class DFAs for k_rx_scan.
//
// The reference hands a pattern that "does not match itself as a regex" (include/xsearch/utils/utils.h:17-25) to
// RE2 and walks the chunk with RE2::PartialMatch(input, pattern, &match)
// (include/xsearch/string_search/search_wrappers.h:63-87, 209-271): the match is the LEFTMOST one and, among the
// matches that start there, the one a backtracking matcher would find FIRST (greedy operators prefer more, lazy
// ones less, `a|b` prefers a).  xsg_classseq.h serves the expressions whose matches all have one length in k_scan
// itself; this file serves the rest of the syntax that has no zero-width operators:
//
//   x* x+ x? x{n} x{n,} x{n,m} and their lazy forms x*? x+? x?? x{n,m}?, on atoms and groups
//   `a|b` with alternatives of any lengths, ( ) (?: ), and every atom of xsg_classseq.cpp
//   (literals, escapes, [classes], [:posix:], \d \w \s, and with their ASCII meaning . [^..] \D \W \S)
//
// How RE2 itself finds such a match (re2/dfa.cc, re2/re2.cc -- un-vendored submodule) is what is
// restated here: (1) a FORWARD automaton over the expression with a lowest-priority `any byte` loop in front, its
// states being PRIORITY-ORDERED lists of NFA positions, cut behind the first matching one ("leftmost-first"): the
// last input position at which a state holds a match before the automaton dies is the END of the match;
// (2) a REVERSE automaton over the mirrored expression, anchored at that end, longest match: the START.
// Both are determinised on the host over byte classes (bytes no set of the expression tells apart share a class)
// and shipped as tables of pre-multiplied 16-bit row offsets: next = table[state_row + class_of[byte]].
//
// Restrictions, all refused at compile time, never approximated:
//   * the expression, and every sub-expression under a repetition operator, must not match the empty string (the
//     reference's walk would not advance on an empty match, search_wrappers.h:75-76);
//   * (not a refusal, a slower route) if no set contains '\n', no match spans two lines, and lines are what the
//     kernel hands to its lanes (a line is walked sequentially, as the reference walks a chunk; lines are
//     independent).  An expression with a set that accepts '\n' (\s+, [^,]*) is `multiline`: its unit of sequential
//     work is the chunk (k_rx_chunk: one lane per chunk), and only the match tags apply to it;
//   * ^ $ \b \B \A \z (they read the context of a re-sliced input in the reference), (?flags), \p, \C, back-refs;
//   * automata over kRxMaxEntries table entries.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "xsg_classseq.h"

namespace xsg {

constexpr uint32_t kRxMaxStates = 4096;    // per automaton, before the table bound below
constexpr uint32_t kRxMaxEntries = 16384;  // states x classes per automaton: 32 KiB of LDS as uint16
constexpr uint32_t kRxMaxNfa = 4096;       // NFA positions (x{n,m} is expanded)

struct RegexDfa {
  uint32_t ncls = 0;     // byte classes; class_of['\n'] is a class of its own (unless multiline: leading to the dead state everywhere)
  uint8_t class_of[256];
  // Row-major transition tables, entries PRE-MULTIPLIED: entry = next_state * ncls.  State 0 is dead (its row is
  // all 0); states [1, first_acc) hold no match, states [first_acc, nstates) do.
  std::vector<uint16_t> fwd, rev;
  uint32_t fwd_states = 0, fwd_start = 0, fwd_first_acc = 0;  // start / first_acc as STATE numbers
  uint32_t rev_states = 0, rev_start = 0, rev_first_acc = 0;
  uint32_t minlen = 0;       // shortest match, >= 1
  // The ANCHORED forward automaton (no any-byte loop in front): from a position at which a match may start, it dies
  // at once or runs to the end of the leftmost-first match that starts exactly there.  Same classes, same encoding.
  std::vector<uint16_t> anc;
  uint32_t anc_states = 0, anc_start = 0, anc_first_acc = 0;
  // PREFILTER: every match starts with `prefix.npos` bytes that one of these class sequences accepts (read off the
  // anchored automaton: its paths of that length, edges grouped by target state).  Empty (npos == 0) if the
  // expression has no selective start (`\\w+ing`) or too many.  When present, the synchronous entry points find the
  // candidate positions with the scan kernel's class-sequence matcher at streaming speed and run the anchored
  // automaton at candidates only (csrc/xsg_rx_kernels.hip: k_rx_verify).
  ClassExpr prefix;
  // FACTOR: a class sequence that every match CONTAINS somewhere (`\\w+ing` -> `\\wing`), for expressions without a
  // selective start (prefix.npos == 0) that cannot match across lines.  A line without the factor has no match, so
  // k_rx_scan only needs the tiles in which a line with a factor occurrence starts (csrc/xsg_api.cpp:
  // ensure_factor_mask).  npos == 0: none found.
  ClassExpr factor;
  bool ascii_only = false;   // as ClassExpr::ascii_only: a search refuses data with a byte >= 0x80
  bool multiline = false;    // some set accepts '\n': matches may span lines ('\n' is then an ordinary byte for the automata)
};

// ignore_case: every set is closed under ASCII case (the data is NOT folded on this route).
bool compile_regex_dfa(const uint8_t* re, size_t n, bool ignore_case, RegexDfa* out, std::string* err);

}  // namespace xsg
