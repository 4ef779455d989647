// xsg_linesum.h -- line summaries for XSG_COUNT_LINES (host/device).
//
// search::count(data, pattern, skip_to_nl=true)
// (include/xsearch/string_search/search_wrappers.h:163-185) counts the lines
// that contain at least one occurrence.  A byte range is summarised by
//   nl : it contains a '\n'
//   F  : the open segment before its first '\n' (the whole range if none) holds a match start
//   Lh : the open segment after its last '\n' (the whole range if none) holds a match start
//   C  : number of segments closed by '\n' on both sides inside the range that hold a match start
// Summaries of adjacent ranges combine associatively, so lanes, waves, tiles and
// chunks can be reduced in any grouping as long as the order is kept.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define XSG_LS_HD __host__ __device__ __forceinline__
#else
#define XSG_LS_HD inline
#endif

namespace xsg {

constexpr uint32_t kSumNl = 1u, kSumF = 2u, kSumL = 4u, kSumCShift = 3u;

XSG_LS_HD uint32_t sum_combine(uint32_t a, uint32_t b) {
  const uint32_t anl = a & kSumNl, bnl = b & kSumNl;
  const uint32_t aF = (a >> 1) & 1u, aL = (a >> 2) & 1u, bF = (b >> 1) & 1u, bL = (b >> 2) & 1u;
  const uint32_t aC = a >> kSumCShift, bC = b >> kSumCShift;
  if (!anl && !bnl) {
    const uint32_t f = aF | bF;
    return (f << 1) | (f << 2);
  }
  if (anl && !bnl) return kSumNl | (aF << 1) | ((aL | bF) << 2) | (aC << kSumCShift);
  if (!anl && bnl) return kSumNl | ((aF | bF) << 1) | (bL << 2) | (bC << kSumCShift);
  return kSumNl | (aF << 1) | (bL << 2) | ((aC + bC + (aL | bF)) << kSumCShift);
}

// number of matching lines of a whole chunk from its summary
XSG_LS_HD uint64_t sum_total_lines(uint32_t s) {
  return (uint64_t)(s >> kSumCShift) + ((s >> 1) & 1u) + ((s & kSumNl) ? ((s >> 2) & 1u) : 0u);
}

// summary of one 16-byte unit from its match-start bits h and newline bits n
// (a position is never both: line modes reject patterns containing '\n')
XSG_LS_HD uint32_t sum_of_unit(uint32_t h, uint32_t n) {
  if (n == 0) {
    const uint32_t f = h != 0;
    return (f << 1) | (f << 2);
  }
  const uint32_t first = n & (0u - n);                      // lowest newline bit
  const uint32_t F = (h & (first - 1u)) != 0;               // match before the first newline
  const uint32_t top = 31u - (uint32_t)__builtin_clz(n);    // index of the highest newline bit
  const uint32_t Lh = (h >> (top + 1u)) != 0;               // match after the last newline
  // Closed segments that hold a match, without walking the newlines: with g = the positions that are neither
  // match nor newline, adding a 1 just above every newline ripples through the gap bits and lands on the first
  // event after that newline (the next newline stops a ripple before it can meet the next injection).  Where it
  // lands on a match, the segment that starts at that newline holds one; it is closed iff it lies below the
  // last newline.
  const uint32_t ev = h | n;
  const uint32_t g = ~ev & 0xffffu;
  const uint32_t first_after_nl = (g + (n << 1)) & ev & h;
  const uint32_t C = (uint32_t)__builtin_popcount(first_after_nl & ((1u << top) - 1u));
  return kSumNl | (F << 1) | (Lh << 2) | (C << kSumCShift);
}

// Combined summary of 64 consecutive units from the lane masks of their unit
// summaries (bit l = lane l): N = has newline, Fm = F flag, Lm = Lh flag, csum =
// sum of the lanes' closed-segment counts.  O(1) integer work instead of a 6-step
// ordered reduction.  A closed segment that spans lanes runs from the trailing
// open part of one newline-lane over whole newline-less lanes to the leading open
// part of the next newline-lane; "does each such variable-width bit field hold a
// set bit" is one 64-bit add with the fields' top bits as carry stops.
XSG_LS_HD uint32_t sum_combine_lanes(unsigned long long N, unsigned long long Fm, unsigned long long Lm, uint32_t csum) {
  if (N == 0) {
    const uint32_t f = Fm != 0;
    return (f << 1) | (f << 2);
  }
  const int i0 = __builtin_ctzll(N);
  const int i1 = 63 - __builtin_clzll(N);
  const unsigned long long upto_i0 = i0 == 63 ? ~0ull : ((2ull << i0) - 1ull);
  const uint32_t Fw = (Fm & upto_i0) != 0;
  const uint32_t Lw = (uint32_t)((Lm >> i1) & 1ull) | (uint32_t)(i1 == 63 ? 0 : ((Fm >> (i1 + 1)) != 0));
  const unsigned long long range = ((1ull << i1) - 1ull) & ~((1ull << i0) - 1ull);  // lanes [i0, i1)
  const unsigned long long A = (Lm & N) | (Fm & ~N);  // what a lane contributes to the segment running right
  const unsigned long long B = Fm & N;                 // leading open part of a newline-lane: closes the field below it
  const unsigned long long X = (A | (B >> 1)) & range;
  const unsigned long long Hm = (N >> 1) & range;      // top bit of every field [i, j): j - 1
  const unsigned long long flags = (((X & ~Hm) + (~Hm & range)) | X) & Hm;
  const uint32_t C = csum + (uint32_t)__builtin_popcountll(flags);
  return kSumNl | (Fw << 1) | (Lw << 2) | (C << kSumCShift);
}

}  // namespace xsg
