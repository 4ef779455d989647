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
  uint32_t C = 0;
  uint32_t rest = n & (n - 1u);  // newlines after the first
  uint32_t lo = first;           // a closed segment = the bits strictly between lo and the next newline bit
  while (rest) {
    const uint32_t nx = rest & (0u - rest);
    const uint32_t between = (nx - 1u) & ~(lo | (lo - 1u));
    C += (h & between) != 0;
    lo = nx;
    rest &= rest - 1u;
  }
  return kSumNl | (F << 1) | (Lh << 2) | (C << kSumCShift);
}

}  // namespace xsg
