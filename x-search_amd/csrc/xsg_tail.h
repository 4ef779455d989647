// xsg_tail.h -- emulation of the reference's end-of-chunk behaviour, shared by
// the device "finish" kernels and by host-side unit tests (plain C++ when
// compiled without hipcc).
//
// Background (reference src/string_search/simd_search.cpp:162-204): strstr
// searches 32-byte blocks exactly while at least 32+plen bytes remain, then
// hands the rest to scalar_strstr (:58-78), which is LOSSY: after a partial
// prefix match of k bytes it resumes k+1 bytes further on, skipping occurrences
// that overlap the partial match.  Where the exact part stops depends on where
// the call started (the end of the previous match), so the last plen+31 bytes
// of a chunk cannot be decided position-by-position: they are replayed
// sequentially here, exactly as the reference's walk would
// (include/xsearch/string_search/search_wrappers.h:29-52).
//
// Contract used by the kernels: the bulk scan reports every occurrence that
// starts at o < Z, Z = tail_zone_begin(L, plen).  This file decides [Z, L).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define XSG_HD __host__ __device__ inline
#else
#define XSG_HD inline
#endif

namespace xsg {

// First offset whose outcome depends on the sequential walk.  plen == 1 goes
// through strchr, which is exact everywhere -> no zone.
XSG_HD uint64_t tail_zone_begin(uint64_t L, uint32_t plen) {
  if (plen <= 1) return L;
  const uint64_t z = (uint64_t)plen + 31u;
  return L > z ? L - z : 0;
}

// simd::toLower on one byte (src/utils/string_utils.cpp:11-33): 'A'..'Z' -> 'a'..'z'.
// With icase the pattern has already been lowered by the host.
XSG_HD uint8_t fold(uint8_t b, bool icase) { return (icase && b >= 'A' && b <= 'Z') ? (uint8_t)(b + 32) : b; }

// scalar_strstr (simd_search.cpp:58-78) on d[from, L): absolute offset or -1.
XSG_HD int64_t lossy_scalar_find(const uint8_t* d, uint64_t from, uint64_t L, const uint8_t* pat, uint32_t plen,
                                 bool icase) {
  uint64_t shift = from;
  while (shift < L) {
    if (L - shift < plen) return -1;
    uint32_t k = 0;
    while (k < plen && fold(d[shift + k], icase) == pat[k]) ++k;
    if (k == plen) return (int64_t)shift;
    shift += (uint64_t)k + 1u;
  }
  return -1;
}

XSG_HD bool occurs_at(const uint8_t* d, uint64_t o, const uint8_t* pat, uint32_t plen, bool icase) {
  for (uint32_t k = 0; k < plen; ++k)
    if (fold(d[o + k], icase) != pat[k]) return false;
  return true;
}

// findNext(pattern, d, L, shift) (simd_search.cpp:289-295) for plen >= 2,
// GIVEN that no occurrence starts in [shift, Z) (the bulk scan found none).
XSG_HD int64_t tail_find_next(const uint8_t* d, uint64_t L, const uint8_t* pat, uint32_t plen, uint64_t shift,
                              uint64_t Z, bool icase) {
  if (shift > L) return -1;
  const uint64_t R = L - shift;
  if (R < 32u + (uint64_t)plen) return lossy_scalar_find(d, shift, L, pat, plen, icase);
  const uint64_t T = shift + 32u * ((R - plen) / 32u);  // end of the exact 32-byte-block part; Z <= T <= L-plen
  for (uint64_t o = shift > Z ? shift : Z; o < T; ++o)
    if (occurs_at(d, o, pat, plen, icase)) return (int64_t)o;
  return lossy_scalar_find(d, T, L, pat, plen, icase);
}

XSG_HD int64_t next_newline(const uint8_t* d, uint64_t from, uint64_t L) {
  for (uint64_t i = from; i < L; ++i)
    if (d[i] == '\n') return (int64_t)i;
  return -1;
}

// Replays the reference walk over the tail zone.
//   shift0     : where the walk stands when it reaches the zone: end of the
//                last bulk match (match modes) / start of the line after the
//                last bulk matching line (line modes) / 0 if there was none.
//                In line modes the caller passes shift0 = UINT64_MAX when the
//                walk already ended (last matching line had no '\n').
//   skip_to_nl : line modes (search_wrappers.h:43-49)
//   out        : receives up to cap match offsets (chunk-local); may be null
// Returns the number of matches the walk finds in the zone.
XSG_HD uint32_t tail_walk(const uint8_t* d, uint64_t L, const uint8_t* pat, uint32_t plen, uint64_t shift0,
                          bool skip_to_nl, uint64_t* out, uint32_t cap, bool icase = false) {
  if (plen <= 1 || shift0 == UINT64_MAX) return 0;
  const uint64_t Z = tail_zone_begin(L, plen);
  uint32_t n = 0;
  uint64_t shift = shift0;
  while (shift < L) {
    const int64_t m = tail_find_next(d, L, pat, plen, shift, Z, icase);
    if (m < 0) break;
    if (out && n < cap) out[n] = (uint64_t)m;
    ++n;
    shift = (uint64_t)m + plen;
    if (skip_to_nl) {
      const int64_t nl = next_newline(d, shift, L);
      if (nl < 0) break;
      shift = (uint64_t)nl + 1u;
    }
  }
  return n;
}

// ---------------------------------------------------------------------------
// The same walk on bit masks (k_count_finish: one wave per chunk).
// A zone of at most 64 positions (plen <= kTailMaskMaxPlen): position i stands for
// offset Z + i.  Per position, decided by 64 lanes in parallel:
//   K[i]  = number of leading pattern bytes that match at Z+i (0 if fewer than plen
//           bytes remain: the scalar search gives up there, simd_search.cpp:62)
//   full  = bit i <=> K[i] == plen,  nz = bit i <=> K[i] > 0,  nlm = bit i <=> d[Z+i] == '\n'
// The lossy scalar search then never looks at bytes: from `cur` it slides over
// positions with K == 0 one by one (next candidate = lowest nz bit at or above cur),
// and at a position with 0 < K < plen it resumes K+1 bytes further on -- exactly
// scalar_strstr's `shift = str_index` (:75).  The chain has one step per VISITED
// partial match, not per byte.  k_at(i) returns K[i] (device: v_readlane).
// ---------------------------------------------------------------------------
constexpr uint32_t kTailMaskMaxPlen = 33;  // plen + 31 <= 64 positions

XSG_HD uint64_t bits_from(uint32_t i) { return i >= 64 ? 0ull : ~0ull << i; }  // bits [i, 64)

template <typename KAt>
XSG_HD int64_t lossy_scalar_find_masks(uint64_t cur, uint64_t L, uint64_t Z, uint32_t plen, uint64_t full, uint64_t nz,
                                       KAt k_at) {
  for (;;) {
    if (cur >= L || L - cur < plen) return -1;
    const uint64_t above = nz & bits_from((uint32_t)(cur - Z));
    if (!above) return -1;  // only mismatches at the first byte up to where the room ends
    const uint32_t j = (uint32_t)__builtin_ctzll(above);
    if ((full >> j) & 1ull) return (int64_t)(Z + j);
    cur = Z + j + (uint64_t)k_at(j) + 1u;
  }
}

// out (optional): receives up to cap match offsets (chunk-local), as tail_walk's does.
template <typename KAt>
XSG_HD uint32_t tail_walk_masks(uint64_t L, uint32_t plen, uint64_t shift0, bool skip_to_nl, uint64_t full, uint64_t nz,
                                uint64_t nlm, KAt k_at, uint64_t* out = nullptr, uint32_t cap = 0) {
  if (plen <= 1 || shift0 == UINT64_MAX) return 0;
  const uint64_t Z = tail_zone_begin(L, plen);
  uint32_t n = 0;
  uint64_t shift = shift0;
  while (shift < L) {
    // tail_find_next
    int64_t m;
    const uint64_t R = L - shift;
    if (R < 32u + (uint64_t)plen) {
      m = lossy_scalar_find_masks(shift, L, Z, plen, full, nz, k_at);  // shift >= Z here
    } else {
      const uint64_t T = shift + 32u * ((R - plen) / 32u);
      const uint64_t a = shift > Z ? shift : Z;
      const uint64_t exact = a < T ? (full & bits_from((uint32_t)(a - Z)) & ~bits_from((uint32_t)(T - Z))) : 0ull;
      m = exact ? (int64_t)(Z + (uint32_t)__builtin_ctzll(exact)) : lossy_scalar_find_masks(T, L, Z, plen, full, nz, k_at);
    }
    if (m < 0) break;
    if (out && n < cap) out[n] = (uint64_t)m;
    ++n;
    shift = (uint64_t)m + plen;
    if (skip_to_nl) {
      const uint64_t after = shift >= L ? 0ull : (nlm & bits_from((uint32_t)(shift - Z)));
      if (!after) break;
      shift = Z + (uint32_t)__builtin_ctzll(after) + 1u;
    }
  }
  return n;
}

// Upper bound on matches tail_walk can return: they are >= plen apart in a
// zone of plen+31 bytes.
XSG_HD uint32_t tail_max_matches(uint32_t plen) { return plen <= 1 ? 0u : (plen + 31u) / plen + 1u; }

}  // namespace xsg
