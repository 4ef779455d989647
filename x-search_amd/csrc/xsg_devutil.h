// xsg_devutil.h -- device-side helpers shared by the kernel translation units (xsg_kernels.hip: the bulk scan and
// its finish kernel; xsg_list_kernels.hip: ranks, keep steps, tail walks and outputs of the list tags).
// Cross-lane exchanges (wave64), byte tests on 16-byte units, newline searches by a lane and by a whole wave.
#pragma once
#include "xsg_internal.h"
#include "xsg_linesum.h"
#include "xsg_tail.h"

namespace xsg {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr bool is_cls(int kind) { return kind == kClass || kind == kClassFast; }

// ---------------------------------------------------------------------------
// cross-lane helpers (wave64)
// ---------------------------------------------------------------------------
// value of lane+1 (lane 63 receives `edge`)
__device__ __forceinline__ uint32_t from_next_lane(uint32_t x, uint32_t edge, uint32_t lane) {
#if defined(XSG_USE_DPP_SHIFT)
  // v_mov_b32_dpp wave_shl:1 -- lane i reads lane i+1 (gfx9 DPP wavefront shift).
  // Lane 63 has no source lane: with bound_ctrl off it keeps the destination's old
  // value, which is preset to `edge` -- two instructions per dword in all.
  (void)lane;
  return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)x, 0x130, 0xf, 0xf, false);
#else
  const uint32_t y = (uint32_t)__shfl_down((int)x, 1);
  return lane == 63u ? edge : y;
#endif
}

// Wave-wide reductions on the DPP data path: four shifts inside the rows of 16 lanes, then the two row broadcasts
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) leave the total in lane 63; v_readlane makes it
// wave-uniform.  Seven VALU instructions and no LDS: the butterfly over __shfl_xor compiles to six DEPENDENT
// ds_bpermute_b32 (address arithmetic + a trip through the LDS crossbar each) -- ~700 cycles of latency for the two
// reductions in k_scan's epilogue, paid by every wave that holds a match (every wave, for a needle like `the`).
#define XSG_DPP_STEP(OP, x, ctrl, rows) x = OP(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rows, 0xf, false))
__device__ __forceinline__ uint32_t dpp_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t dpp_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  XSG_DPP_STEP(dpp_add, v, 0x111, 0xf);  // row_shr:1
  XSG_DPP_STEP(dpp_add, v, 0x112, 0xf);  // row_shr:2
  XSG_DPP_STEP(dpp_add, v, 0x114, 0xf);  // row_shr:4
  XSG_DPP_STEP(dpp_add, v, 0x118, 0xf);  // row_shr:8   -> lane 15 of every row holds the row's sum
  XSG_DPP_STEP(dpp_add, v, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  XSG_DPP_STEP(dpp_add, v, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// maximum over the wave (unsigned; lanes without a value contribute 0)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  XSG_DPP_STEP(dpp_max, v, 0x111, 0xf);
  XSG_DPP_STEP(dpp_max, v, 0x112, 0xf);
  XSG_DPP_STEP(dpp_max, v, 0x114, 0xf);
  XSG_DPP_STEP(dpp_max, v, 0x118, 0xf);
  XSG_DPP_STEP(dpp_max, v, 0x142, 0xa);
  XSG_DPP_STEP(dpp_max, v, 0x143, 0xc);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
#undef XSG_DPP_STEP
// inclusive prefix sum over lanes: the same DPP sequence as wave_sum_u32 -- it IS a Hillis-Steele scan inside the rows,
// and the two row broadcasts add the totals of the rows below
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, uint32_t lane) {
  (void)lane;
#define XSG_DPP_ADD(x, ctrl, rows) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rows, 0xf, false)
  XSG_DPP_ADD(v, 0x111, 0xf);
  XSG_DPP_ADD(v, 0x112, 0xf);
  XSG_DPP_ADD(v, 0x114, 0xf);
  XSG_DPP_ADD(v, 0x118, 0xf);
  XSG_DPP_ADD(v, 0x142, 0xa);
  XSG_DPP_ADD(v, 0x143, 0xc);
#undef XSG_DPP_ADD
  return v;
}

// ordered reduction over the 64 lanes; result valid in lane 0
__device__ __forceinline__ uint32_t wave_sum_combine(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const uint32_t o = (uint32_t)__shfl_down((int)v, s);
    if ((lane & (uint32_t)(2 * s - 1)) == 0u) v = sum_combine(v, o);
  }
  return v;
}

// Combined summary of the 64 consecutive units of one wave-load: ballots of the
// lanes' unit summaries, then xsg_linesum.h's O(1) mask algebra (result wave-uniform).
__device__ __forceinline__ uint32_t wave_units_combine(uint32_t us) {
  const unsigned long long N = __ballot(us & kSumNl);
  const unsigned long long Fm = __ballot(us & kSumF);
  const unsigned long long Lm = __ballot(us & kSumL);
  const uint32_t c = us >> kSumCShift;  // <= 7 closed segments inside one 16-byte unit
  const uint32_t csum = (uint32_t)__popcll(__ballot(c & 1u)) + 2u * (uint32_t)__popcll(__ballot(c & 2u)) +
                        4u * (uint32_t)__popcll(__ballot(c & 4u));
  return sum_combine_lanes(N, Fm, Lm, csum);
}

// ---------------------------------------------------------------------------
// per-unit byte tests
// ---------------------------------------------------------------------------
// exact 0x80-per-byte flags of bytes equal to '\n'
__device__ __forceinline__ uint32_t nl_flags(uint32_t d) {
  const uint32_t x = d ^ 0x0a0a0a0au;
  const uint32_t t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu;
  return ~t;  // 0x80 in every byte that was '\n'
}
// number of '\n' among the 16 bytes of the unit.  t = nl_flags' intermediate has
// the low 7 bits of every byte set and bit 7 set iff the byte is NOT a newline, so
// popcount(t) = 28 + (non-newline bytes) and the four popcounts chain through
// v_bcnt_u32_b32's accumulator operand: 4 ops per dword + 1.
__device__ __forceinline__ uint32_t nl_count16(const uint32_t (&d)[8]) {
  uint32_t acc = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t x = d[q] ^ 0x0a0a0a0au;
    const uint32_t t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu;
    acc += (uint32_t)__popc(t);
  }
  return 128u - acc;
}
// bit b set <=> byte b of the unit is '\n'
__device__ __forceinline__ uint32_t nl_mask16(const uint32_t (&d)[8]) {
  uint32_t m = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t f = nl_flags(d[q]) >> 7;  // bit 0, 8, 16, 24
    const uint32_t nib = (f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u);
    m |= nib << (4 * q);
  }
  return m;
}
__device__ __forceinline__ bool nl_any16(const uint32_t (&d)[8]) {
  uint32_t acc = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t x = d[q] ^ 0x0a0a0a0au;
    acc |= (x - 0x01010101u) & ~x;
  }
  return (acc & 0x80808080u) != 0;
}

// ---- newline searches over arbitrary distances (list kernels, walk_entry) -----------------------------
// One thread, aligned 16-byte loads only (chunk buffers are padded to 16), 64 bytes per step while nothing is
// found: a line of a few dozen bytes costs one or two loads, and a scan that has to cross megabytes of a
// newline-less line moves 8-16x faster than a byte loop would.
__device__ __forceinline__ uint32_t nl_mask_of_unit(const uint8_t* p) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const uint32_t d[8] = {v.x, v.y, v.z, v.w, 0u, 0u, 0u, 0u};
  return nl_mask16(d);
}
__device__ __forceinline__ uint32_t unit_has_nl(const uint8_t* p) {  // 0 / 1; combined with | so that the loads stay independent
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const uint32_t d[8] = {v.x, v.y, v.z, v.w, 0u, 0u, 0u, 0u};
  return nl_any16(d) ? 1u : 0u;
}
// does any of the 256 bytes at p (16-byte aligned) hold a newline?  Sixteen independent loads, one combined test.
__device__ __forceinline__ bool block_has_nl(const uint8_t* p) {
  uint32_t acc = 0;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const uint4 v = *reinterpret_cast<const uint4*>(p + u * kUnit);
    const uint32_t x0 = v.x ^ 0x0a0a0a0au, x1 = v.y ^ 0x0a0a0a0au, x2 = v.z ^ 0x0a0a0a0au, x3 = v.w ^ 0x0a0a0a0au;
    acc |= ((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1) | ((x2 - 0x01010101u) & ~x2) | ((x3 - 0x01010101u) & ~x3);
  }
  return (acc & 0x80808080u) != 0;  // exact as an existence test
}
// 64 consecutive bytes at p (16-byte aligned) as one newline mask; units at or beyond `end` are not read.  The four
// loads are independent: a line of text (~30 bytes) is settled by ONE memory latency instead of two or three
// dependent ones -- the list kernels are chains of such latencies, little else.
__device__ __forceinline__ uint64_t nl_mask_of_64(const uint8_t* d, uint64_t p, uint64_t end) {
  uint64_t m = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (p + (uint64_t)u * kUnit < end) m |= (uint64_t)(nl_mask_of_unit(d + p + (uint64_t)u * kUnit) & 0xffffu) << (16 * u);
  return m;
}
// offset of the first '\n' in d[lo, hi), or -1
__device__ __forceinline__ int64_t first_newline_in(const uint8_t* d, uint64_t lo, uint64_t hi) {
  if (lo >= hi) return -1;
  uint64_t p = lo & ~(uint64_t)15;
  {  // first step: the unit that holds lo and the three behind it
    uint64_t m = nl_mask_of_64(d, p, hi) & (~0ull << (uint32_t)(lo - p));
    if (hi - p < 64u) m &= (1ull << (uint32_t)(hi - p)) - 1ull;
    if (m) return (int64_t)(p + (uint32_t)__builtin_ctzll(m));
    p += 4 * kUnit;
  }
  for (;;) {
    // past the first step: skip 256, then 64 bytes at a time while they hold no newline
    while (p + 16 * kUnit <= hi && !block_has_nl(d + p)) p += 16 * kUnit;
    while (p + 4 * kUnit <= hi && !(unit_has_nl(d + p) | unit_has_nl(d + p + kUnit) | unit_has_nl(d + p + 2 * kUnit) |
                                    unit_has_nl(d + p + 3 * kUnit)))
      p += 4 * kUnit;
    if (p >= hi) return -1;
    uint32_t m = nl_mask_of_unit(d + p) & 0xffffu;
    if (hi - p < kUnit) m &= (1u << (uint32_t)(hi - p)) - 1u;
    if (m) return (int64_t)(p + (uint32_t)__ffs((int)m) - 1u);
    p += kUnit;
  }
}
// offset of the last '\n' in d[lo, hi), or -1
__device__ __forceinline__ int64_t last_newline_in(const uint8_t* d, uint64_t lo, uint64_t hi) {
  if (lo >= hi) return -1;
  uint64_t p = (hi - 1u) & ~(uint64_t)15;  // unit of the last byte of the range
  const uint64_t lo_unit = lo & ~(uint64_t)15;
  {  // first step: the (up to) four units that end with the one holding hi - 1
    const uint64_t q = p >= lo_unit + 3 * kUnit ? p - 3 * kUnit : lo_unit;  // first unit of the group
    uint64_t m = nl_mask_of_64(d, q, p + kUnit);
    if (hi - q < 64u) m &= (1ull << (uint32_t)(hi - q)) - 1ull;
    if (q < lo) m &= ~0ull << (uint32_t)(lo - q);
    if (m) return (int64_t)(q + 63u - (uint32_t)__builtin_clzll(m));
    if (q <= lo_unit) return -1;
    p = q - kUnit;
  }
  uint32_t range = 0xffffu;
  for (;;) {
    if (range == 0xffffu) {
      while (p >= lo_unit + 16 * kUnit && !block_has_nl(d + p - 15 * kUnit)) p -= 16 * kUnit;
      while (p >= lo_unit + 4 * kUnit && !(unit_has_nl(d + p) | unit_has_nl(d + p - kUnit) | unit_has_nl(d + p - 2 * kUnit) |
                                           unit_has_nl(d + p - 3 * kUnit)))
        p -= 4 * kUnit;
    }
    uint32_t m = nl_mask_of_unit(d + p) & range;
    if (p < lo) m &= 0xffffu << (uint32_t)(lo - p);  // only in the unit that holds lo
    m &= 0xffffu;
    if (m) return (int64_t)(p + 31u - (uint32_t)__clz(m));
    if (p <= lo_unit) return -1;
    p -= kUnit;
    range = 0xffffu;
  }
}

// ---- the same searches by a whole wave ---------------------------------------------------------------
// One lane moves ~0.25 GB/s through a newline-less stretch however the loads are arranged; a line of hundreds of
// megabytes (a minified file, a binary blob) would keep a list kernel busy for seconds per scan.  All 64 lanes
// together read 4 KiB per step.  Every lane must call these with the SAME arguments (and all 64 must be active).
__device__ __forceinline__ int64_t wave_first_newline_in(const uint8_t* d, uint64_t lo, uint64_t hi, uint32_t lane) {
  if (lo >= hi) return -1;
  constexpr int U = 4;
  for (uint64_t p = lo & ~(uint64_t)15; p < hi; p += (uint64_t)U * 64u * kUnit) {
    uint32_t m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t a = p + ((uint64_t)u * 64u + lane) * kUnit;
      uint32_t x = 0;
      if (a < hi) {
        x = nl_mask_of_unit(d + a) & 0xffffu;
        if (a < lo) x &= 0xffffu << (uint32_t)(lo - a);  // the unit that holds lo
        if (hi - a < kUnit) x &= (1u << (uint32_t)(hi - a)) - 1u;
      }
      m[u] = x;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long b = __ballot(m[u] != 0);
      if (b) {
        const int L = __builtin_ctzll(b);  // lowest address of the group
        const uint32_t mm = (uint32_t)__shfl((int)m[u], L);
        return (int64_t)(p + ((uint64_t)u * 64u + (uint64_t)L) * kUnit + (uint32_t)__ffs((int)mm) - 1u);
      }
    }
  }
  return -1;
}
__device__ __forceinline__ int64_t wave_last_newline_in(const uint8_t* d, uint64_t lo, uint64_t hi, uint32_t lane) {
  if (lo >= hi) return -1;
  constexpr int U = 4;
  const uint64_t top = (hi - 1u) & ~(uint64_t)15;               // unit of the last byte
  const uint64_t K = (top - (lo & ~(uint64_t)15)) / kUnit;      // units are numbered downwards from the top: 0..K
  for (uint64_t kb = 0; kb <= K; kb += (uint64_t)U * 64u) {
    uint32_t m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t k = kb + (uint64_t)u * 64u + lane;
      uint32_t x = 0;
      if (k <= K) {
        const uint64_t a = top - k * kUnit;
        x = nl_mask_of_unit(d + a) & 0xffffu;
        if (a < lo) x &= 0xffffu << (uint32_t)(lo - a);
        if (hi - a < kUnit) x &= (1u << (uint32_t)(hi - a)) - 1u;
      }
      m[u] = x;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long b = __ballot(m[u] != 0);
      if (b) {
        const int L = __builtin_ctzll(b);  // smallest k = highest address of the group
        const uint32_t mm = (uint32_t)__shfl((int)m[u], L);
        const uint64_t a = top - (kb + (uint64_t)u * 64u + (uint64_t)L) * kUnit;
        return (int64_t)(a + 31u - (uint32_t)__clz(mm));
      }
    }
  }
  return -1;
}

// A newline query per lane (live lanes only), FORWARD: first in [lo, hi), else last in [lo, hi).  Each lane looks
// kSoloScan bytes far on its own -- that settles every ordinary line -- and the whole wave then finishes the
// queries that are still open, one after the other.  All 64 lanes of the wave must call this together.
constexpr uint64_t kSoloScan = 4096;
template <bool FORWARD>
__device__ __forceinline__ int64_t newline_query(bool live, const uint8_t* d, uint64_t lo, uint64_t hi, uint32_t lane) {
  int64_t res = -1;
  bool pending = false;
  if (live && lo < hi) {
    if (FORWARD) {
      const uint64_t cut = hi - lo > kSoloScan ? lo + kSoloScan : hi;
      res = first_newline_in(d, lo, cut);
      if (res < 0 && cut < hi) pending = true, lo = cut;
    } else {
      const uint64_t cut = hi - lo > kSoloScan ? hi - kSoloScan : lo;
      res = last_newline_in(d, cut, hi);
      if (res < 0 && cut > lo) pending = true, hi = cut;
    }
  }
  unsigned long long pend = __ballot(pending);
  while (pend) {  // wave-uniform
    const int L = __builtin_ctzll(pend);
    const uint8_t* dd = reinterpret_cast<const uint8_t*>((uintptr_t)__shfl((long long)(uintptr_t)d, L));
    const uint64_t l2 = (uint64_t)__shfl((long long)lo, L), h2 = (uint64_t)__shfl((long long)hi, L);
    const int64_t r = FORWARD ? wave_first_newline_in(dd, l2, h2, lane) : wave_last_newline_in(dd, l2, h2, lane);
    if ((int)lane == L) res = r;
    pend &= pend - 1ull;
  }
  return res;
}


__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t* sh) {
  // sh: kWaves entries
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += (uint64_t)__shfl_xor((long long)v, s);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  uint64_t t = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < kWaves; ++w) t += sh[w];
  __syncthreads();
  return t;  // valid in thread 0
}

// where the reference walk stands when it reaches the tail zone of a chunk: after the last bulk match (match
// modes), at the start of the line after the last bulk matching line (line modes; UINT64_MAX if that line has
// no '\n': the walk ended).  Wave-uniform arguments, all lanes active.
__device__ __forceinline__ uint64_t wave_walk_entry(const uint8_t* d, uint64_t L, uint64_t last_end, bool skip_to_nl,
                                                   uint32_t lane) {
  if (last_end == 0) return 0;
  if (!skip_to_nl) return last_end;
  const int64_t nl = wave_first_newline_in(d, last_end, L, lane);
  return nl < 0 ? UINT64_MAX : (uint64_t)nl + 1u;
}

constexpr uint32_t kZoneStage = 96;  // bytes of a chunk's tail zone staged in LDS: <= 64 positions + 2 x 15 of alignment

// The tail zone of one chunk, decided by the whole wave (xsg_tail.h, tail_walk_masks): lane i owns position Z + i,
// computes how many leading pattern bytes match there (from the zone's bytes staged in LDS -- a byte-by-byte walk
// by one lane through global memory cost ~10 us per chunk in dependent loads), ballots give the masks and the walk
// itself runs on the scalar unit.  Returns (wave-uniform) the matches of the match-mode walk from `entry_m` and
// of the line-mode walk from `entry_l`.  plen <= kTailMaskMaxPlen; all 64 lanes active.
__device__ __forceinline__ void wave_tail_counts(const uint8_t* d, uint64_t L, const uint8_t* s_pat, uint32_t plen,
                                                 bool icase, uint8_t* zone, uint32_t lane, bool want_m, uint64_t entry_m,
                                                 bool want_l, uint64_t entry_l, uint32_t* n_m, uint32_t* n_l,
                                                 uint64_t* out = nullptr, uint32_t out_cap = 0) {
  const uint64_t Z = tail_zone_begin(L, plen);
  const uint32_t n = (uint32_t)(L - Z);  // <= plen + 31 <= 64
  const uint64_t Zal = Z & ~(uint64_t)15;
  const uint64_t Lr = (L + 15u) & ~(uint64_t)15u;
  const uint32_t off = (uint32_t)(Z - Zal);
  if (Zal + (uint64_t)lane * kUnit < Lr && lane < kZoneStage / kUnit)
    *reinterpret_cast<uint4*>(zone + lane * kUnit) = *reinterpret_cast<const uint4*>(d + Zal + (uint64_t)lane * kUnit);
  // the zone row belongs to this wave alone: a wavefront-scope release/acquire orders its lanes' LDS accesses
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool in_zone = lane < n;
  const bool room = in_zone && (L - (Z + lane) >= plen);
  uint32_t k = 0;
  bool alive = room;
  for (uint32_t j = 0; j < plen; ++j) {
    if (!__any(alive)) break;
    if (alive) {
      if (fold(zone[off + lane + j], icase) == s_pat[j])
        k = j + 1u;
      else
        alive = false;
    }
  }
  const unsigned long long full = __ballot(room && k == plen);
  const unsigned long long nz = __ballot(k != 0);
  const unsigned long long nlm = __ballot(in_zone && zone[off + lane] == '\n');
  auto k_at = [&](uint32_t j) { return (uint32_t)__builtin_amdgcn_readlane((int)k, (int)__builtin_amdgcn_readfirstlane(j)); };
  // (out: the positions of the ONE walk asked for -- the list kernels want either the match-mode or the line-mode walk)
  *n_m = want_m ? tail_walk_masks(L, plen, entry_m, false, full, nz, nlm, k_at, want_l ? nullptr : out, out_cap) : 0u;
  *n_l = want_l ? tail_walk_masks(L, plen, entry_l, true, full, nz, nlm, k_at, out, out_cap) : 0u;
}

}  // namespace xsg
