// xsg_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the literal scan.
//
// What they replace in the reference (all byte/integer work, no MFMA):
//   k_scan        simd::strstr body + the per-chunk walks that call it
//                 (src/string_search/simd_search.cpp:162-204,
//                  include/xsearch/string_search/search_wrappers.h:29-52,163-185)
//   k_count_finish  the end-of-chunk part of those walks (simd_search.cpp:58-78,203)
//   k_line_*      previous_new_line_offset_relative_to_match + findNextNewLine
//                 (search_wrappers.h:111-123,187-207; simd_search.cpp:116-144,297-303)
//
// Design: the bulk scan is a pure HBM stream.  Every lane reads
// 16-byte units with non-temporal global_load_dwordx4 (a wave-instruction = 1 KiB
// contiguous, all loads of a wave issued before the first use), fetches the first
// 8 bytes of its right neighbour's unit with DPP wave shifts (no LDS traffic),
// builds the 20 unaligned dword windows of its unit with v_alignbyte_b32 and
// compares them against the first 8 pattern bytes held in SGPRs; the lane masks
// are OR-ed on the scalar unit.  A wave-load whose 64 lanes see no candidate --
// the normal case at text match densities -- costs ~3 VALU ops per byte and
// leaves the loop without touching LDS or memory again, and a wave that found
// nothing writes nothing (the per-tile arrays are preset by the host).
// Everything else (exact verification, popcounts, ordered compaction, line
// summaries) lives in a wave-uniform slow path.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "xsg_devutil.h"
#include "xsg_linesum.h"
#include "xsg_tail.h"

namespace xsg {

// (cross-lane helpers, byte tests and newline searches: xsg_devutil.h)

constexpr uint32_t kLdsPattern = 1024;  // bytes of a long pattern kept in LDS (k_scan<kLong>); the rest is read from its device copy

// w[b] = the 4 bytes starting at byte b of the lane's 32-byte view (own unit + neighbour's)
// simd::toLower on 4 bytes at once (src/utils/string_utils.cpp:11-33): bytes in
// 'A'..'Z' get bit 5 set, everything else (incl. bytes >= 0x80) is unchanged.
__device__ __forceinline__ uint32_t fold4(uint32_t x) {
  const uint32_t t = x & 0x7f7f7f7fu;
  const uint32_t ge_A = t + 0x3f3f3f3fu;  // bit 7 set <=> low 7 bits >= 0x41
  const uint32_t gt_Z = t + 0x25252525u;  // bit 7 set <=> low 7 bits >= 0x5b
  const uint32_t upper = ge_A & ~gt_Z & ~x & 0x80808080u;
  return x | (upper >> 2);
}

template <int N>
__device__ __forceinline__ void windows(const uint32_t (&d)[8], uint32_t (&w)[N]) {
#pragma unroll
  for (int b = 0; b < N; ++b) {
    const int q = b >> 2, r = b & 3;
    w[b] = r ? __builtin_amdgcn_alignbyte(d[q + 1], d[q], (uint32_t)r) : d[q];
  }
}

template <int KIND>
__device__ __forceinline__ bool cand_at(const uint32_t (&w)[20], int b, const PatternDev& P) {
  if (KIND == kMask1) return (w[b] & P.m0) == P.p0;
  if (is_cls(KIND)) return ((w[b] & P.m0) == P.p0) & ((w[b + 4] & P.m1) == P.p1);
  if (KIND == kOne) return w[b] == P.p0;
  if (KIND == kMask2) return (w[b] == P.p0) & ((w[b + 4] & P.m1) == P.p1);
  return (w[b] == P.p0) & (w[b + 4] == P.p1);
}

// true if any of the 16 positions of the unit passes the prefix filter
template <int KIND>
__device__ __forceinline__ bool cand_any(const uint32_t (&d)[8], const PatternDev& P) {
  uint32_t w[20];
  windows<20>(d, w);
  bool any = false;
#pragma unroll
  for (int b = 0; b < 16; ++b) any |= cand_at<KIND>(w, b, P);
  return any;
}

// The hot trigger of the 8-byte-window kinds (kTwo, kLong, kClass): is there a lane whose unit MAY hold a window?
// Instead of building the 20 unaligned windows of the unit and comparing each against the two window dwords
// (15 v_alignbyte + 32 v_cmp per 16 bytes), the ALIGNED dwords of the view are compared against the four
// sub-dwords of the window: a window that starts at byte b of the unit, b = 4i + r, contains the aligned dword
// d[i+1] (r != 0) or d[i] (r == 0) at window offset (4 - r) % 4, so
//     r = 0: d[i] == W[0:4] and d[i+1] == W[4:8]     r = 1: d[i+1] == W[3:7]
//     r = 2: d[i+1] == W[2:6]                         r = 3: d[i+1] == W[1:5]
// 25 compares on d[0..5], no byte shuffling at all; the sub-dwords are wave-uniform (scalar registers).  A superset
// of the exact candidates (three of the four alignments test 4 of the 8 bytes): a wave-load that triggers runs the
// exact slow path.  Whole-word decoys of the window's halves ("Sher", "lock" for `Sherlock`) only pass the r = 0
// test together; the other alignments look at "herl", "erlo", "rloc", which are words of nobody's lexicon.
template <int KIND>
__device__ __forceinline__ bool trigger_aligned(const uint32_t (&d)[8], const PatternDev& P) {
  const uint64_t w = ((uint64_t)P.p1 << 32) | P.p0;
  const uint32_t a1 = (uint32_t)(w >> 8), a2 = (uint32_t)(w >> 16), a3 = (uint32_t)(w >> 24);
  bool t = false;
  if (is_cls(KIND)) {  // masked: a position of the window pins only the bits its set agrees on
    const uint64_t mw = ((uint64_t)P.m1 << 32) | P.m0;
    const uint32_t m1 = (uint32_t)(mw >> 8), m2 = (uint32_t)(mw >> 16), m3 = (uint32_t)(mw >> 24);
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      t |= ((d[q] & m1) == a1) | ((d[q] & m2) == a2) | ((d[q] & m3) == a3);
      t |= ((d[q] & P.m0) == P.p0) & ((d[q + 1] & P.m1) == P.p1);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      t |= (d[q] == a1) | (d[q] == a2) | (d[q] == a3);
      t |= (d[q] == P.p0) & (d[q + 1] == P.p1);
    }
  }
  return t;
}

template <int KIND>
__device__ __forceinline__ uint32_t cand_mask16(const uint32_t (&d)[8], const PatternDev& P) {
  uint32_t w[20];
  windows<20>(d, w);
  uint32_t m = 0;
#pragma unroll
  for (int b = 0; b < 16; ++b) m |= (uint32_t)cand_at<KIND>(w, b, P) << b;
  return m;
}

// Exact match bits of one unit: filter, position limit, long-pattern verify.
// Bit b stands for the filter WINDOW at byte b of the unit; the match it belongs to
// starts P.koff bytes earlier (koff is 0 except for long patterns).
// m: the window filter's candidate bits of the unit (cand_mask16, or the hot filter's own compare results).
template <int KIND, bool ICASE>
__device__ __forceinline__ uint32_t match_mask16_from(uint32_t m, const uint32_t (&d)[8], const PatternDev& P,
                                                      const uint8_t* cbase, uint64_t unit_off, uint64_t limit,
                                                      const uint8_t* lds_pat, uint8_t* lds_view) {
  const uint32_t koff = KIND >= kLong ? P.koff : 0u;
  // the match starts at o = window - koff and must satisfy 0 <= o < limit
  const uint64_t lim_w = limit + koff;
  if (unit_off >= lim_w) return 0;
  if (unit_off + kUnit > lim_w) m &= (1u << (uint32_t)(lim_w - unit_off)) - 1u;
  if (KIND >= kLong && unit_off < koff)
    m &= koff - unit_off >= kUnit ? 0u : ~((1u << (uint32_t)(koff - unit_off)) - 1u);
  if (KIND == kLong) {
    // everything outside the window [koff, koff+8): compare from memory against the pattern in LDS (rare: the
    // 8 bytes of the window already matched).  Tried and dropped: an in-register check of the next 8 bytes (28
    // windows live: 69-87 VGPRs), and, in round 2, comparing from the lane's 32-byte view parked in LDS as the
    // class sequences do -- 60 VGPRs / 100 SGPRs instead of 47 / 74 cost the memory-bound long patterns an eighth
    // of their rate (`Sherlock Holmes` 7.0 -> 6.1 TB/s) and bought `detective street` nothing: its cost is how
    // often its window occurs (xsg_api.cpp picks the window by measurement now), not how a candidate is verified.
    uint32_t c = m;
    while (c) {
      const uint32_t b = (uint32_t)__ffs((int)c) - 1u;
      c &= c - 1u;
      const uint8_t* s = cbase + unit_off + b - koff;  // start of the match
      // eight bytes between exits: their loads are independent and wait once (the tile is streamed with
      // non-temporal loads, so these come from HBM: a byte at a time with an exit after each was a chain of
      // ~2 us stalls per candidate -- `detective street` spent 65 % of its wave cycles in s_waitcnt)
      uint32_t diff = 0;
      for (uint32_t k0 = 0; k0 < P.plen && !diff; k0 += 8u) {
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
          const uint32_t k = k0 + j;
          if (k < P.plen && (k < koff || k >= koff + 8u)) diff |= (uint32_t)(fold(s[k], ICASE) ^ (k < kLdsPattern ? lds_pat[k] : P.d_pat[k]));
        }
      }
      if (diff) m &= ~(1u << b);
    }
  }
  if (is_cls(KIND) && !P.cls_exact) {
    // The window compare saw only the bits the members of each set agree on: now every position of the candidate
    // against its 256-bit set in LDS (8 dwords per position).  A candidate whose bytes all lie in the lane's
    // 32-byte view (own unit + the next one, d[0..8)) is decided without touching memory and without an early
    // exit, so the LDS reads of all positions are in flight together.  (Byte
    // loads from global memory with an early exit -- the first version -- cost a microsecond per candidate:
    // `She[r ]lock`, which really occurs every few KiB of the bench corpus, ran at 4.6 TB/s.)  Other candidates
    // (match starts before the unit, expressions over kRegVerify positions) read memory, also without an early exit.
    constexpr uint32_t kRegVerify = 12;
    const uint32_t* sets = reinterpret_cast<const uint32_t*>(lds_pat);
    // the lane's view goes to its 48-byte slot in LDS once (the wave is here together: no lane is missing, and a
    // wave's LDS accesses execute in order), a candidate then reads its dwords at its own byte offset -- LDS takes
    // unaligned addresses -- instead of shifting 8 registers into place (which cost 25 VGPRs in every variant)
    uint8_t* lane_view = lds_view + (threadIdx.x * 48u);
    if (m) {
      *reinterpret_cast<uint4*>(lane_view) = make_uint4(d[0], d[1], d[2], d[3]);
      *reinterpret_cast<uint4*>(lane_view + 16) = make_uint4(d[4], d[5], d[6], d[7]);
    }
    uint32_t c = m;
    while (c) {
      const uint32_t b = (uint32_t)__ffs((int)c) - 1u;
      c &= c - 1u;
      const int32_t start = (int32_t)b - (int32_t)koff;
      uint32_t ok = 1u;
      if (start >= 0 && (uint32_t)start + P.plen <= 32u && P.plen <= kRegVerify) {
        typedef uint32_t u32_unaligned __attribute__((aligned(1)));
        uint32_t w[kRegVerify / 4];
#pragma unroll
        for (int i = 0; i < (int)kRegVerify / 4; ++i)
          w[i] = *reinterpret_cast<const u32_unaligned*>(lane_view + (uint32_t)start + 4u * (uint32_t)i);
        uint32_t any = 0;
        for (uint32_t a = 0; a < P.nalt; ++a) {  // one alternative must accept every position
          const uint32_t* sa = sets + a * P.plen * 8u;
          uint32_t oka = 1u;
#pragma unroll
          for (int k = 0; k < (int)kRegVerify; ++k) {
            if ((uint32_t)k < P.plen) {
              const uint32_t x = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;  // already folded when ICASE
              oka &= sa[(uint32_t)k * 8u + (x >> 5)] >> (x & 31u);
            }
          }
          any |= oka;
        }
        ok = any;
      } else {
        const uint8_t* s = cbase + unit_off + b - koff;
        uint32_t alive = P.nalt >= 32u ? 0xffffffffu : (1u << P.nalt) - 1u;  // bit a: alternative a still accepts
        for (uint32_t k = 0; k < P.plen; ++k) {
          const uint32_t x = fold(s[k], ICASE);
          uint32_t acc = 0;
          for (uint32_t a = 0; a < P.nalt; ++a) acc |= ((sets[(a * P.plen + k) * 8u + (x >> 5)] >> (x & 31u)) & 1u) << a;
          alive &= acc;
        }
        ok = alive != 0;
      }
      if (!(ok & 1u)) m &= ~(1u << b);
    }
  }
  return m;
}

template <int KIND, bool ICASE>
__device__ __forceinline__ uint32_t match_mask16(const uint32_t (&d)[8], const PatternDev& P, const uint8_t* cbase,
                                                 uint64_t unit_off, uint64_t limit, const uint8_t* lds_pat,
                                                 uint8_t* lds_view) {
  return match_mask16_from<KIND, ICASE>(cand_mask16<KIND>(d, P), d, P, cbase, unit_off, limit, lds_pat, lds_view);
}

// Class sequences of at most 8 positions (the window IS the expression: koff = 0) decide their candidates in
// registers, position by position of the unit.  `cm` is the lane mask of the candidates at byte b of the unit
// (wave-uniform: most of the 16 are zero and cost a scalar test); the window's 8 bytes are w[b], w[b + 4] -- STATIC
// registers, no view in LDS, no per-lane bit loop -- and only the positions the hot filter did not already decide
// exactly (PatternDev::cls_chk; `She[r ]lock` under the 16 + 32 bit filter: two of eight) are looked up in the
// 256-bit sets.  Lanes outside `cm` compute along (their lookups stay inside the sets) and are masked at the end.
// Replaces, for these expressions, the LDS-view verification of match_mask16_from: 250 VALU + 22 LDS instructions
// per wave-load that entered the slow path became ~40 (profiles/r03_class_ab.txt).
// window dword at byte B of the lane's view, on demand (B static: one v_alignbyte, no array of windows kept alive)
template <int B>
__device__ __forceinline__ uint32_t win_at(const uint32_t (&d)[8]) {
  constexpr int q = B >> 2, r = B & 3;
  return r ? __builtin_amdgcn_alignbyte(d[q + 1], d[q], (uint32_t)r) : d[q];
}
// does some lane hold a candidate at byte B of its unit?  (behind the aligned trigger, or refolded under ignore_case:
// the masked window compare itself)
template <int B>
__device__ __forceinline__ uint32_t cls_any_at(const uint32_t (&d)[8], const PatternDev& P) {
  const bool lo = (win_at<B>(d) & P.m0) == P.p0, hi = (win_at<B + 4>(d) & P.m1) == P.p1;
  return __ballot(lo && hi) != 0 ? 1u << B : 0u;
}
// The candidates at the positions of `pm` (bit b: some lane of the wave has a candidate at byte b of its unit), decided
// in registers.  One loop iteration per such position, b wave-uniform: the 8 bytes at byte b of the lane's 24-byte view
// are picked with scalar-controlled selects (no array indexed by a register, nothing for the compiler to put in
// scratch) and aligned with v_alignbyte; a lane is a candidate there iff the masked window compare passes, and only
// the positions that compare does not decide exactly (PatternDev::cls_chk) are looked up in the 256-bit sets.
__device__ __forceinline__ uint32_t cls_verify_positions(uint32_t pm, const uint32_t (&d)[8], const PatternDev& P,
                                                         const uint32_t* sets) {
  uint32_t m = 0;
  for (pm = __builtin_amdgcn_readfirstlane(pm); pm; pm &= pm - 1u) {  // scalar loop
    const uint32_t b = (uint32_t)__builtin_ctz(pm);
    const uint32_t q = b >> 2, r = b & 3u;
    const uint32_t x0 = q == 0 ? d[0] : q == 1 ? d[1] : q == 2 ? d[2] : d[3];
    const uint32_t x1 = q == 0 ? d[1] : q == 1 ? d[2] : q == 2 ? d[3] : d[4];
    const uint32_t x2 = q == 0 ? d[2] : q == 1 ? d[3] : q == 2 ? d[4] : d[5];
    const uint32_t wlo = __builtin_amdgcn_alignbyte(x1, x0, r);  // r == 0: x0
    const uint32_t whi = __builtin_amdgcn_alignbyte(x2, x1, r);
    const uint32_t cand = ((wlo & P.m0) == P.p0 && (whi & P.m1) == P.p1) ? 1u : 0u;
    const uint64_t W = ((uint64_t)whi << 32) | wlo;
    uint32_t any = 0;
    for (uint32_t a = 0; a < P.nalt; ++a) {  // scalar loop; one alternative must accept every position
      uint32_t ok = 1u;
      const uint32_t* sa = sets + a * P.plen * 8u;
      for (uint32_t chk = P.cls_chk; chk; chk &= chk - 1u) {  // scalar loop over the undecided positions
        const uint32_t k = (uint32_t)__builtin_ctz(chk);
        const uint32_t x = (uint32_t)(W >> (8u * k)) & 0xffu;
        ok &= sa[k * 8u + (x >> 5)] >> (x & 31u);
      }
      any |= ok;
    }
    m |= (any & cand & 1u) << b;
  }
  return m;
}

// per-wave running state of k_scan
struct WaveState {
  uint32_t cnt = 0;       // matches found by this lane
  uint32_t nlc = 0;       // newlines counted by this lane (WANT_NL)
  uint32_t last_rel = 0;  // end of this lane's last match, relative to the TILE start (> 0 once there is one; < 2^16)
  uint32_t wsum = 0;      // line summary of the wave span so far (lane 0); 0 is the identity
  bool run_nl = false;    // the current run of match-less loads holds a newline (wave-uniform)
  uint32_t masks[4] = {0, 0, 0, 0};
  uint32_t hi = 0;        // OR of the lane's bytes (ascii_only expressions: bit 7 of any byte set = non-ASCII data)
  // count_lines of a 1..3-byte needle (lines_flag_step): lines are counted where their closing newline stands
  uint32_t lacc = 0;      // newline-closed lines with a match counted by this lane
  uint32_t lsacc = 0;     // ... and on the scalar unit (a wave-load without a match start whose first newline closes one) (wave-uniform)
  uint32_t lcin = 0;      // the line that is open at this point of the span already holds a match (wave-uniform 0 / 1)
  unsigned long long lseen_m = 0;  // != 0: the span has shown a newline (wave-uniform; the OR of the wave-loads' newline ballots)
  unsigned long long lFm = 0;      // != 0: a match start before the span's first newline (wave-uniform)
  // kMask1: 0 = nobody asked for the number of matches (count_lines alone, ScanArgs::lines_only): `cnt` stays 0
  uint32_t count_on = 1;  // (wave-uniform)
  uint32_t track_last = 1;  // kMask1: last_rel is needed (the finish kernel walks the end of the chunk: plen > 1, lossy tail)
  // kMask1, counting: where the wave's last match ends is worked out ONCE, in the epilogue, from the flags of the last
  // wave-load that held a match (positions ascend with load, lane, byte: the highest lane of that load holds it).  Every
  // wave-load leaves its flags in its OWN registers (no copy: round 3 moved four registers per wave-load that held a
  // match) and its ballot on the scalar side; the epilogue picks the last load with a match.
  uint32_t lnf[4][4] = {{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
                        {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}};
  unsigned long long lMm[4] = {0, 0, 0, 0};  // their ballots (wave-uniform); all 0 = no match in the span
};

// ---------------------------------------------------------------------------
// count_lines for needles of 1..3 bytes (kMask1), which are in most lines of a text.  The unit stays in the byte-flag
// domain the matcher works in (0x80 per byte, the 16 bytes of the unit read as one 128-bit number, flags INVERTED: nf /
// nn are all ones except bit 7 of a byte where the pattern starts / that is a newline).
//
// Round 4: a line is counted where its CLOSING NEWLINE stands, by one 128-bit addition.  Read nn as 16 digits: a plain
// byte is 0xff, a newline 0x7f.  Add M = ~nf (0x80 at every match start) and a carry-in "the line that is open when
// the unit begins already holds a match":
//     plain byte   0xff + carry        -> passes a carry on                       (the open line still holds its match)
//     match start  0xff + 0x80 + carry -> always carries out                      (now it holds one)
//     newline      0x7f + carry        -> absorbs it; bit 7 of the digit = carry  (this line held a match: counted)
// so lines closed in the unit = popcount(S & ~nn) with S = nn + ~nf + cin = nn - nf - (1 - cin): four v_subb_co_u32,
// the carry-in as the first borrow-in, and the carry out of the last byte is "the open line holds a match" for the
// next unit.  Across the 64 units of a wave-load that carry is the generate / propagate recurrence round 3 already
// solved with ONE 64-bit scalar add: generate = the chain's carry-out with no carry-in (a first pass of the four
// subtractions, results unused), propagate = no newline in the unit.  25 VALU instructions per unit behind the
// newline flags where the first-match formulation of round 3 (M & ~(E - B), a select for the segment start, three
// 64-bit compares for the generate bit) took 41 -- `the`: 464 -> ~350 VALU instructions per 4 KiB wave
// (profiles/r04_dense_variants.txt).  The wave's summary (F, L, C of xsg_linesum.h) falls out at the end of the span:
// T = lines closed with a match, F = a match before the span's first newline (the first of them), L = the final carry.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void lines_flag_step(const uint32_t (&nf)[4], const unsigned long long Mm, const uint32_t (&src)[8],
                                                uint32_t lane, WaveState& st) {
  // The line state is wave-uniform and lives on the scalar unit.  "Seen a newline" and "a match before the first newline"
  // are kept as 64-bit MASKS (non-zero = true), OR-ed together from ballots: a 0 / 1 flag made from a comparison
  // (`Nm != 0 ? 1 : 0`) is built by the compiler through v_cndmask / v_cmp / v_readfirstlane -- two to four VALU
  // instructions per use, per wave-load.
  const uint32_t lcin0 = st.lcin;
  const unsigned long long seen0 = st.lseen_m;
  if (Mm != 0 || lcin0 != 0 || seen0 == 0) {  // otherwise nothing here can change the state or the counts
    uint32_t nn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t y = src[q] ^ 0x0a0a0a0au;
      nn[q] = ((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu;
    }
    const unsigned long long Nm = __ballot((nn[0] & nn[1] & nn[2] & nn[3]) != 0xffffffffu);  // units with a newline
    if (seen0 == 0 && Mm != 0) {  // a match start before the span's first newline?  (once per span; scalar)
      unsigned long long fm = Mm;  // no newline yet: any match start
      if (Nm != 0) {
        const int i0 = __builtin_ctzll(Nm);
        fm = Mm & ((1ull << i0) - 1ull);  // the units before the one with the first newline
        uint32_t open = 0xffffffffu;      // all ones while no newline has been met inside that unit
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t fq = ~(uint32_t)__builtin_amdgcn_readlane((int)nf[q], i0);
          const uint32_t nq = ~(uint32_t)__builtin_amdgcn_readlane((int)nn[q], i0);
          const uint32_t below = (nq & (0u - nq)) - 1u;  // below the dword's first newline (all ones if it has none)
          fm |= (unsigned long long)(open & fq & below);
          open &= (uint32_t)(((unsigned long long)nq - 1ull) >> 32);  // all ones iff nq == 0, by arithmetic (no compare + select)
        }
      }
      st.lFm |= fm;
    }
    // One block of assembly, so that nothing between the two passes has to be "declared uniform" again (an asm result
    // counts as divergent whatever its constraint says, and every readfirstlane of a scalar costs a v_mov and a
    // v_readfirstlane).  It also serves a wave-load WITHOUT a match start: no unit generates, the carry-in runs up to the
    // first newline, which is counted if the open line held a match.
    //   pass 1  nn - nf - 1 (borrow-in set everywhere; differences unused): no borrow out = the unit carries out on its
    //           own = its last event is a match start: generate, G = ~vcc
    //   scalar  carries of a + b + cin with a = G | ~N (generate or propagate: no newline in the unit), b = G, through
    //           the scalar carry flag -- s_cmp sets it to cin, two s_addc_u32 thread it through, the second leaves the
    //           carry out of bit 63 = "the line open at the end of the wave-load holds a match"; C = sum ^ a ^ b = the
    //           carry INTO every lane's unit; vcc = ~C
    //   pass 2  nn - nf - (1 - cin): the sums with the true carry-in; popcount(S & ~nn) = lines closed here with a match
    //           (v_bcnt_u32_b32 adds to an accumulator: the lane's count is threaded through the four)
    uint32_t x0, x1, x2, x3, glo, ghi, alo, ahi, slo, shi, cout;
    asm("s_mov_b64 vcc, -1\n\t"
        "v_subb_co_u32_e32 %0, vcc, %12, %16, vcc\n\t"
        "v_subb_co_u32_e32 %0, vcc, %13, %17, vcc\n\t"
        "v_subb_co_u32_e32 %0, vcc, %14, %18, vcc\n\t"
        "v_subb_co_u32_e32 %0, vcc, %15, %19, vcc\n\t"
        "s_not_b32 %4, vcc_lo\n\t"
        "s_not_b32 %5, vcc_hi\n\t"
        "s_orn2_b32 %6, %4, %20\n\t"
        "s_orn2_b32 %7, %5, %21\n\t"
        "s_cmp_lg_u32 %22, 0\n\t"
        "s_addc_u32 %8, %6, %4\n\t"
        "s_addc_u32 %9, %7, %5\n\t"
        "s_cselect_b32 %10, 1, 0\n\t"
        "s_xor_b32 %8, %8, %6\n\t"
        "s_xor_b32 %9, %9, %7\n\t"
        "s_xnor_b32 vcc_lo, %8, %4\n\t"
        "s_xnor_b32 vcc_hi, %9, %5\n\t"
        "v_subb_co_u32_e32 %0, vcc, %12, %16, vcc\n\t"
        "v_subb_co_u32_e32 %1, vcc, %13, %17, vcc\n\t"
        "v_subb_co_u32_e32 %2, vcc, %14, %18, vcc\n\t"
        "v_subb_co_u32_e32 %3, vcc, %15, %19, vcc\n\t"
        "v_bitop3_b32 %0, %0, %12, %0 bitop3:0x30\n\t"
        "v_bitop3_b32 %1, %1, %13, %1 bitop3:0x30\n\t"
        "v_bitop3_b32 %2, %2, %14, %2 bitop3:0x30\n\t"
        "v_bitop3_b32 %3, %3, %15, %3 bitop3:0x30\n\t"
        "v_bcnt_u32_b32 %11, %0, %11\n\t"
        "v_bcnt_u32_b32 %11, %1, %11\n\t"
        "v_bcnt_u32_b32 %11, %2, %11\n\t"
        "v_bcnt_u32_b32 %11, %3, %11"
        : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&s"(glo), "=&s"(ghi), "=&s"(alo), "=&s"(ahi), "=&s"(slo), "=&s"(shi),
          "=&s"(cout), "+v"(st.lacc)
        : "v"(nn[0]), "v"(nn[1]), "v"(nn[2]), "v"(nn[3]), "v"(nf[0]), "v"(nf[1]), "v"(nf[2]), "v"(nf[3]), "s"((uint32_t)Nm),
          "s"((uint32_t)(Nm >> 32)), "s"(lcin0)
        : "vcc", "scc");
    st.lcin = (uint32_t)__builtin_amdgcn_readfirstlane((int)cout);
    st.lseen_m = seen0 | Nm;
  }
}

// kMask1: the inverted match flags of one unit (all ones except bit 7 of a byte where the pattern starts).  PL = the
// pattern's length where it is 1..3 (a template parameter since round 4: the dispatch on the length ran once per
// wave-load, a dozen scalar instructions and two or three taken branches each time), 0 = 4..8 bytes, read at run time:
// a needle of the window kinds that an earlier count found DENSE in this data (launch_scan re-routes it here).  Its hot
// filter would send every wave-load into the slow path; decided byte-parallel, every position costs the same whether it
// matches or not: (plen + 6) instructions per dword -- three shifted views shared by all pattern bytes, one
// (view ^ byte) | z per pattern byte, the zero-byte test -- and the scan stays bound by memory.
// z has a zero byte at byte i of dword q iff the pattern starts at position 4q+i; the zero-byte test stops before its
// final NOT (counting and the line arithmetic work on the complement as well, lines_flag_step).
template <int PL>
__device__ __forceinline__ void mask1_flags(const uint32_t (&d)[8], const PatternDev& P, uint32_t (&nf)[4]) {
  const uint32_t c0 = (P.p0 & 0xffu) * 0x01010101u;
  const uint32_t c1 = ((P.p0 >> 8) & 0xffu) * 0x01010101u;
  const uint32_t c2 = ((P.p0 >> 16) & 0xffu) * 0x01010101u;
  uint32_t z[4];
  if (PL == 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) z[q] = d[q] ^ c0;
  } else if (PL == 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q)  // (a ^ c) | b in one v_bitop3_b32 (0xde)
      z[q] = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_alignbyte(d[q + 1], d[q], 1), d[q] ^ c0, c1, 0xde);
  } else if (PL == 3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t u = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_alignbyte(d[q + 1], d[q], 1), d[q] ^ c0, c1, 0xde);
      z[q] = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_alignbyte(d[q + 1], d[q], 2), u, c2, 0xde);
    }
  } else if (PL == 4) {
    const uint32_t c3 = (P.p0 >> 24) * 0x01010101u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t u = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_alignbyte(d[q + 1], d[q], 1), d[q] ^ c0, c1, 0xde);
      u = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_alignbyte(d[q + 1], d[q], 2), u, c2, 0xde);
      z[q] = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_alignbyte(d[q + 1], d[q], 3), u, c3, 0xde);
    }
  } else {
    const uint32_t c3 = (P.p0 >> 24) * 0x01010101u;
    uint32_t a1[5], a2[5], a3[5];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a1[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 1);
      a2[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 2);
      a3[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 3);
      z[q] = (d[q] ^ c0) | (a1[q] ^ c1) | (a2[q] ^ c2) | (a3[q] ^ c3);
    }
    if (P.plen > 4) {
      asm volatile("");  // (keeps the branch a branch: as selects, a four-byte needle paid for an eight-byte one)
      const uint32_t c4 = (P.p1 & 0xffu) * 0x01010101u;
#pragma unroll
      for (int q = 0; q < 4; ++q) z[q] |= d[q + 1] ^ c4;
      if (P.plen > 5) {
        asm volatile("");
        const uint32_t c5 = ((P.p1 >> 8) & 0xffu) * 0x01010101u;
        a1[4] = __builtin_amdgcn_alignbyte(d[5], d[4], 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) z[q] |= a1[q + 1] ^ c5;
        if (P.plen > 6) {
          asm volatile("");
          const uint32_t c6 = ((P.p1 >> 16) & 0xffu) * 0x01010101u;
          a2[4] = __builtin_amdgcn_alignbyte(d[5], d[4], 2);
#pragma unroll
          for (int q = 0; q < 4; ++q) z[q] |= a2[q + 1] ^ c6;
          if (P.plen > 7) {
            asm volatile("");
            const uint32_t c7 = (P.p1 >> 24) * 0x01010101u;
            a3[4] = __builtin_amdgcn_alignbyte(d[5], d[4], 3);
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] |= a3[q + 1] ^ c7;
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) nf[q] = ((z[q] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z[q] | 0x7f7f7f7fu;
}

// One wave-load (1 KiB): `cur` is this lane's 16-byte unit, `nx` the unit that
// follows the wave-load (lane 0's unit of the next load, or the bytes after the
// span).  CAREFUL: the wave-load may reach past the end of the chunk.
// Returns the lane's exact match-start bits (also accumulated into `st`).
// PL: kMask1 only -- the pattern length where it is 1..4, 0 = read at run time (mask1_flags); j: the load's number in
// the wave's span (a constant once the caller's loop is unrolled: it names the registers the load's flags stay in).
template <int KIND, bool WANT_NL, bool WANT_LINES, bool EMIT, bool ICASE, bool CAREFUL, bool ALIGNED, int PL = 0>
__device__ __forceinline__ uint32_t scan_load(const uint4 cur, const uint4 nx, bool nx_is_vgpr, uint64_t unit_off,
                                              uint32_t unit_rel, uint32_t lane, uint64_t L, uint64_t limit,
                                              const PatternDev& P,
                                              const uint8_t* cbase, const uint8_t* s_pat, uint8_t* s_view,
                                              WaveState& st, const bool near_limit, const int j) {
  // ignore_case, patterns of 4+ bytes (LAZY): the hot filter does not need the exact fold.  (x | 0x20) == (p | 0x20)
  // holds for every byte x that folds to the pattern byte p (exactly those when p is a letter, one more byte value
  // otherwise), so the candidate test runs on data OR-ed with 0x20 -- one op per dword instead of fold4's seven --
  // against P.q0/q1, and only a wave-load with a candidate folds its bytes properly for the exact decision.
  // Patterns of 1..3 bytes (byte-parallel exact path, usually dense) fold up front: the neighbour's bytes then
  // arrive already folded through the DPP exchange and the wave-uniform edge values fold on the scalar unit.
  // '\n' is not a letter, but OR-ing changes it: the newline tests read the raw bytes (`r`).
  constexpr bool LAZY = ICASE && KIND != kMask1;
  constexpr uint32_t k20 = 0x20202020u;
  uint32_t r[8] = {cur.x, cur.y, cur.z, cur.w, 0u, 0u, 0u, 0u};
  uint32_t d[8] = {LAZY ? (cur.x | k20) : cur.x, LAZY ? (cur.y | k20) : cur.y, LAZY ? (cur.z | k20) : cur.z, LAZY ? (cur.w | k20) : cur.w,
                   0u, 0u, 0u, 0u};
  // The byte-parallel path under ignore_case: a needle of letters only (PatternDev::lazy_exact; the usual case) needs no
  // fold either -- (x | 0x20) == p decides it exactly, one instruction per dword where fold4 takes seven.
  constexpr bool FOLD1 = ICASE && KIND == kMask1;
  const bool orfold = FOLD1 && P.lazy_exact;  // (wave-uniform)
  if (FOLD1) {
    if (orfold) {
      asm volatile("");
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] |= k20;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = fold4(d[q]);
    }
  }
  const uint32_t own0 = d[0], own1 = d[1];  // what the left neighbour reads (never cleared)
  if (CAREFUL) {
    // bytes at or beyond L are not part of the chunk: clear them once
    if (unit_off + kUnit > L) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint64_t o = unit_off + 4u * q;
        const uint32_t keep = o >= L ? 0u : (o + 4u > L ? ((1u << (8u * (uint32_t)(L - o))) - 1u) : 0xffffffffu);
        d[q] &= keep;
        r[q] &= keep;
      }
    }
  }
  // the neighbour's first 8 bytes: lane+1's unit, lane 63 takes lane 0 of the next load / the edge
  const uint32_t e0r = nx_is_vgpr ? __builtin_amdgcn_readfirstlane(nx.x) : nx.x;
  const uint32_t e1r = nx_is_vgpr ? __builtin_amdgcn_readfirstlane(nx.y) : nx.y;
  const uint32_t e0 = (LAZY || orfold) ? (e0r | k20) : ICASE ? fold4(e0r) : e0r;
  const uint32_t e1 = (LAZY || orfold) ? (e1r | k20) : ICASE ? fold4(e1r) : e1r;
  d[4] = from_next_lane(own0, e0, lane);
  d[5] = from_next_lane(own1, e1, lane);
  const uint32_t(&nlsrc)[8] = ICASE ? r : d;  // own bytes as the newline tests must see them (fold4 leaves '\n' alone, OR-ing does not)
  if (is_cls(KIND)) {
    if (P.ascii_only) st.hi |= r[0] | r[1] | r[2] | r[3];  // own bytes (beyond the chunk end: cleared); folding keeps bit 7
  }

  if (WANT_NL) st.nlc += nl_count16(nlsrc);

  uint32_t m = 0;
  if (KIND == kMask1) {
    // plen 1..3 (and 4..8-byte needles found dense, launch_scan: dense_bytes_route) -- usually dense in text, so there
    // is no cheap "nothing here" case to filter for: decide all 16 positions byte-parallel instead.  z has a zero byte at
    // byte i of dword q iff the pattern starts at position 4q+i; fl[q] flags exactly
    // those bytes with 0x80 (5 to 11 ops per dword instead of ~24 for windows + masks).
    // nf: the flags INVERTED -- all ones except bit 7 of the byte where the pattern starts
    uint32_t nf[4];
    mask1_flags<PL>(d, P, nf);
    bool has = (nf[0] & nf[1] & nf[2] & nf[3]) != 0xffffffffu;
    unsigned long long Mm = __ballot(has);
    // positions at or beyond the limit belong to the tail walk.  A span that ends below the limit (wave-uniform; all
    // but a chunk's last) does not look.
    if (CAREFUL && near_limit && Mm != 0) {  // (the fast body only runs spans that end below the limit: scan_tile)
      asm volatile("");  // (a branch, not a mask on the per-lane test)
      if (unit_off + kUnit > limit) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint64_t o = unit_off + 4u * q;
          nf[q] = o >= limit ? 0xffffffffu : (o + 4u > limit ? nf[q] | ~((1u << (8u * (uint32_t)(limit - o))) - 1u) : nf[q]);
        }
        has = (nf[0] & nf[1] & nf[2] & nf[3]) != 0xffffffffu;
      }
      Mm = __ballot(has);
    }
    if (EMIT) {
      if (Mm != 0) {
        // the emit pass needs the position bits: 0x80-per-byte flags -> one bit per position
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t f = ~nf[q] >> 7;  // bits 0, 8, 16, 24
          m |= ((f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u)) << (4 * q);
        }
      }
    } else {
      if (Mm != 0) {
        // counting needs no bit mask at all: popcount(nf) = 32 - matches of the dword, chained through v_bcnt's accumulator
        if (!(WANT_LINES && !EMIT)) {
          const uint32_t n = 128u - ((uint32_t)__popc(nf[0]) + (uint32_t)__popc(nf[1]) + (uint32_t)__popc(nf[2]) + (uint32_t)__popc(nf[3]));
          st.cnt += n;
        }
      }
      // where the last match ends: only the end-of-chunk walk asks (a one-byte pattern has none: strchr is exact
      // everywhere, xsg_tail.h).  The flags stay where they are, the ballot goes to the scalar side: WaveState::lnf
      st.lnf[j][0] = nf[0], st.lnf[j][1] = nf[1], st.lnf[j][2] = nf[2], st.lnf[j][3] = nf[3];
      st.lMm[j] = Mm;
      if (WANT_LINES) lines_flag_step(nf, Mm, nlsrc, lane, st);
      return 0;
    }
  } else {
    PatternDev Pf = P;  // what the hot filter compares against
    if (LAZY) {
      Pf.p0 = P.q0;
      Pf.p1 = P.q1;
    }
    // 8-byte-window kinds: two hot filters, one per kernel instantiation.  ALIGNED: the aligned-dword trigger (25
    // compares; 4 of the 8 window bytes for three of the four alignments) straight in front of the exact slow
    // path -- half the VALU work of the window filter, the right choice whenever the window's 4-byte pieces are
    // rare in the text.  Otherwise the window filter proper (20 unaligned windows x 2 compares).  Which one runs
    // is decided per shard and pattern by measurement (xsg_api.cpp: choose_hot_filter); running both as a cascade
    // was measured slower than either (the straight-line body outgrows the instruction cache).
    constexpr bool kTrigger = ALIGNED && (KIND == kTwo || KIND == kLong || KIND == kClass);  // (kClassFast: never ALIGNED)
    // The window filter's 16 compare results are kept (as lane masks on the scalar side) and become the candidate
    // bits of the slow path: a needle that is in most wave-loads (`that`) used to build its windows and compare them
    // twice.  Not under LAZY (the slow path compares properly folded bytes) nor behind the aligned trigger (which
    // has no per-position results).
    constexpr bool kReuse = !kTrigger && !LAZY;
    // (Round 2 also tried the class sequences' masked compares in two stages with a wave-wide exit in between -- the
    // high dword of the window for all 16 positions, the low dword only if some lane passed -- to save the second
    // 32 operations in wave-loads that hold no piece of the window: slower everywhere on the bench corpus, `She[r ]lock`
    // 2.77 against 3.87 TB/s, `[Ss]herlock` 3.4 against 5.6 on the same box -- as with the trigger cascade, straight-
    // line code beats a cheaper expected path with a ballot and a branch in it.  Removed.)
    bool cb[16];
    bool any_c;
    unsigned long long cmk[16];        // cls_fast: the filter's results as lane masks
    unsigned long long fast_any = 0;   // ... and their OR
    if (kTrigger) {
      any_c = trigger_aligned<KIND>(d, Pf);
    } else if (KIND == kClassFast) {
      // A class sequence whose window pins its upper four positions and its first two completely (`She[r ]lock`:
      // "Sh", then a position and a class, then "lock"): the filter drops the two positions in between -- a superset,
      // every candidate is verified against all sets anyway -- and becomes two EXACT compares per position, a 16-bit
      // and a 32-bit one, no `and`: 47 operations per 16 bytes like a literal of 8 bytes, instead of 79.  The
      // compiler widens a 16-bit equality to and + compare, so the compare is spelled out; its results are lane
      // masks in scalar registers and stay there (combined on the scalar unit) until a wave-load enters the slow path.
      uint32_t w[20];
      windows<20>(d, w);
      fast_any = 0;
      const uint32_t lo16 = Pf.p0 & 0xffffu;
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        unsigned long long a16;
        asm("v_cmp_eq_u16_e64 %0, %1, %2" : "=s"(a16) : "v"(w[b]), "s"(lo16));
        cmk[b] = a16 & __builtin_amdgcn_uicmp(w[b + 4], Pf.p1, 32 /* eq */);
        fast_any |= cmk[b];
      }
      any_c = false;
    } else {
      uint32_t w[20];
      windows<20>(d, w);
      any_c = false;
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        cb[b] = cand_at<KIND>(w, b, Pf);
        any_c |= cb[b];
      }
    }
    if (KIND == kClassFast ? fast_any != 0 : __ballot(any_c) != 0) {
      // ignore_case with a window of letters only: the hot filter's verdict on (data | 0x20) is already the exact
      // one (PatternDev::lazy_exact) -- no refold, and its compare results stand like those of a case-sensitive search
      const bool lazy_done = LAZY && !kTrigger && !is_cls(KIND) && P.lazy_exact;
      if (LAZY && !lazy_done) {  // now the exact view: properly folded bytes, own and neighbour's
        const uint32_t f0 = fold4(cur.x), f1 = fold4(cur.y);
        d[0] = fold4(r[0]), d[1] = fold4(r[1]), d[2] = fold4(r[2]), d[3] = fold4(r[3]);  // r is already cleared
        d[4] = from_next_lane(f0, fold4(e0r), lane);
        d[5] = from_next_lane(f1, fold4(e1r), lane);
      }
      if (is_cls(KIND) && P.cls_inreg) {
        // expressions of up to 8 positions: candidates decided in registers (d is the exact view by now).  Which of
        // the 16 positions hold a candidate in SOME lane is wave-uniform knowledge: the hot filter's lane masks, or --
        // behind the aligned trigger / under ignore_case -- the window compare run here.
        const uint32_t* sets = reinterpret_cast<const uint32_t*>(s_pat);
        uint32_t pm = 0;
        if (KIND == kClassFast && kReuse) {
#pragma unroll
          for (int b = 0; b < 16; ++b) pm |= cmk[b] != 0 ? 1u << b : 0u;
        } else if (kReuse) {
#pragma unroll
          for (int b = 0; b < 16; ++b) pm |= __ballot(cb[b]) != 0 ? 1u << b : 0u;
        } else {
          pm = cls_any_at<0>(d, P) | cls_any_at<1>(d, P) | cls_any_at<2>(d, P) | cls_any_at<3>(d, P) | cls_any_at<4>(d, P) |
               cls_any_at<5>(d, P) | cls_any_at<6>(d, P) | cls_any_at<7>(d, P) | cls_any_at<8>(d, P) | cls_any_at<9>(d, P) |
               cls_any_at<10>(d, P) | cls_any_at<11>(d, P) | cls_any_at<12>(d, P) | cls_any_at<13>(d, P) |
               cls_any_at<14>(d, P) | cls_any_at<15>(d, P);
        }
        m = cls_verify_positions(pm, d, P, sets);
        // positions at or beyond the limit belong to the end-of-chunk walk (koff = 0 here)
        if (unit_off >= limit)
          m = 0;
        else if (unit_off + kUnit > limit)
          m &= (1u << (uint32_t)(limit - unit_off)) - 1u;
      } else {
      if (is_cls(KIND)) {
        // class sequences verify their candidates in the lane's view: the rest of the neighbour's unit joins it
        // (raw own bytes go out -- a lane's own view of them may be cleared at the chunk end, the reader's not)
        const uint32_t e2r = nx_is_vgpr ? __builtin_amdgcn_readfirstlane(nx.z) : nx.z;
        const uint32_t e3r = nx_is_vgpr ? __builtin_amdgcn_readfirstlane(nx.w) : nx.w;
        d[6] = from_next_lane(ICASE ? fold4(cur.z) : cur.z, ICASE ? fold4(e2r) : e2r, lane);
        d[7] = from_next_lane(ICASE ? fold4(cur.w) : cur.w, ICASE ? fold4(e3r) : e3r, lane);
      }
      if (KIND == kClassFast && kReuse) {
        uint32_t m0 = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) {  // this lane's bits of the masks
          uint32_t t;
          asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(t) : "s"(cmk[b]));
          m0 |= t << b;
        }
        m = match_mask16_from<KIND, ICASE>(m0, d, P, cbase, unit_off, limit, s_pat, s_view);
      } else if (kReuse || lazy_done) {
        uint32_t m0 = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) m0 |= (uint32_t)cb[b] << b;
        m = match_mask16_from<KIND, ICASE>(m0, d, P, cbase, unit_off, limit, s_pat, s_view);
      } else {
        m = match_mask16<KIND, ICASE>(d, P, cbase, unit_off, limit, s_pat, s_view);
      }
      }
    }
  }
  if (EMIT) {
    st.cnt += (uint32_t)__popc(m);
  } else {
    if (m) {
      st.cnt += (uint32_t)__popc(m);
      st.last_rel = unit_rel + (31u - (uint32_t)__clz(m)) + P.plen - (KIND >= kLong ? P.koff : 0u);  // > 0: plen > koff
    }
    if (WANT_LINES) {
      // A wave-load without any match start (the common case) summarises to "has a
      // newline or not", and a run of such loads to the OR of that: once a newline
      // has been seen in the run, the remaining loads of the run need no test at all.
      if (__ballot(m != 0) != 0) {
        if (st.run_nl) st.wsum = sum_combine(st.wsum, kSumNl);
        st.run_nl = false;
        st.wsum = sum_combine(st.wsum, wave_units_combine(sum_of_unit(m, nl_mask16(nlsrc))));
      } else if (!st.run_nl) {
        st.run_nl = __ballot(nl_any16(nlsrc)) != 0;
      }
    }
  }
  return m;
}

// ---------------------------------------------------------------------------
// k_scan: the bulk pass.  EMIT=false: per-tile counts / newline counts / line
// summaries.  EMIT=true: the same decisions, writing every match offset at its
// rank (tile_off[tile] + rank inside the tile).
// ---------------------------------------------------------------------------
template <int KIND, bool WANT_NL, bool WANT_LINES, bool EMIT, int LOADS, bool ICASE, bool ALIGNED>
__device__ __forceinline__ void scan_tile(const ScanArgs& A, const uint64_t tile) {
  constexpr int kLoads = LOADS;                          // 16-byte units per lane
  constexpr uint32_t kWaveSpan = kWaveLoad * kLoads;     // contiguous bytes per wave
  constexpr uint32_t kTile = kWaveSpan * kWaves;         // bytes per workgroup
  __shared__ uint32_t s_cnt[kWaves];
  __shared__ uint32_t s_nl[kWaves];
  __shared__ __attribute__((aligned(16))) uint32_t s_ep[KIND == kMask1 ? kWaves : 1][4];  // kMask1's epilogue: one writer per tile
  __shared__ __attribute__((aligned(16))) uint8_t s_pat[is_cls(KIND) ? 2048 : KIND == kLong ? kLdsPattern : 16];
  __shared__ __attribute__((aligned(16))) uint8_t s_view[is_cls(KIND) ? kBlock * 48 : 16];  // match_mask16<kClass>

  if (tile >= A.ntiles) return;
  // (a tile from the hit list holds a match by construction: no look at its count -- the emit pass over a sparse list
  // is a chain of dependent memory round trips per tile, 2048 tiles resident at a time, and this was one of them)
  if (EMIT && !A.hit_tiles && A.tile_cnt[tile] == 0) return;

  const PatternDev P = A.pat;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps wbase and the branches on it scalar

  if (KIND == kLong) {  // the first KiB of the pattern; a candidate of a longer one is verified against the device copy beyond it
    for (uint32_t k = tid; k < P.plen && k < kLdsPattern; k += kBlock) s_pat[k] = P.d_pat[k];
    __syncthreads();
  }

  const uint32_t c = A.tile_chunk ? A.tile_chunk[tile] : 0u;
  const ChunkDev ch = A.chunks[c];
  const uint8_t* cbase = A.base + ch.offset;
  const uint64_t L = ch.length;
  const uint64_t Lr = (L + 15u) & ~(uint64_t)15u;
  // positions o < limit are decided here; [limit, L) belongs to the tail walk
  const uint64_t limit =
      P.exact_tail ? (L >= P.plen ? L - P.plen + 1u : 0u) : tail_zone_begin(L, P.plen);
  const uint64_t toff = (tile - A.chunk_tile0[c]) * (uint64_t)kTile;
  const uint64_t wbase = toff + (uint64_t)wave * kWaveSpan;

  // ---- issue all loads of the wave span up front (kLoads KiB in flight per wave).
  // No branch around a load: units past the end of the chunk re-read its last
  // unit (an L2 hit) and are cleared below, so the loads stay back to back.
  // Stagger the load bursts of the four waves of a workgroup: wave w waits
  // w * stagger * 64 clocks before it issues its loads, so a workgroup pulls its
  // 16 KiB as four 4 KiB bursts one after the other instead of all at once.
  // Measured on the 50 GiB shard, plain count (scripts/tune_sweep.py): 7.07 TB/s
  // without, 7.22 / 7.35 / 7.46 / 7.41 / 7.23 TB/s at stagger 9 / 12 / 14 / 16 / 20,
  // 6.56 at 32.  launch_scan picks the value per kernel variant (0 for VALU-bound ones).
  {
    const uint32_t n = (A.tune & 0xffu) * wave;
    for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
  }
  const uint64_t last_unit = Lr - kUnit;  // L >= 1 here: a chunk of length 0 has no tiles
  uint4 v[kLoads];
  uint4 edge;
  // Emit pass with wave marks (the one-sync list route): the count pass left one bit per wave that found something,
  // and a wave without its bit has nothing to emit -- it reads nothing (a sparse list re-reads 4 KiB per match
  // instead of the tile's 16: the emit pass over 7 000 tiles of a 10 GiB shard is a 118 MB random read otherwise) and
  // contributes 0 to the ranks.  A stale bit of an older pass only costs its 4 KiB.
  bool wave_on = true;
  uint32_t wm_word = 0;
  if (EMIT && A.tile_wmask) {
    wm_word = A.tile_wmask[tile >> 2];
    wave_on = ((wm_word >> (((uint32_t)tile & 3u) * 8u + wave)) & 1u) != 0;
  }
  // non-temporal: every byte is read once, so keeping it out of L2/MALL allocation
  // is worth +8 % on this stream (7.1 vs 6.55 TB/s, scripts/read_variants.py)
  if (EMIT && !wave_on) {
#pragma unroll
    for (int j = 0; j < kLoads; ++j) v[j] = make_uint4(0u, 0u, 0u, 0u);
    edge = make_uint4(0u, 0u, 0u, 0u);
  } else if (wbase + kWaveSpan + kUnit <= Lr) {  // wave-uniform: span and edge inside the chunk -> no clamping
    const uint8_t* p0 = cbase + wbase + (uint64_t)lane * kUnit;
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
      const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p0 + (uint64_t)j * kWaveLoad));
      v[j] = make_uint4(t.x, t.y, t.z, t.w);
    }
    // the 16 bytes that follow the span (wave-uniform address)
    edge = *reinterpret_cast<const uint4*>(cbase + wbase + kWaveSpan);
  } else {
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
      uint64_t off = wbase + (uint64_t)j * kWaveLoad + (uint64_t)lane * kUnit;
      off = off < last_unit ? off : last_unit;
      const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(cbase + off));
      v[j] = make_uint4(t.x, t.y, t.z, t.w);
    }
    uint64_t eoff = wbase + kWaveSpan;
    eoff = eoff < last_unit ? eoff : last_unit;
    edge = *reinterpret_cast<const uint4*>(cbase + eoff);
  }
  // keep every load ahead of the first use of any of them (otherwise the
  // scheduler sinks a copy of load 0 between the loads and stalls the issue)
  __builtin_amdgcn_sched_barrier(0);
  if (is_cls(KIND)) {
    // The sets (256 bits per alternative and position, at most 64 sets = 2 KiB) go to LDS BEHIND the tile's loads:
    // only the slow path reads them, and staged first -- a load, a store and a barrier in front of everything --
    // they held back the moment the workgroup's 16 KiB are requested.
    for (uint32_t k = tid; k < P.plen * P.nalt * 8u; k += kBlock)
      reinterpret_cast<uint32_t*>(s_pat)[k] = reinterpret_cast<const uint32_t*>(P.d_pat)[k];
    __syncthreads();
  }

  // The loop body exists twice: a wave whose whole span lies inside the chunk (all
  // but the last tile of a chunk) skips every end-of-chunk check.
  // (The emit pass has only the careful body: it visits few tiles, and what it costs there is the FETCH of its code --
  // 23 KB with both bodies -- by every compute unit that gets a workgroup, not the end-of-chunk checks.)
  WaveState st;
  if (KIND == kMask1) {
    st.track_last = (!P.exact_tail && P.plen > 1) ? 1u : 0u;
    // (xs::count_lines never asks for the number of matches: ScanArgs::lines_only is set by every caller of the WANT_LINES
    // variant of this kind, so the instantiation does not look at it wave-load by wave-load)
    st.count_on = (WANT_LINES && !EMIT) ? 0u : 1u;
  }
  const bool near_limit = wbase + kWaveSpan > limit;  // wave-uniform: only a chunk's last spans reach the tail walk's zone
  // the wave's loads, one after the other; PL: see scan_load (kMask1: the dispatch on the pattern length sits OUTSIDE the loop)
  auto run_loads = [&](auto careful_tag, auto pl_tag) {
    constexpr bool kCareful = decltype(careful_tag)::value;
    constexpr int kPl = decltype(pl_tag)::value;
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
      const uint4 nx = j + 1 < kLoads ? v[j + 1 < kLoads ? j + 1 : j] : edge;
      st.masks[j < 4 ? j : 0] = scan_load<KIND, WANT_NL, WANT_LINES, EMIT, ICASE, kCareful, ALIGNED, kPl>(
          v[j], nx, j + 1 < kLoads, wbase + (uint64_t)j * kWaveLoad + (uint64_t)lane * kUnit,
          wave * kWaveSpan + (uint32_t)j * kWaveLoad + lane * kUnit, lane, L, limit, P, cbase, s_pat, s_view, st, near_limit, j);
    }
  };
  auto run_body = [&](auto careful_tag) {
    if (KIND == kMask1) {
      const uint32_t pl = __builtin_amdgcn_readfirstlane(P.plen);
      if (pl == 1) run_loads(careful_tag, std::integral_constant<int, 1>{});
      else if (pl == 2) run_loads(careful_tag, std::integral_constant<int, 2>{});
      else if (pl == 3) run_loads(careful_tag, std::integral_constant<int, 3>{});
      else if (pl == 4) run_loads(careful_tag, std::integral_constant<int, 4>{});
      else run_loads(careful_tag, std::integral_constant<int, 0>{});
    } else {
      run_loads(careful_tag, std::integral_constant<int, 0>{});
    }
  };
  if (EMIT && !wave_on) {
    // nothing to decide
  } else if (!EMIT && wbase + kWaveSpan <= L && !(KIND == kMask1 && near_limit)) {
    // the fast body: the whole span lies inside the chunk -- and, for the byte-parallel kinds, below the limit beyond which
    // positions belong to the end-of-chunk walk, so that it tests neither (the other kinds apply the limit in their slow path)
    run_body(std::false_type{});
  } else {
    run_body(std::true_type{});
  }
  const uint32_t cnt = st.cnt, nlc = st.nlc;
  uint32_t wsum = st.wsum;
  const bool run_nl = st.run_nl;
  uint32_t masks[kLoads];
#pragma unroll
  for (int j = 0; j < kLoads; ++j) masks[j] = st.masks[j < 4 ? j : 0];
  if (WANT_LINES && !EMIT && run_nl) wsum = sum_combine(wsum, kSumNl);
  if (WANT_LINES && !EMIT && KIND == kMask1) {
    // lines_flag_step's state -> the span's summary (xsg_linesum.h): T lines closed by a newline of the span hold a match;
    // the first of them is F (its match lies before the span's first newline), the rest are C; L = the final carry: the
    // line that is open at the end of the span holds a match.  Without a newline in the span nothing was closed and that
    // carry says whether the span holds a match at all.
    if (st.lseen_m == 0) {
      const uint32_t f = st.lcin;
      wsum = (f << 1) | (f << 2);
    } else {
      const uint32_t lF = st.lFm != 0 ? 1u : 0u;
      const uint32_t T = (__any(st.lacc != 0) ? wave_sum_u32(st.lacc) : 0u) + st.lsacc;
      wsum = kSumNl | (lF << 1) | (st.lcin << 2) | ((T - lF) << kSumCShift);
    }
  }

  if (is_cls(KIND)) {
    if (P.ascii_only && __any((st.hi & 0x80808080u) != 0) && lane == 0) atomicOr(A.flags, 1u);  // the search must refuse
  }
  if (!EMIT) {
    // ---- epilogue.  A store per tile, however small, interleaves writes into the
    // read stream (DRAM bus turnarounds): a probe kernel with k_scan's loads lost
    // 2-4 % to it (scripts/probe_parts.py).  So the per-tile arrays hold "nothing
    // found" at rest (tile_cnt = 0, tile_sum = 0 = "has a newline, no match";
    // k_count_finish puts that back as it consumes them) and only waves that
    // found something else write: no LDS, no barrier, no store on the common path.
    bool wave_has;
    const unsigned long long any_m = KIND == kMask1 ? (st.lMm[0] | st.lMm[1] | st.lMm[2] | st.lMm[3]) : 0ull;
    if (KIND == kMask1) wave_has = st.count_on ? __any(cnt != 0) : any_m != 0;  // (lines only, no end-of-chunk walk: nothing to report)
    else wave_has = __any(cnt != 0);
    if (KIND == kMask1) {
      // The byte-parallel kind is what DENSE needles run on (1..3 bytes, and 4..8-byte ones found dense): every wave has
      // something to report, and four waves x three atomics per 16 KiB of text were 2.5-4 % of the scan (20 GiB, `e` / `the` /
      // `that`: 0.857 of peak, 0.885 with the atomics taken out, 0.88 with one wave's only -- profiles/r04_dense_variants.txt).
      // So the waves meet in LDS and ONE lane writes the tile's words with plain stores: the tile count (at rest 0, one writer),
      // the wave marks as the tile's byte of tile_wmask, the end of the last match with this pass's tag, the four line
      // summaries as one 16-byte store, the newline count.
      uint32_t wc = 0, rel = 0;
      if (wave_has) {
        wc = wave_sum_u32(cnt);
        rel = 1u;  // no end-of-chunk walk reads it (it must stay inside the tile: tile_last's tag)
        if (st.track_last && any_m != 0) {  // scalar: the highest lane of the last wave-load with a match
          bool found = false;
#pragma unroll
          for (int j = 3; j >= 0; --j) {
            if (j < kLoads && !found && st.lMm[j] != 0) {
              found = true;
              const int l = 63 - __builtin_clzll(st.lMm[j]);
              const uint32_t f3 = ~(uint32_t)__builtin_amdgcn_readlane((int)st.lnf[j][3], l), f2 = ~(uint32_t)__builtin_amdgcn_readlane((int)st.lnf[j][2], l);
              const uint32_t f1 = ~(uint32_t)__builtin_amdgcn_readlane((int)st.lnf[j][1], l), f0 = ~(uint32_t)__builtin_amdgcn_readlane((int)st.lnf[j][0], l);
              const uint32_t hq = f3 ? 3u : f2 ? 2u : f1 ? 1u : 0u;
              const uint32_t hf = f3 ? f3 : f2 ? f2 : f1 ? f1 : f0;
              rel = wave * kWaveSpan + (uint32_t)j * kWaveLoad + (uint32_t)l * kUnit + 4u * hq + ((31u - (uint32_t)__builtin_clz(hf)) >> 3) + P.plen;
            }
          }
        }
      }
      uint32_t wn = 0;
      if (WANT_NL) wn = wave_sum_u32(nlc);
      if (lane == 0) {
        s_ep[wave][0] = wc;
        s_ep[wave][1] = rel;  // 0: this wave reports nothing
        s_ep[wave][2] = WANT_LINES ? (wsum ^ kSumNl) : 0u;  // 0 = "a newline, no match"
        s_ep[wave][3] = wn;
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t tc = 0, tl = 0, marks = 0, tn = 0;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) {
          tc += s_ep[w][0];
          tl = s_ep[w][1] > tl ? s_ep[w][1] : tl;
          marks |= s_ep[w][1] ? 1u << w : 0u;
          tn += s_ep[w][3];
        }
        if (marks) {
          if (tc) A.tile_cnt[tile] = tc;
          if (A.tile_wmask) reinterpret_cast<uint8_t*>(A.tile_wmask)[tile] = (uint8_t)marks;  // for the emit pass (the tile's byte of its word)
          // end of the last match relative to the tile start (1 .. tile + plen < 2^16), tagged with this pass's epoch
          A.tile_last[tile] = (A.epoch << 16) | tl;
        }
        if (WANT_LINES) {
          const uint4 q = make_uint4(s_ep[0][2], s_ep[1][2], s_ep[2][2], s_ep[3][2]);
          if (q.x | q.y | q.z | q.w) *reinterpret_cast<uint4*>(A.tile_sum + tile * kWaves) = q;
        }
        if (WANT_NL) A.tile_nl[tile] = tn;
      }
    } else {
      if (wave_has) {
        const uint32_t wc = wave_sum_u32(cnt);
        // the end of the wave's last match, relative to the tile start: it fits 16 bits, so the reduction runs on
        // 32-bit values (a 64-bit max costs three times the lane exchanges; a needle that is dense in the text pays
        // this epilogue in every wave)
        const uint32_t rel = wave_max_u32(st.last_rel);
        if (lane == 0) {
          // per-TILE words: 4-way contention at most (a per-chunk max was 4000-way and
          // cut dense patterns to a third)
          atomicAdd(A.tile_cnt + tile, wc);
          if (A.tile_wmask) atomicOr(A.tile_wmask + (tile >> 2), 1u << (((uint32_t)tile & 3u) * 8u + wave));  // for the emit pass
          // end of the last match relative to the tile start (1 .. tile + plen < 2^16), tagged with this pass's
          // epoch: a word of an older pass loses the max, so the array is never reset
          atomicMax(A.tile_last + tile, (A.epoch << 16) | rel);
        }
      }
      if (WANT_LINES) {
        if (lane == 0 && wsum != kSumNl) A.tile_sum[tile * kWaves + wave] = wsum ^ kSumNl;  // 0 = "a newline, no match"
      }
      if (WANT_NL) {  // every wave has newlines to report: one store per tile through LDS
        const uint32_t wn = wave_sum_u32(nlc);
        if (lane == 0) s_nl[wave] = wn;
        __syncthreads();
        if (tid == 0) A.tile_nl[tile] = s_nl[0] + s_nl[1] + s_nl[2] + s_nl[3];
      }
    }
  } else {
    // ---- ordered emission: wave spans are consecutive, loads within a span
    // are consecutive, lanes within a load are consecutive, bits ascend.
    const uint32_t wc = wave_sum_u32(cnt);
    if (lane == 0) s_cnt[wave] = wc;
    __syncthreads();
    if (A.tile_wmask && tid == 0 && ((wm_word >> (((uint32_t)tile & 3u) * 8u)) & 0xffu) != 0)
      atomicAnd(A.tile_wmask + (tile >> 2), ~(0xffu << (((uint32_t)tile & 3u) * 8u)));  // at rest for the next pass
    uint64_t rank = A.tile_off[tile];
    for (uint32_t w = 0; w < wave; ++w) rank += s_cnt[w];
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
      const uint32_t m = masks[j];
      const uint32_t pc = (uint32_t)__popc(m);
      const uint32_t incl = wave_incl_scan_u32(pc, lane);
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      if (m) {
        uint64_t r = rank + (incl - pc);
        const uint64_t unit_off = wbase + (uint64_t)j * kWaveLoad + (uint64_t)lane * kUnit;
        uint32_t mm = m;
        while (mm) {
          const uint32_t b = (uint32_t)__ffs((int)mm) - 1u;
          mm &= mm - 1u;
          if (A.m_cap == 0 || r < A.m_cap) {  // bounded emission (xsg_count_async): what does not fit is counted, not stored
            A.m_pos[r] = unit_off + b - (KIND >= kLong ? P.koff : 0u);
            A.m_chunk[r] = c;
          }
          ++r;
        }
      }
      rank += total;
    }
  }
}

// The kernel proper.  (Round 2 also measured an occupancy request for the class-sequence kinds -- they keep sixteen
// lane masks and the window dwords in scalar registers and take all 106 SGPRs the compiler may use, 7 waves per SIMD
// instead of 8; `amdgpu_waves_per_eu(8, 8)` brings them to 78 SGPRs and a dozen more spills.  Three builds, two
// interleaved rounds on one box, 50 GiB: no gain for `She[r ]lock` (5.29 against 5.36 TB/s), a loss for `[Ss]herlock`
// and `[0-9]{4}-[0-9]{2}` (6.50 against 6.88, 6.52 against 6.94): not adopted.  scripts/ab_waves.sh.)
template <int KIND, bool WANT_NL, bool WANT_LINES, bool EMIT, int LOADS, bool ICASE, bool ALIGNED>
__global__ __launch_bounds__(kBlock) void k_scan(const ScanArgs A) {
  if (EMIT) {
    if (A.hit_tiles) {
      // the one-sync list route: the tiles that hold a match are listed (in order) on the device and so is their
      // number; a bounded grid strides over the list -- a workgroup per tile of the shard, each leaving at once
      // unless its count is non-zero, costs 150 us on a 10 GiB shard whatever the number of matches
      uint64_t H = *A.n_hits_dev;
      H = H < A.hit_cap ? H : A.hit_cap;
      for (uint64_t h = blockIdx.x; h < H; h += gridDim.x) {
        scan_tile<KIND, WANT_NL, WANT_LINES, EMIT, LOADS, ICASE, ALIGNED>(A, (uint64_t)A.hit_tiles[h]);
        __syncthreads();  // the tile's LDS words are reused by the next one
      }
      return;
    }
  }
  scan_tile<KIND, WANT_NL, WANT_LINES, EMIT, LOADS, ICASE, ALIGNED>(A, (uint64_t)blockIdx.x + (uint64_t)blockIdx.y * gridDim.x);
}

template <int KIND, bool ICASE, bool ALIGNED>
static hipError_t launch_scan_kind(const ScanArgs& a, bool want_nl, bool want_lines, bool emit, dim3 grid,
                                   hipStream_t s) {
  constexpr int LOADS = 4;  // 16 KiB tiles (32 KiB measured 8 % slower)
#define KSCAN_LAUNCH(NL, LINES, EM) hipLaunchKernelGGL((k_scan<KIND, NL, LINES, EM, LOADS, ICASE, ALIGNED>), grid, dim3(kBlock), 0, s, a)
  if (emit) {
    KSCAN_LAUNCH(false, false, true);
  } else if (want_lines) {
    if (want_nl)
      KSCAN_LAUNCH(true, true, false);
    else
      KSCAN_LAUNCH(false, true, false);
  } else {
    if (want_nl)
      KSCAN_LAUNCH(true, false, false);
    else
      KSCAN_LAUNCH(false, false, false);
  }
#undef KSCAN_LAUNCH
  return hipGetLastError();
}

template <int KIND>
static hipError_t launch_scan_loads(const ScanArgs& a, bool want_nl, bool want_lines, bool emit, dim3 grid,
                                    hipStream_t s) {
  if (a.tile_bytes != kDefaultTileBytes) return hipErrorInvalidValue;
  constexpr bool kWindow = KIND == kTwo || KIND == kLong || KIND == kClass;
  if (kWindow && a.pat.hot)  // the aligned-dword trigger exists for the 8-byte-window kinds only
    return a.pat.icase ? launch_scan_kind<KIND, true, kWindow>(a, want_nl, want_lines, emit, grid, s)
                       : launch_scan_kind<KIND, false, kWindow>(a, want_nl, want_lines, emit, grid, s);
  return a.pat.icase ? launch_scan_kind<KIND, true, false>(a, want_nl, want_lines, emit, grid, s)
                     : launch_scan_kind<KIND, false, false>(a, want_nl, want_lines, emit, grid, s);
}

static dim3 tile_grid(uint64_t ntiles) {
  // grid.x is limited to 2^31-1 blocks; fold very large shards into y
  const uint64_t maxx = 1u << 30;
  if (ntiles <= maxx) return dim3((unsigned)ntiles, 1, 1);
  return dim3((unsigned)maxx, (unsigned)((ntiles + maxx - 1) / maxx), 1);
}

// A needle of 4..8 bytes that an earlier count found dense in this data is decided byte-parallel, like the 1..3-byte
// needles (scan_load, kMask1): no hot filter, no slow path -- every wave-load would take it.  (ignore_case only for
// needles of letters, which need no fold: folding every byte up front costs more than the slow path does;
// XSG_DENSE_BYTES=0 switches the re-routing off.)
// The matching-lines count of a 4..6-byte needle goes that way at ANY density: those kinds have no aligned-dword trigger,
// and their window filter plus the line bookkeeping is more VALU work per wave-load than deciding every byte (natural
// text, 10 GiB, kernel alone: `yield`, one per 100 KiB, 0.864 -> 0.907 of 8 TB/s; `import`, one per 6 KiB, 0.817 -> 0.861;
// the bench corpus' `Sherl`, one per 1.6 MB, 0.879 -> 0.915; 7 and 8 bytes lose: `finally` 0.865 -> 0.848, `Sherlock`
// 0.932 -> 0.807 -- profiles/r04_density_routes.txt).
static bool dense_bytes_route(const ScanArgs& a, bool want_lines, bool emit) {
  static const bool on = [] { const char* e = getenv("XSG_DENSE_BYTES"); return !(e && *e == '0'); }();
  if (!on || (a.pat.icase && !a.pat.lazy_exact)) return false;
  if (a.pat.kind != kOne && a.pat.kind != kMask2 && a.pat.kind != kTwo) return false;
  if (a.dense_hint) return true;
  static const bool lines_rule = [] { const char* e = getenv("XSG_DENSE_LINES"); return !(e && *e == '0'); }();
  return lines_rule && want_lines && !emit && a.pat.plen <= 6u;
}

static uint32_t pick_stagger(const ScanArgs& a, bool want_nl, bool want_lines, bool emit) {
  if (a.tune != kTuneAuto) return a.tune & 0xffu;
  // The stagger only pays where the kernel is memory-bound: it takes issue slots
  // away from the VALU-bound variants (scripts/tune_sweep.py: plain count +5 %,
  // count_lines +1 % at a small stagger, every heavier variant -1..-3 %).
  // (unmasked compares = plen >= 4; a needle that is dense in the text makes any variant VALU-heavy
  // and loses 2-3 % to the stagger -- not knowable before the scan: xsg_shard_tune measures it)
  // ignore_case: the hot loop of an 8-byte pattern only ORs 0x20 into the data (LAZY in scan_load) and stays
  // memory-bound (7.0 -> 7.36 TB/s with the stagger); the other kinds measured 2-3 % slower with it
  const bool light = (a.pat.kind == kOne || a.pat.kind == kTwo || a.pat.kind == kLong) &&
                     (!a.pat.icase || a.pat.kind == kTwo) && !want_nl && !emit;
  // count_lines too: 7.1 TB/s at 4 against 7.46-7.49 at 16 on the 50 GiB shard.  A needle an earlier count found dense in
  // this data (ScanArgs::dense_hint) keeps the slow path busy: 4 (`that`, 191 M matches in 50 GiB: 8.25 ms at 16, 7.66 at 4)
  return !light ? 0u : a.dense_hint >= 2u ? 4u : kDefaultStagger;
}

void describe_scan(const ScanArgs& a, bool want_nl, bool want_lines, bool emit, char* out, size_t cap) {
  if (!out || !cap) return;
  const char* b[2] = {"false", "true"};
  if (a.pat.kind == kDfa) {
    // count passes: one wave per 4 KiB span (k_rx_count); the emit pass: the tile-cooperative k_rx_scan
    static const char* const rw = getenv("XSG_RX_WAVE");
    char name[48];
    if (emit || (rw && *rw == '0'))
      snprintf(name, sizeof name, "xsg::k_rx_scan<%s, %s>", b[emit], b[emit ? 0 : want_lines]);
    else
      snprintf(name, sizeof name, "xsg::k_rx_count<%s>", b[want_lines]);
    snprintf(out, cap, "%s states=%u classes=%u%s", name, a.pat.rx_ncls ? a.pat.rx_fwd_n / a.pat.rx_ncls : 0u, a.pat.rx_ncls,
             a.tile_mask ? " (tiles marked by the factor prefilter only)" : "");
    return;
  }
  ScanArgs r = a;
  if (dense_bytes_route(a, want_lines, emit)) r.pat.kind = kMask1;
  const bool window = r.pat.kind == kTwo || r.pat.kind == kLong || r.pat.kind == kClass;
  const int kind = (r.pat.kind == kClass && r.pat.cls_fast && !r.pat.hot) ? (int)kClassFast : (int)r.pat.kind;
  snprintf(out, cap, "xsg::k_scan<%d, %s, %s, %s, 4, %s, %s> stagger=%u%s", kind, b[emit ? 0 : want_nl],
           b[emit ? 0 : want_lines], b[emit], b[r.pat.icase ? 1 : 0], b[window && r.pat.hot ? 1 : 0],
           pick_stagger(r, want_nl, want_lines, emit), r.pat.kind != a.pat.kind ? " (dense: byte-parallel)" : "");
}

static hipError_t launch_scan(const ScanArgs& a_in, bool want_nl, bool want_lines, bool emit, hipStream_t s) {
  if (a_in.ntiles == 0) return hipSuccess;
  ScanArgs a = a_in;
  if (dense_bytes_route(a, want_lines, emit)) a.pat.kind = kMask1;
  a.tune = pick_stagger(a, want_nl, want_lines, emit);
  static const uint64_t emit_grid = [] { const char* e = getenv("XSG_EMIT_GRID"); return e && atoll(e) > 0 ? (uint64_t)atoll(e) : 16384ull; }();
  const dim3 grid = (emit && a.hit_tiles) ? dim3((unsigned)std::min<uint64_t>(std::max<uint64_t>(a.hit_cap, 1), emit_grid), 1, 1)
                                         : tile_grid(a.ntiles);
  switch (a.pat.kind) {
    case kMask1: return launch_scan_loads<kMask1>(a, want_nl, want_lines, emit, grid, s);
    case kOne: return launch_scan_loads<kOne>(a, want_nl, want_lines, emit, grid, s);
    case kMask2: return launch_scan_loads<kMask2>(a, want_nl, want_lines, emit, grid, s);
    case kTwo: return launch_scan_loads<kTwo>(a, want_nl, want_lines, emit, grid, s);
    case kClass:
      // a window that takes the exact 16 + 32 bit filter has its own instantiation (window filter only)
      if (a.pat.cls_fast && !a.pat.hot) return launch_scan_loads<kClassFast>(a, want_nl, want_lines, emit, grid, s);
      return launch_scan_loads<kClass>(a, want_nl, want_lines, emit, grid, s);
    default: return launch_scan_loads<kLong>(a, want_nl, want_lines, emit, grid, s);
  }
}

hipError_t launch_scan_count(const ScanArgs& a, bool want_nl, bool want_lines, hipStream_t s) {
  if (a.pat.kind == kDfa) return launch_rx_count(a, want_nl, want_lines, s);
  return launch_scan(a, want_nl, want_lines, false, s);
}
hipError_t launch_scan_emit(const ScanArgs& a, hipStream_t s) {
  if (a.pat.kind == kDfa) return launch_rx_emit(a, s);
  return launch_scan(a, false, false, true, s);
}

// ---------------------------------------------------------------------------
// k_count_finish: sums the per-tile outputs, replays the reference walk over
// every chunk's tail zone, and leaves the per-tile arrays as the next pass
// needs them (tile_cnt and tile_sum zero again).  One launch, no host-side
// memset before or after; the last workgroup to arrive adds up the per-block
// partial sums and writes the four counters (device and, if given, a pinned
// host mirror), so the counters need no zeroing either.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(kBlock) void k_count_finish(const FinishArgs A) {
  __shared__ uint64_t sh[kWaves];
  __shared__ __attribute__((aligned(16))) uint8_t s_zone[kWaves][kZoneStage];
  __shared__ uint8_t s_pat[kTailMaskMaxPlen + 3];
  __shared__ uint32_t s_is_last;
  const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t gsz = (uint64_t)gridDim.x * kBlock;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool need_tail = !A.pat.exact_tail && A.pat.plen > 1;
  const bool mask_tail = need_tail && A.pat.plen <= kTailMaskMaxPlen;
  if (mask_tail) {
    if (threadIdx.x < A.pat.plen) s_pat[threadIdx.x] = A.pat.d_pat[threadIdx.x];
    __syncthreads();
  }

  // ---- per-tile sums; tile_cnt goes back to zero as it is read (every word has exactly one reader)
  uint64_t cm = 0, cn = 0;
  for (uint64_t t = gid; t < A.ntiles; t += 4 * gsz) {
    uint32_t v[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t tt = t + (uint64_t)u * gsz;
      v[u] = tt < A.ntiles ? A.tile_cnt[tt] : 0u;
      w[u] = (A.want_nl && tt < A.ntiles) ? A.tile_nl[tt] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (v[u]) A.tile_cnt[t + (uint64_t)u * gsz] = 0u;
      cm += v[u];
      cn += w[u];
    }
  }
  if (!A.want_matches) cm = 0;

  // ---- one wave per chunk: where the walk stands at the end of the bulk part (from the
  // last tile that holds a match), the chunk's matching lines, then its tail zone
  uint64_t lines = 0;
  const uint64_t wave_id = gid >> 6, nwaves = gsz >> 6;
  for (uint64_t c = wave_id; c < A.nchunks; c += nwaves) {
    const uint64_t t0 = A.chunk_tile0[c], t1 = A.chunk_tile0[c + 1];
    const ChunkDev ch = A.chunks[c];
    const uint8_t* d = A.base + ch.offset;
    uint64_t last_end = 0;
    if (need_tail) {
      uint64_t t = t1;
      while (t > t0 && last_end == 0) {  // wave-uniform; 256 tiles a step, four independent loads
        const uint64_t lo = t - t0 >= 256 ? t - 256 : t0;
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint64_t idx = lo + (uint64_t)u * 64u + lane;
          const uint32_t x = idx < t ? A.tile_last[idx] : 0u;
          v[u] = (x >> 16) == A.epoch ? (x & 0xffffu) : 0u;  // words of older passes do not count
        }
#pragma unroll
        for (int u = 3; u >= 0; --u) {
          const unsigned long long bal = __ballot(v[u] != 0);
          if (bal && last_end == 0) {
            const int hi = 63 - __clzll((long long)bal);
            const uint32_t vv = (uint32_t)__builtin_amdgcn_readlane((int)v[u], hi);
            last_end = (lo + (uint64_t)u * 64u + (uint64_t)hi - t0) * (uint64_t)A.tile_bytes + vv;
          }
        }
        t = lo;
      }
    }
    if (A.want_lines) {
      // 4 per-wave summaries per tile, 256 entries a step.  An entry that was never written reads 0 = "a newline,
      // no match"; a group of 64 such entries combines to the same, so the common step is four ballots.
      uint32_t run = 0;  // identity
      const uint64_t e1 = t1 * kWaves;
      for (uint64_t e0 = t0 * kWaves; e0 < e1; e0 += 256) {
        uint32_t raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint64_t e = e0 + (uint64_t)u * 64u + lane;
          raw[u] = e < e1 ? A.tile_sum[e] : kSumNl;  // beyond the chunk: the identity (0 after the XOR)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (e0 + (uint64_t)u * 64u >= e1) break;  // wave-uniform
          const uint64_t e = e0 + (uint64_t)u * 64u + lane;
          if (__ballot(raw[u] != 0) == 0) {
            run = sum_combine(run, kSumNl);
          } else {
            const uint32_t v = raw[u] ^ kSumNl;
            if (raw[u] != 0 && e < e1) A.tile_sum[e] = 0u;  // back to "nothing found"
            const uint32_t csum = wave_sum_u32(v >> kSumCShift);
            run = sum_combine(run, sum_combine_lanes(__ballot(v & kSumNl), __ballot(v & kSumF), __ballot(v & kSumL), csum));
          }
        }
      }
      if (lane == 0) lines += sum_total_lines(run);
    }
    if (need_tail && ch.length) {
      // the line-mode entry point may lie a whole huge line away: found by the wave, not by lane 0 alone
      const uint64_t entry_lines = A.want_lines ? wave_walk_entry(d, ch.length, last_end, true, lane) : 0;
      if (mask_tail) {
        uint32_t nm = 0, nl = 0;
        wave_tail_counts(d, ch.length, s_pat, A.pat.plen, A.pat.icase != 0, s_zone[wave], lane, A.want_matches != 0,
                         last_end, A.want_lines != 0, entry_lines, &nm, &nl);
        if (lane == 0) cm += nm, lines += nl;
      } else {
        // long patterns: the zone does not fit one position per lane.  One lane walks it byte by byte after a
        // coalesced sweep of the whole wave has pulled the zone into the caches.
        const uint64_t Lr = (ch.length + 15u) & ~(uint64_t)15u;
        for (uint64_t off = (tail_zone_begin(ch.length, A.pat.plen) & ~(uint64_t)15u) + (uint64_t)lane * kUnit; off < Lr;
             off += kWaveLoad) {
          const uint4 v = *reinterpret_cast<const uint4*>(d + off);
          asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        }
        if (lane == 0) {
          if (A.want_matches)
            cm += tail_walk(d, ch.length, A.pat.d_pat, A.pat.plen, last_end, false, nullptr, 0, A.pat.icase != 0);
          if (A.want_lines)
            lines += tail_walk(d, ch.length, A.pat.d_pat, A.pat.plen, entry_lines, true, nullptr, 0, A.pat.icase != 0);
        }
      }
    }
  }

  // ---- per-block partial sums; the last block to arrive adds them up
  uint64_t t;
  t = block_sum_u64(cm, sh);
  if (threadIdx.x == 0) A.partials[3u * blockIdx.x + 0u] = t;
  t = block_sum_u64(lines, sh);
  if (threadIdx.x == 0) A.partials[3u * blockIdx.x + 1u] = t;
  t = block_sum_u64(cn, sh);
  if (threadIdx.x == 0) {
    A.partials[3u * blockIdx.x + 2u] = t;
    __threadfence();  // partials before the ticket
    // (one ticket word for up to 2048 workgroups is not what the kernel's time goes to: a two-level ticket, 16 group
    // words in front of it, left the 50 GiB shard's finish at its 48 us under rocprofv3 -- measured, removed)
    s_is_last = atomicAdd(A.ticket, 1u) == gridDim.x - 1u;
  }
  __syncthreads();
  if (!s_is_last) return;
  __threadfence();
  uint64_t a0 = 0, a1 = 0, a2 = 0;
  for (uint32_t b = threadIdx.x; b < gridDim.x; b += kBlock) {
    a0 += __hip_atomic_load(A.partials + 3u * b + 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a1 += __hip_atomic_load(A.partials + 3u * b + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a2 += __hip_atomic_load(A.partials + 3u * b + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  a0 = block_sum_u64(a0, sh);
  a1 = block_sum_u64(a1, sh);
  a2 = block_sum_u64(a2, sh);
  if (threadIdx.x == 0) {
    // an ascii_only expression met non-ASCII data: no number is handed out (all four counters read UINT64_MAX)
    const bool refuse = (A.pat.kind == kClass || A.pat.kind == kDfa) && A.pat.ascii_only &&
                        (__hip_atomic_load(A.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) != 0;
    if (A.cnt_is_lines) a1 = a0, a0 = 0;  // k_rx_scan counted matching lines into tile_cnt
    // a refusal poisons the counters (UINT64_MAX) -- or, for a caller that gave a status word, zeroes them and says why
    const uint64_t bad = A.status ? 0ull : UINT64_MAX;
    if (A.status) *A.status = refuse ? (uint64_t)XSG_STATUS_NONASCII : 0ull;
    const uint64_t v[XSG_NUM_COUNTERS] = {refuse ? bad : a0, refuse ? bad : a1, refuse ? bad : a2, refuse ? bad : A.total_bytes};
    for (int k = 0; k < XSG_NUM_COUNTERS; ++k) {
      A.counters[k] = v[k];
      if (A.host_counters) A.host_counters[k] = v[k];
    }
    *A.flags = 0u;
    *A.ticket = 0u;  // at rest for the next launch (stream order)
  }
}

hipError_t launch_count_finish(const FinishArgs& a, hipStream_t s) {
  // enough workgroups that every chunk gets its own wave (the per-chunk part is a
  // chain of dependent loads: latency-bound) and that a thread sums ~8 tiles
  uint64_t blocks = (a.nchunks + kWaves - 1) / kWaves;
  const uint64_t for_tiles = (a.ntiles + (uint64_t)kBlock * 8 - 1) / ((uint64_t)kBlock * 8);
  if (for_tiles > blocks) blocks = for_tiles;
  if (blocks < 1) blocks = 1;
  if (blocks > kFinishBlocks) blocks = kFinishBlocks;
  hipLaunchKernelGGL(k_count_finish, dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}


// Loading a code object costs milliseconds the first time one of its kernels is launched: xsg_ctx_create launches this
// empty kernel of every kernel file, so that the first search of a process does not pay for it (5.5 ms of the first
// xsg_count, scripts/first_call.py).
__global__ void k_warm_scan() {}
hipError_t warm_scan_kernels(hipStream_t s) {
  hipLaunchKernelGGL(k_warm_scan, dim3(1), dim3(1), 0, s);
  return hipGetLastError();
}

}  // namespace xsg
