// xsg_rx_kernels.hip -- k_rx_scan: the regex walks of the reference for expressions of variable length
// (include/xsearch/string_search/search_wrappers.h:63-87 `_regex_byte_offsets`, :250-269 `regex::count`), one LINE
// per lane.
//
// The reference walks a chunk sequentially: RE2::PartialMatch from the current position, report, continue behind
// the match (or behind the line).  The expressions served here (xsg_regex.h) cannot match '\n', so no match spans
// two lines and the walk of a chunk is the concatenation of the walks of its lines: lines are independent units,
// and a tile of 16 KiB of text holds a few hundred of them.  A workgroup stages its tile in LDS (coalesced 16-byte
// loads, as k_scan reads), every lane takes the lines that START in its 64-byte segment of the tile, and walks each
// with the two automata RE2 itself would use:
//   forward  (leftmost-first, an any-byte loop of lowest priority in front): from the current position to the state
//            going dead (a '\n' kills every state); the last position at which the state held a match is the END of
//            the leftmost-first match;
//   reverse  (longest match, anchored at that end): back to the current position; the last position at which the
//            state held a match is the START.  Only needed where offsets are reported (EMIT).
// A line that runs past the end of the tile is followed through global memory by the lane that owns its start.
// Per-tile outputs are those of k_scan (tile_cnt: matches -- or matching lines -- of the lines that start in the
// tile, nothing stored when there are none; tile_nl on request; emission at tile_off ranks), so the finish kernel
// and the whole list pipeline behind it are shared.  The automata are tables of pre-multiplied uint16 row offsets
// in LDS: next = fwd[state + class_of[byte]]; all of it is byte/integer work, bound by instruction issue and LDS latency, no MFMA.
// With a factor prefilter (xsg_api.cpp: ensure_factor_mask) the kernel returns at once from tiles in which no line with
// an occurrence of the factor starts.
// The staged tile is translated to class codes in place (one table read per byte, off every dependency chain), the
// walks read four codes at a time, and a walk in the start state skips from trigger byte to trigger byte.
#include <cstdlib>

#include "xsg_internal.h"

namespace xsg {

constexpr uint32_t kRxTile = kDefaultTileBytes;     // 16 KiB: the tile geometry of the shard's per-tile arrays
constexpr uint32_t kRxSeg = kRxTile / kBlock;       // 64 bytes of the tile per lane
// byte p of the tile lives at p + 4 * (p / 64): a lane's segment starts 17 dwords after its neighbour's, so the 64
// lanes of a wave, each somewhere in its own segment, spread over all LDS banks (64-byte strides would put them on two)
constexpr uint32_t kRxTileLds = kRxTile + 4 * kBlock;
__device__ __forceinline__ uint32_t rx_addr(uint32_t p) { return p + ((p >> 6) << 2); }

struct RxCtx {
  const uint32_t* tilew;   // LDS, swizzled (rx_addr): the tile as CLASS CODES, one per byte (bit 7, if `skip`: trigger)
  const uint8_t* cls;      // LDS, 256 bytes: class of a byte, same encoding (for bytes read from global memory)
  const unsigned long long* trig;  // LDS, one word per 64-byte segment of the tile: bit i <=> byte i is a trigger
  uint32_t skip, cmask;    // skip: trigger bits exist; cmask: 0x7f then, else 0xff
  const uint16_t* fwd;     // LDS
  const uint16_t* rev;     // global (EMIT only: read for reported matches, not per scanned byte)
  const uint8_t* cbase;    // the chunk in global memory
  uint64_t toff;           // chunk-relative offset of the tile
  uint64_t L;              // chunk length
  uint32_t fwd_start, fwd_acc, rev_start, rev_acc;
  uint32_t tile_bytes = kRxTile;  // bytes staged at `tilew` (k_rx_scan: the 16 KiB tile; k_rx_count: a wave's 4 KiB span)
  uint32_t nseg = kBlock;         // ... = this many 64-byte segments (trigger words)
  const uint8_t* la = nullptr;    // LDS: the (up to) 64 RAW bytes behind the staged bytes, '\n' beyond the chunk (k_rx_count)
};

// class code of the byte at q (< L), wherever it lives
__device__ __forceinline__ uint32_t rx_class(const RxCtx& X, uint64_t q) {
  const uint64_t rel = q - X.toff;
  if (rel < X.tile_bytes) {
    const uint32_t r = (uint32_t)rel;
    return (X.tilew[rx_addr(r & ~3u) >> 2] >> (8u * (r & 3u))) & X.cmask;
  }
  return X.cls[X.cbase[q]] & X.cmask;
}

// One line, from its start `cur` (chunk-relative): the reference's walk restricted to the line.  Returns the number
// of matches (LINES: 1 if there is any).  EMITTING: their start offsets go to m_pos[rank...], rank advances.
// Positions inside the tile are 32-bit and tile-relative (`rel`); a line that leaves the tile continues through
// global memory with 64-bit offsets, byte by byte (rare: one line per tile at most).
template <bool EMITTING, bool LINES>
__device__ __forceinline__ uint32_t rx_walk_line(const RxCtx& X, uint64_t cur, const ScanArgs& A, uint32_t chunk,
                                                 uint64_t& rank) {
  uint32_t n = 0;
  for (;;) {
    uint32_t st = X.fwd_start;
    uint64_t last_end = 0;  // a match ends behind at least one byte: 0 = none yet
    bool stop = false;      // the state died, or (LINES) a match was seen
    uint64_t q = cur;
    if (cur - X.toff < X.tile_bytes) {
      uint32_t rel = (uint32_t)(cur - X.toff);
      uint32_t last_rel = 0;
      while (rel < X.tile_bytes) {
        // In its start state the automaton only waits for a byte that can begin a match: every other byte leaves it
        // where it is.  Those bytes (and '\n') are the tile's TRIGGERS, flagged for all 16 KiB at once in the staging
        // phase; the walk jumps from one to the next on the bit masks instead of stepping through the text.
        if (X.skip && st == X.fwd_start) {
          uint32_t w = rel >> 6;
          unsigned long long m = X.trig[w] & (~0ull << (rel & 63u));
          while (!m && ++w < X.nseg) m = X.trig[w];
          if (!m) {  // none left in the tile (the line runs on behind it)
            rel = X.tile_bytes;
            break;
          }
          rel = w * kRxSeg + (uint32_t)__builtin_ctzll(m);
        }
        // the class codes of four bytes with one read; bytes at or beyond L are '\n' here, which kills every state
        // The four steps are straight-line and predicated instead of leaving the loop one lane at a time (the exec-mask
        // bookkeeping of a divergent exit per byte cost more than the step): a lane that is dead stays dead -- row 0
        // of the table is all zeros --, a lane whose dword started in the middle sits out the last steps.  Stepping on
        // in the start state to the end of the dword is harmless (skipping is an optimisation, not a condition).
        uint32_t cw = X.tilew[(rel >> 2) + (rel >> 6)] >> (8u * (rel & 3u));  // = rx_addr(rel & ~3) / 4
        const uint32_t nsteps = 4u - (rel & 3u);
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
          const uint32_t nst = X.fwd[st + (cw & X.cmask)];
          cw >>= 8u;
          const bool on = k < nsteps && st != 0u;
          st = on ? nst : st;
          const bool moved = on && nst != 0u;
          rel += moved ? 1u : 0u;
          last_rel = (moved && nst >= X.fwd_acc) ? rel : last_rel;
        }
        if (st == 0u || (LINES && last_rel != 0u)) {  // dead (the line ended, or nothing can outrank the match seen); a matching line
          stop = true;
          break;
        }
      }
      q = X.toff + rel;
      if (last_rel) last_end = X.toff + last_rel;
    }
    while (!stop && q < X.L) {  // behind the tile
      const uint64_t bq = q - (X.toff + X.tile_bytes);  // the first 64 of those bytes may be at hand in LDS
      const uint32_t byte = (X.la && bq < 64u) ? X.la[bq] : X.cbase[q];
      st = X.fwd[st + (X.cls[byte] & X.cmask)];
      if (st == 0) break;
      ++q;
      if (st >= X.fwd_acc) {
        last_end = q;
        if (LINES) break;
      }
    }
    if (!last_end) break;  // no (further) match in this line
    ++n;
    if (LINES) break;
    if (EMITTING) {
      uint32_t rs = X.rev_start;
      uint64_t r = last_end, start = last_end;
      while (r > cur) {
        rs = X.rev[rs + rx_class(X, r - 1)];
        if (rs == 0) break;
        --r;
        if (rs >= X.rev_acc) start = r;
      }
      if (A.m_cap == 0 || rank < A.m_cap) {  // bounded emission, as k_scan
        A.m_pos[rank] = start;
        A.m_chunk[rank] = chunk;
      }
      ++rank;
    }
    cur = last_end;
  }
  return n;
}

__device__ __forceinline__ uint32_t rx_wave_sum(uint32_t v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += (uint32_t)__shfl_xor((int)v, s);
  return v;
}

template <bool EMIT, bool LINES>
__global__ __launch_bounds__(kBlock) void k_rx_scan(const ScanArgs A, const uint32_t want_nl) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];  // the forward table
  __shared__ __attribute__((aligned(16))) uint8_t s_tile[kRxTileLds];
  __shared__ uint8_t s_cls[256];
  __shared__ uint8_t s_lastnl[kBlock];  // does the lane's segment end in '\n'?  (the next lane's first byte is a line start then)
  __shared__ unsigned long long s_trig[kBlock];
  __shared__ uint32_t s_w[kWaves];
  __shared__ uint32_t s_tail[16];  // the 64 bytes behind the tile (no-trigger tiles: where the line that runs on ends)

  const uint64_t tile = (uint64_t)blockIdx.x + (uint64_t)blockIdx.y * gridDim.x;
  if (tile >= A.ntiles) return;
  if (EMIT && A.tile_cnt[tile] == 0) return;
  // the factor prefilter: no line that starts in this tile holds the factor every match contains -> nothing to find
  // (a pass that also counts newlines needs every tile)
  if (A.tile_mask && !want_nl && A.tile_mask[tile] == 0u) return;
  const PatternDev P = A.pat;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t c = A.tile_chunk ? A.tile_chunk[tile] : 0u;
  const ChunkDev ch = A.chunks[c];
  const uint8_t* cbase = A.base + ch.offset;
  const uint64_t L = ch.length;
  const uint64_t Lr = (L + 15u) & ~(uint64_t)15u;
  const uint64_t toff = (tile - A.chunk_tile0[c]) * (uint64_t)kRxTile;

  // ---- the tile's bytes, in registers first (bytes at or beyond L read as '\n': every line ends in one).  With few
  // trigger byte values (PatternDev::rx_ntrig: `Sherlock|Holmes` has S and H) the loads are tested for them
  // byte-parallel, as k_scan tests for a one-byte needle: a tile that holds none cannot begin a match, and it leaves
  // here -- nothing staged in LDS (not the class map, not the automaton, not the tile), no class translation (64
  // table reads per lane: 1122 VALU + 132 LDS instructions per 4 KiB before a single line was walked,
  // profiles/r02_rx_pmc.txt) -- except for the one line that starts in it and runs on behind it (below).
  const bool quick = P.rx_ntrig != 0 && !EMIT;
  uint32_t dd[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t off = toff + ((uint64_t)j * kBlock + tid) * kUnit;
    dd[j][0] = dd[j][1] = dd[j][2] = dd[j][3] = 0x0a0a0a0au;
    if (off < Lr) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(cbase + off));  // read once: as k_scan
      dd[j][0] = v.x, dd[j][1] = v.y, dd[j][2] = v.z, dd[j][3] = v.w;
    }
  }
  uint32_t prev_byte = '\n';  // the byte in front of the tile (lane 0; fetched with the tile's loads, not behind the barrier)
  if (quick && tid == 0 && toff != 0) prev_byte = cbase[toff - 1];
  if (quick && tid < 4u) {  // four lanes fetch the 64 bytes behind the tile together with the tile's own loads
    const uint64_t off = toff + kRxTile + (uint64_t)tid * kUnit;
    uint32_t t[4] = {0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au};  // at or beyond the chunk's end: '\n'
    if (off < Lr) {
      const uint4 v = *reinterpret_cast<const uint4*>(cbase + off);
      t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint64_t o = off + 4u * q;
        const uint32_t keep = o >= L ? 0u : (o + 4u > L ? ((1u << (8u * (uint32_t)(L - o))) - 1u) : 0xffffffffu);
        t[q] = (t[q] & keep) | (0x0a0a0a0au & ~keep);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) s_tail[tid * 4u + q] = t[q];
  }
  uint32_t hi = 0, any_trig = 0, any_nl = 0, nl_cnt = 0;
  const uint32_t tv0 = (P.rx_trig4 & 0xffu) * 0x01010101u, tv1 = ((P.rx_trig4 >> 8) & 0xffu) * 0x01010101u;
  const uint32_t tv2 = ((P.rx_trig4 >> 16) & 0xffu) * 0x01010101u, tv3 = (P.rx_trig4 >> 24) * 0x01010101u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t off = toff + ((uint64_t)j * kBlock + tid) * kUnit;
    if (off < Lr && off + kUnit > L) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint64_t o = off + 4u * q;
        const uint32_t keep = o >= L ? 0u : (o + 4u > L ? ((1u << (8u * (uint32_t)(L - o))) - 1u) : 0xffffffffu);
        dd[j][q] = (dd[j][q] & keep) | (0x0a0a0a0au & ~keep);
      }
    }
    hi |= dd[j][0] | dd[j][1] | dd[j][2] | dd[j][3];
    if (quick) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // bit 7 of some byte set <=> some byte equals the value (exact as an existence test); unused slots repeat value 0
        const uint32_t x0 = dd[j][q] ^ tv0, x1 = dd[j][q] ^ tv1;
        any_trig |= ((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1);
        if (P.rx_ntrig > 2u) {  // (scalar)
          const uint32_t x2 = dd[j][q] ^ tv2, x3 = dd[j][q] ^ tv3;
          any_trig |= ((x2 - 0x01010101u) & ~x2) | ((x3 - 0x01010101u) & ~x3);
        }
        const uint32_t y = dd[j][q] ^ 0x0a0a0a0au;
        any_nl |= (y - 0x01010101u) & ~y;
      }
    }
  }
  if (P.ascii_only && __any((hi & 0x80808080u) != 0) && lane == 0) atomicOr(A.flags, 1u);  // the search must refuse
  if (quick) {
    // one barrier: every wave's verdicts (bit 0: a trigger, bit 1: a '\n') through LDS
    // (bit 2: the tile's last byte is a '\n' -- lane 255 holds it; bit 3: its first byte opens a line -- lane 0)
    const uint32_t wf = (__ballot((any_trig & 0x80808080u) != 0) != 0 ? 1u : 0u) | (__ballot((any_nl & 0x80808080u) != 0) != 0 ? 2u : 0u) |
                        (__ballot(tid == (uint32_t)kBlock - 1u && (dd[3][3] >> 24) == 0x0au) != 0 ? 4u : 0u) |
                        (__ballot(tid == 0u && prev_byte == 0x0au) != 0 ? 8u : 0u);
    if (lane == 0) s_w[wave] = wf;
    __syncthreads();
    const uint32_t tf = s_w[0] | s_w[1] | s_w[2] | s_w[3];
    if (!(tf & 1u)) {
      if (want_nl) {  // bytes at or beyond L were replaced by '\n': count inside the chunk only
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint64_t off = toff + ((uint64_t)j * kBlock + tid) * kUnit;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint64_t o = off + 4u * q;
            const uint32_t keep = o >= L ? 0u : (o + 4u > L ? ((1u << (8u * (uint32_t)(L - o))) - 1u) : 0xffffffffu);
            const uint32_t y = (dd[j][q] & keep) ^ (0x0a0a0a0au & keep) ^ ~keep;  // bytes outside the chunk: not a newline
            nl_cnt += (uint32_t)__popc(~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu));
          }
        }
        const uint32_t wn = rx_wave_sum(nl_cnt);
        __syncthreads();  // every wave has read the verdicts
        if (lane == 0) s_w[wave] = wn;
        __syncthreads();
        if (tid == 0) A.tile_nl[tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
      }
      // does a line START in this tile and run on behind it?  It does iff the tile does not end the chunk, its last byte
      // is no '\n', and it holds a line start: a '\n' somewhere, or its first byte opens the chunk / follows a '\n'.
      if (tid == 0) {
        const uint64_t tend = toff + kRxTile;
        if (tend < L && !(tf & 4u) && (tf & (2u | 8u))) {
          // Inside the tile the line holds no trigger: the automaton reaches the tile's end in its start state.  Nearly
          // always the line ends within the next few dozen bytes without a trigger either -- decided on the 64 bytes
          // fetched with the tile, a few LDS reads; stepping the automaton through global memory byte by byte, one
          // dependent load each, held the workgroup's slot for ~15 us per tile (2.2 TB/s whatever the text).
          bool walk = true;
          for (uint32_t k = 0; k < 16u; ++k) {
            const uint32_t x = s_tail[k];
            const uint32_t y = x ^ 0x0a0a0a0au;
            const uint32_t nf = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);  // 0x80 in every byte that is '\n' (exact)
            uint32_t tg = 0;
            for (uint32_t v = 0; v < P.rx_ntrig; ++v) {
              const uint32_t z = x ^ (((P.rx_trig4 >> (8u * v)) & 0xffu) * 0x01010101u);
              tg |= ~(((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z | 0x7f7f7f7fu);
            }
            if (nf) {  // the line ends in this dword: a trigger before its '\n'?
              walk = (tg & ((nf & (0u - nf)) - 1u)) != 0u;
              break;
            }
            if (tg) break;  // a trigger, and the line goes on: walk it
          }
          if (walk) {  // rare: the automaton's tables straight from global memory (nothing was staged)
            RxCtx X;
            X.tilew = nullptr, X.cls = P.d_pat, X.fwd = reinterpret_cast<const uint16_t*>(P.d_pat + 256);
            X.trig = nullptr, X.skip = P.rx_skip, X.cmask = 0x7fu;
            X.rev = nullptr;
            X.cbase = cbase, X.toff = toff, X.L = L;
            X.fwd_start = P.rx_fwd_start, X.fwd_acc = P.rx_fwd_acc, X.rev_start = P.rx_rev_start, X.rev_acc = P.rx_rev_acc;
            uint64_t rank = 0;
            const uint32_t n = rx_walk_line<false, LINES>(X, tend, A, c, rank);
            if (n) atomicAdd(A.tile_cnt + tile, n);
          }
        }
      }
      return;
    }
    __syncthreads();  // s_w is reused below
  }
  // ---- stage: class map, forward table, the tile
  s_cls[tid] = P.d_pat[tid];
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(P.d_pat + 256);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_dyn);
    for (uint32_t k = tid; k < (P.rx_fwd_n + 1u) / 2u; k += kBlock) dst[k] = src[k];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t p = ((uint32_t)j * kBlock + tid) * kUnit;
    uint32_t* w = reinterpret_cast<uint32_t*>(s_tile + rx_addr(p));
    w[0] = dd[j][0], w[1] = dd[j][1], w[2] = dd[j][2], w[3] = dd[j][3];
  }
  __syncthreads();

  // ---- this lane's 64-byte segment: where its lines start (a position < L that follows a '\n' or opens the chunk),
  // and its bytes translated to CLASS CODES in place -- the automata only ever ask for the class of a byte, so the
  // table read is done once per byte here, 64 reads per lane that depend on nothing but the data and are issued in
  // batches of 16, instead of once per step inside a walk, where it sits on the state's dependency chain.  With
  // `skip`, bit 7 of a class code flags a trigger byte, and the flags of the segment become its trigger word.
  const uint32_t seg = tid * kRxSeg;
  unsigned long long nlm = 0, tgm = 0;
  {
    uint32_t* w = reinterpret_cast<uint32_t*>(s_tile + rx_addr(seg));
    uint32_t d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = w[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t x = d[k] ^ 0x0a0a0a0au;
      const uint32_t f = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu) >> 7;  // bit 0, 8, 16, 24 <=> byte is '\n'
      const uint32_t nib = (f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u);
      nlm |= (unsigned long long)nib << (4 * k);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint32_t t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = s_cls[(d[4 * g + (i >> 2)] >> (8 * (i & 3))) & 0xffu];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t cw = t[4 * j] | (t[4 * j + 1] << 8) | (t[4 * j + 2] << 16) | (t[4 * j + 3] << 24);
        w[4 * g + j] = cw;
        const uint32_t f = (cw >> 7) & 0x01010101u;
        const uint32_t nib = (f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u);
        tgm |= (unsigned long long)nib << (4 * (4 * g + j));
      }
    }
  }
  s_lastnl[tid] = (uint8_t)(nlm >> 63);
  s_trig[tid] = tgm;  // '\n' is a trigger by construction (xsg_api.cpp); all zero without `skip`, never read then
  __syncthreads();
  const uint64_t seg_off = toff + seg;
  const uint32_t nvalid = seg_off >= L ? 0u : (L - seg_off >= kRxSeg ? kRxSeg : (uint32_t)(L - seg_off));
  const unsigned long long valid = nvalid >= 64u ? ~0ull : ((1ull << nvalid) - 1ull);
  bool prev_nl;
  if (tid == 0) prev_nl = toff == 0 || cbase[toff - 1] == '\n';
  else prev_nl = s_lastnl[tid - 1u] != 0;
  const unsigned long long starts = ((nlm << 1) | (prev_nl ? 1ull : 0ull)) & valid;

  RxCtx X;
  X.tilew = reinterpret_cast<const uint32_t*>(s_tile), X.cls = s_cls, X.fwd = reinterpret_cast<const uint16_t*>(s_dyn);
  X.trig = s_trig, X.skip = P.rx_skip, X.cmask = P.rx_skip ? 0x7fu : 0xffu;
  X.rev = reinterpret_cast<const uint16_t*>(P.d_pat + ((256u + 2u * (size_t)P.rx_fwd_n + 15u) & ~(size_t)15u));
  X.cbase = cbase, X.toff = toff, X.L = L;
  X.fwd_start = P.rx_fwd_start, X.fwd_acc = P.rx_fwd_acc, X.rev_start = P.rx_rev_start, X.rev_acc = P.rx_rev_acc;

  uint32_t cnt = 0;
  uint64_t rank = 0;
  for (unsigned long long m = starts; m; m &= m - 1ull)
    cnt += rx_walk_line<false, LINES>(X, seg_off + (uint32_t)__builtin_ctzll(m), A, c, rank);

  if (!EMIT) {
    // as k_scan: tile_cnt is zero at rest and only a wave that found something writes
    if (__any(cnt != 0)) {
      const uint32_t wc = rx_wave_sum(cnt);
      if (lane == 0) atomicAdd(A.tile_cnt + tile, wc);
    }
    if (want_nl) {
      const uint32_t wn = rx_wave_sum((uint32_t)__popcll(nlm & valid));
      if (lane == 0) s_w[wave] = wn;
      __syncthreads();
      if (tid == 0) A.tile_nl[tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
  } else {
    // ranks: lanes own ascending segments, a lane's lines ascend, a line's matches ascend
    uint32_t incl = cnt;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, s);
      if (lane >= (uint32_t)s) incl += o;
    }
    if (lane == 63u) s_w[wave] = incl;
    __syncthreads();
    rank = A.tile_off[tile] + (incl - cnt);
    for (uint32_t w = 0; w < wave; ++w) rank += s_w[w];
    if (cnt)
      for (unsigned long long m = starts; m; m &= m - 1ull)
        (void)rx_walk_line<true, false>(X, seg_off + (uint32_t)__builtin_ctzll(m), A, c, rank);
  }
}

// ---------------------------------------------------------------------------
// k_rx_count: the COUNT passes of the line-walking route, one WAVE per 4 KiB span, no workgroup barrier.
// k_rx_scan's workgroup decides together whether its tile can be left without staging (one barrier for the four waves'
// verdicts); measured on text without a trigger byte, that barrier is the difference between 4.3 and 6.7 TB/s -- a
// tile takes as long as the slowest of its four waves' loads, where k_scan's waves leave one by one.  Here the wave is
// the unit: it reads its own contiguous 4 KiB (as a k_scan wave does), tests them for the expression's few trigger
// bytes, and if there is none it is done -- apart from the one line that starts in the span and runs on behind it,
// decided on the 64 bytes behind the span.  A wave whose span holds a trigger stages and translates ITS span in its
// own quarter of the workgroup's LDS (wave-local fences only), walks the lines that start there, and follows a line
// that leaves the span through those 64 bytes and then global memory, exactly as k_rx_scan follows a line that leaves
// the tile.  The per-tile outputs are sums over the tile's waves (atomicAdd; tile_nl is zeroed by the launcher).
// The emit pass stays with k_rx_scan (its ranks need the tile's waves in step); both count the same thing: the
// matches of the lines that START in the tile.
// ---------------------------------------------------------------------------
constexpr uint32_t kRxSpan = kRxTile / kWaves;           // 4 KiB
constexpr uint32_t kRxSpanLds = kRxSpan + 4 * 64 + 64;    // swizzled span + the 64 raw bytes behind it

template <bool LINES>
__global__ __launch_bounds__(kBlock) void k_rx_count(const ScanArgs A, const uint32_t want_nl) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];  // the forward table
  __shared__ __attribute__((aligned(16))) uint8_t s_span[kWaves][kRxSpanLds];
  __shared__ uint8_t s_cls[256];
  __shared__ unsigned long long s_trig[kBlock];

  const uint64_t tile = (uint64_t)blockIdx.x + (uint64_t)blockIdx.y * gridDim.x;
  if (tile >= A.ntiles) return;
  if (A.tile_mask && !want_nl && A.tile_mask[tile] == 0u) return;  // the factor prefilter (see k_rx_scan)
  const PatternDev P = A.pat;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t c = A.tile_chunk ? A.tile_chunk[tile] : 0u;
  const ChunkDev ch = A.chunks[c];
  const uint8_t* cbase = A.base + ch.offset;
  const uint64_t L = ch.length;
  const uint64_t Lr = (L + 15u) & ~(uint64_t)15u;
  const uint64_t soff = (tile - A.chunk_tile0[c]) * (uint64_t)kRxTile + (uint64_t)wave * kRxSpan;  // this wave's span
  if (soff >= L) return;  // (only in a chunk's last tile)

  // ---- the span's bytes (bytes at or beyond L read as '\n'), the 256 bytes behind it (lane l: dword l), the byte
  // in front of it.  A span that lies inside the chunk together with those 256 bytes -- all but a chunk's last -- needs no
  // clamping at all (wave-uniform branch, as in k_scan).
  uint32_t dd[4][4];
  uint32_t la = 0x0a0a0a0au;
  const bool inner = soff + kRxSpan + 256u <= L;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  if (inner) {
    const uint8_t* p0 = cbase + soff + (uint64_t)lane * kUnit;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p0 + (uint64_t)j * 1024u));
      dd[j][0] = v.x, dd[j][1] = v.y, dd[j][2] = v.z, dd[j][3] = v.w;
    }
    la = *reinterpret_cast<const uint32_t*>(cbase + soff + kRxSpan + (uint64_t)lane * 4u);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t off = soff + ((uint64_t)j * 64u + lane) * kUnit;
      dd[j][0] = dd[j][1] = dd[j][2] = dd[j][3] = 0x0a0a0a0au;
      if (off < Lr) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(cbase + off));
        dd[j][0] = v.x, dd[j][1] = v.y, dd[j][2] = v.z, dd[j][3] = v.w;
        if (off + kUnit > L) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint64_t o = off + 4u * q;
            const uint32_t keep = o >= L ? 0u : (o + 4u > L ? ((1u << (8u * (uint32_t)(L - o))) - 1u) : 0xffffffffu);
            dd[j][q] = (dd[j][q] & keep) | (0x0a0a0a0au & ~keep);
          }
        }
      }
    }
    {
      const uint64_t o = soff + kRxSpan + (uint64_t)lane * 4u;
      if (o < Lr) {
        la = *reinterpret_cast<const uint32_t*>(cbase + o);
        const uint32_t keep = o >= L ? 0u : (o + 4u > L ? ((1u << (8u * (uint32_t)(L - o))) - 1u) : 0xffffffffu);
        la = (la & keep) | (0x0a0a0a0au & ~keep);
      }
    }
  }
  uint32_t prev_byte = '\n';
  if (lane == 0 && soff != 0) prev_byte = cbase[soff - 1];

  const bool quick = P.rx_ntrig != 0;
  uint32_t hi = 0, any_trig = 0, any_nl = 0;
  const uint32_t tv0 = (P.rx_trig4 & 0xffu) * 0x01010101u, tv1 = ((P.rx_trig4 >> 8) & 0xffu) * 0x01010101u;
  const uint32_t tv2 = ((P.rx_trig4 >> 16) & 0xffu) * 0x01010101u, tv3 = (P.rx_trig4 >> 24) * 0x01010101u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi |= dd[j][0] | dd[j][1] | dd[j][2] | dd[j][3];
    if (quick) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t x0 = dd[j][q] ^ tv0, x1 = dd[j][q] ^ tv1;
        any_trig |= ((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1);
        if (P.rx_ntrig > 2u) {  // (scalar)
          const uint32_t x2 = dd[j][q] ^ tv2, x3 = dd[j][q] ^ tv3;
          any_trig |= ((x2 - 0x01010101u) & ~x2) | ((x3 - 0x01010101u) & ~x3);
        }
        const uint32_t y = dd[j][q] ^ 0x0a0a0a0au;
        any_nl |= (y - 0x01010101u) & ~y;
      }
    }
  }
  if (P.ascii_only && __any((hi & 0x80808080u) != 0) && lane == 0) atomicOr(A.flags, 1u);  // the search must refuse
  if (want_nl) {  // newlines of the span (inside the chunk only: bytes beyond L were replaced by '\n')
    uint32_t nl_cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t off = soff + ((uint64_t)j * 64u + lane) * kUnit;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint64_t o = off + 4u * q;
        const uint32_t keep = o >= L ? 0u : (o + 4u > L ? ((1u << (8u * (uint32_t)(L - o))) - 1u) : 0xffffffffu);
        const uint32_t y = (dd[j][q] & keep) ^ (0x0a0a0a0au & keep) ^ ~keep;
        nl_cnt += (uint32_t)__popc(~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu));
      }
    }
    const uint32_t wn = rx_wave_sum(nl_cnt);
    if (lane == 0 && wn) atomicAdd(A.tile_nl + tile, wn);
  }

  uint8_t* const sp = s_span[wave];
  uint32_t* const law = reinterpret_cast<uint32_t*>(sp + kRxSpan + 4 * 64);  // the 64 raw bytes behind the span (full path)
  const uint64_t send = soff + kRxSpan;
  if (quick && __ballot((any_trig & 0x80808080u) != 0) == 0) {
    // ---- no trigger in the span.  Does a line START in it and run on behind it?  Iff the span does not end the chunk, its
    // last byte is no '\n', and it holds a line start: a '\n' somewhere, or its first byte opens the chunk / follows one.
    const bool last_is_nl = __ballot(lane == 63u && (dd[3][3] >> 24) == 0x0au) != 0;
    const bool has_start = __ballot((any_nl & 0x80808080u) != 0) != 0 || __ballot(lane == 0u && prev_byte == 0x0au) != 0;
    if (send >= L || last_is_nl || !has_start) return;
    // Inside the span the line holds no trigger: the automaton reaches the span's end in its start state.  Nearly always
    // the line ends within the next few dozen bytes without a trigger either: every lane holds a dword of the 256 bytes
    // behind the span; the first lane with a '\n' settles it -- no trigger in the lanes below, none below the '\n' in its
    // own dword.  (One lane looping over such dwords in LDS, the first version, cost every wave ~1500 cycles; with 64
    // bytes of look-ahead one wave in nine met a line that ran further and walked it through global memory: 3.2 TB/s
    // where 7.2 are possible.)
    const uint32_t y = la ^ 0x0a0a0a0au;
    const uint32_t nf = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);  // 0x80 per '\n' (exact)
    uint32_t tg = 0;
    for (uint32_t v = 0; v < P.rx_ntrig; ++v) {
      const uint32_t z = la ^ (((P.rx_trig4 >> (8u * v)) & 0xffu) * 0x01010101u);
      tg |= ~(((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z | 0x7f7f7f7fu);
    }
    const unsigned long long nfb = __ballot(nf != 0), tgb = __ballot(tg != 0);
    bool walk = true;  // no '\n' within 256 bytes: a long line, walked
    if (nfb) {
      const int f = __builtin_ctzll(nfb);
      const uint32_t own = (tg & ((nf & (0u - nf)) - 1u)) != 0u ? 1u : 0u;  // a trigger below the dword's first '\n'
      walk = (tgb & ((1ull << f) - 1ull)) != 0 || __builtin_amdgcn_readlane((int)own, f) != 0;
    }
    if (walk && lane == 0) {  // rare: the automaton's tables straight from global memory (this wave staged nothing)
      RxCtx X;
      X.tilew = nullptr, X.cls = P.d_pat, X.fwd = reinterpret_cast<const uint16_t*>(P.d_pat + 256);
      X.trig = nullptr, X.skip = P.rx_skip, X.cmask = 0x7fu;
      X.rev = nullptr;
      X.cbase = cbase, X.toff = soff, X.L = L;
      X.fwd_start = P.rx_fwd_start, X.fwd_acc = P.rx_fwd_acc, X.rev_start = P.rx_rev_start, X.rev_acc = P.rx_rev_acc;
      X.tile_bytes = kRxSpan, X.nseg = 64u;
      uint64_t rank = 0;
      const uint32_t n = rx_walk_line<false, LINES>(X, send, A, c, rank);
      if (n) atomicAdd(A.tile_cnt + tile, n);
    }
    return;
  }
  if (lane < 16u) law[lane] = la;

  // ---- the span holds a trigger (or the expression has many): stage and translate it, wave-locally.  The class map and
  // the forward table are the workgroup's; every wave that needs them writes them (the same values: idempotent).
  for (uint32_t k = lane; k < 256u / 4u; k += 64u)
    reinterpret_cast<uint32_t*>(s_cls)[k] = reinterpret_cast<const uint32_t*>(P.d_pat)[k];
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(P.d_pat + 256);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_dyn);
    for (uint32_t k = lane; k < (P.rx_fwd_n + 1u) / 2u; k += 64u) dst[k] = src[k];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t p = ((uint32_t)j * 64u + lane) * kUnit;  // span-relative
    uint32_t* w = reinterpret_cast<uint32_t*>(sp + rx_addr(p));
    w[0] = dd[j][0], w[1] = dd[j][1], w[2] = dd[j][2], w[3] = dd[j][3];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // this lane's 64-byte segment of the span: line starts, class codes in place, trigger word (as k_rx_scan)
  const uint32_t seg = lane * kRxSeg;
  unsigned long long nlm = 0, tgm = 0;
  {
    uint32_t* w = reinterpret_cast<uint32_t*>(sp + rx_addr(seg));
    uint32_t d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = w[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t x = d[k] ^ 0x0a0a0a0au;
      const uint32_t f = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu) >> 7;
      const uint32_t nib = (f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u);
      nlm |= (unsigned long long)nib << (4 * k);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint32_t t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = s_cls[(d[4 * g + (i >> 2)] >> (8 * (i & 3))) & 0xffu];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t cw = t[4 * j] | (t[4 * j + 1] << 8) | (t[4 * j + 2] << 16) | (t[4 * j + 3] << 24);
        w[4 * g + j] = cw;
        const uint32_t f = (cw >> 7) & 0x01010101u;
        const uint32_t nib = (f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u);
        tgm |= (unsigned long long)nib << (4 * (4 * g + j));
      }
    }
  }
  s_trig[tid] = tgm;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint64_t seg_off = soff + seg;
  const uint32_t nvalid = seg_off >= L ? 0u : (L - seg_off >= kRxSeg ? kRxSeg : (uint32_t)(L - seg_off));
  const unsigned long long valid = nvalid >= 64u ? ~0ull : ((1ull << nvalid) - 1ull);
  // does the previous segment end in '\n'?  (lane 0: the byte in front of the span)
  const uint32_t up = (uint32_t)__shfl_up((int)(uint32_t)(nlm >> 63), 1);
  const bool prev_nl = lane == 0 ? prev_byte == 0x0au : up != 0u;
  const unsigned long long starts = ((nlm << 1) | (prev_nl ? 1ull : 0ull)) & valid;

  RxCtx X;
  X.tilew = reinterpret_cast<const uint32_t*>(sp), X.cls = s_cls, X.fwd = reinterpret_cast<const uint16_t*>(s_dyn);
  X.trig = s_trig + wave * 64u, X.skip = P.rx_skip, X.cmask = P.rx_skip ? 0x7fu : 0xffu;
  X.rev = nullptr;
  X.cbase = cbase, X.toff = soff, X.L = L;
  X.fwd_start = P.rx_fwd_start, X.fwd_acc = P.rx_fwd_acc, X.rev_start = P.rx_rev_start, X.rev_acc = P.rx_rev_acc;
  X.tile_bytes = kRxSpan, X.nseg = 64u;
  X.la = reinterpret_cast<const uint8_t*>(law);

  uint32_t cnt = 0;
  uint64_t rank = 0;
  for (unsigned long long m = starts; m; m &= m - 1ull)
    cnt += rx_walk_line<false, LINES>(X, seg_off + (uint32_t)__builtin_ctzll(m), A, c, rank);
  if (__any(cnt != 0)) {  // tile_cnt is zero at rest and only a wave that found something writes
    const uint32_t wc = rx_wave_sum(cnt);
    if (lane == 0) atomicAdd(A.tile_cnt + tile, wc);
  }
}

// ---------------------------------------------------------------------------
// k_rx_chunk: expressions whose sets accept '\n' (`\s+`, `[^,]*`).  A match may span lines, so the only unit that
// is independent of its neighbours is the CHUNK -- the unit of the reference's own walk (Searcher.h:100-120 hands a
// chunk to _regex_byte_offsets).  One lane walks one chunk, start to end, with the same two automata, reading its
// bytes through the caches; parallelism is the number of chunks, so this route is slow (tens of MB/s per chunk) and
// exists so that such an expression is served at all.  Match tags only.  The count of a chunk goes to the tile_cnt
// word of the chunk's first tile (the ranks of the emit pass follow from the same prefix sum as everywhere else).
// ---------------------------------------------------------------------------
template <bool EMIT>
__global__ __launch_bounds__(kBlock) void k_rx_chunk(const ScanArgs A, const uint64_t nchunks) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];  // the forward table
  __shared__ uint8_t s_cls[256];
  const PatternDev P = A.pat;
  const uint32_t tid = threadIdx.x;
  s_cls[tid] = P.d_pat[tid];
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(P.d_pat + 256);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_dyn);
    for (uint32_t k = tid; k < (P.rx_fwd_n + 1u) / 2u; k += kBlock) dst[k] = src[k];
  }
  __syncthreads();
  const uint64_t c = (uint64_t)blockIdx.x * kBlock + tid;
  if (c >= nchunks) return;
  const uint64_t t0 = A.chunk_tile0[c];
  if (t0 == A.chunk_tile0[c + 1]) return;  // empty chunk: no tile, no word
  if (EMIT && A.tile_cnt[t0] == 0) return;
  const ChunkDev ch = A.chunks[c];
  const uint8_t* d = A.base + ch.offset;
  const uint64_t L = ch.length;
  const uint16_t* fwd = reinterpret_cast<const uint16_t*>(s_dyn);
  const uint16_t* rev = reinterpret_cast<const uint16_t*>(P.d_pat + ((256u + 2u * (size_t)P.rx_fwd_n + 15u) & ~(size_t)15u));
  uint64_t rank = EMIT ? A.tile_off[t0] : 0;
  uint32_t n = 0, hi = 0;
  uint64_t cur = 0;
  for (;;) {
    uint32_t st = P.rx_fwd_start;
    uint64_t q = cur, last_end = 0;
    while (q < L) {
      const uint32_t b = d[q];
      hi |= b;
      st = fwd[st + s_cls[b]];
      if (st == 0) break;
      ++q;
      if (st >= P.rx_fwd_acc) last_end = q;
    }
    if (!last_end) break;
    ++n;
    if (EMIT) {
      uint32_t rs = P.rx_rev_start;
      uint64_t r = last_end, start = last_end;
      while (r > cur) {
        rs = rev[rs + s_cls[d[r - 1]]];
        if (rs == 0) break;
        --r;
        if (rs >= P.rx_rev_acc) start = r;
      }
      if (A.m_cap == 0 || rank < A.m_cap) {
        A.m_pos[rank] = start;
        A.m_chunk[rank] = (uint32_t)c;
      }
      ++rank;
    }
    cur = last_end;
  }
  if (!EMIT) {
    if (n) A.tile_cnt[t0] = n;  // one writer per word; zero at rest
    // the bytes behind the last match were all read by the final forward scan, so `hi` has seen the whole chunk
    if (P.ascii_only && (hi & 0x80u)) atomicOr(A.flags, 1u);
  }
}

// newline counts per tile on their own (k_rx_chunk has no tile loop to fold them into)
__global__ __launch_bounds__(kBlock) void k_rx_newlines(const ScanArgs A) {
  __shared__ uint32_t s_w[kWaves];
  const uint64_t tile = (uint64_t)blockIdx.x + (uint64_t)blockIdx.y * gridDim.x;
  if (tile >= A.ntiles) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t c = A.tile_chunk ? A.tile_chunk[tile] : 0u;
  const ChunkDev ch = A.chunks[c];
  const uint8_t* cbase = A.base + ch.offset;
  const uint64_t L = ch.length, Lr = (L + 15u) & ~(uint64_t)15u;
  const uint64_t toff = (tile - A.chunk_tile0[c]) * (uint64_t)kRxTile;
  uint32_t cnt = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t off = toff + ((uint64_t)j * kBlock + tid) * kUnit;
    if (off >= Lr) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(cbase + off);
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint64_t o = off + 4u * q;
      const uint32_t keep = o >= L ? 0u : (o + 4u > L ? ((1u << (8u * (uint32_t)(L - o))) - 1u) : 0xffffffffu);
      const uint32_t x = (d[q] & keep) ^ 0x0a0a0a0au;  // bytes beyond L read as 0, which is not '\n'
      cnt += (uint32_t)__popc(~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu));
    }
  }
  const uint32_t wn = rx_wave_sum(cnt);
  if (lane == 0) s_w[wave] = wn;
  __syncthreads();
  if (tid == 0) A.tile_nl[tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// ---------------------------------------------------------------------------
// The prefilter route (xsg_regex.h: RegexDfa::prefix).  Every match starts with one of a few class sequences, so the
// scan kernel's class-sequence matcher finds the CANDIDATE positions at streaming speed (k_scan<kClass>, count +
// emit, xsg_api.cpp: rx_pre_matches) and the automaton only looks at those:
//   k_rx_verify   one lane per candidate: the ANCHORED forward automaton from the candidate -> the length of the
//                 leftmost-first match that starts exactly there, or 0.  (No reverse automaton: the start is known.)
//   k_rx_heads /  the reference's walk over the occurrences of a chunk (shift = match + match.size(),
//   k_rx_chains   search_wrappers.h:72-75): an occurrence is reported iff it starts at or behind the end of the
//                 last reported one -- resolved in parallel from a running maximum of the ends (below).
//   k_rx_compact  the reported occurrences, packed: from here on the list is what k_rx_scan's emit pass would
//                 have produced, and the shared list pipeline takes over.
// Works for expressions that can match a newline as well (the chunk is the unit of the walk, which k_rx_keep is).
// ---------------------------------------------------------------------------
constexpr uint64_t kRxVerifyBudget = 4096;
__global__ __launch_bounds__(kBlock) void k_rx_verify(const RxPreArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];  // the anchored table
  __shared__ uint8_t s_cls[256];
  const PatternDev P = A.pat;
  const uint32_t tid = threadIdx.x;
  s_cls[tid] = P.d_pat[tid] & (P.rx_skip ? 0x7fu : 0xffu);
  {
    const size_t rev_off = (256u + 2u * (size_t)P.rx_fwd_n + 15u) & ~(size_t)15u;
    const size_t anc_off = (rev_off + 2u * (size_t)P.rx_rev_n + 15u) & ~(size_t)15u;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(P.d_pat + anc_off);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_dyn);
    for (uint32_t k = tid; k < (P.rx_anc_n + 1u) / 2u; k += kBlock) dst[k] = src[k];
  }
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + tid;
  if (i >= A.n) return;
  const ChunkDev ch = A.chunks[A.c_chunk[i]];
  const uint8_t* d = A.base + ch.offset;
  const uint16_t* anc = reinterpret_cast<const uint16_t*>(s_dyn);
  const uint64_t p0 = A.c_pos[i];
  uint64_t q = p0, end = 0;
  uint32_t st = P.rx_anc_start;
  // A budget per candidate: an expression that runs far from every candidate (`aaa+` on a megabyte of `a`: every
  // position is a candidate and scans to the end of the run) would make this pass quadratic where walking the text
  // once (k_rx_scan, k_rx_chunk) is linear.  A candidate still alive after kRxVerifyBudget bytes raises flag bit 1
  // and the host redoes the search on the other route.
  const uint64_t stop_at = ch.length - p0 > kRxVerifyBudget ? p0 + kRxVerifyBudget : ch.length;
  // once some candidate has outrun its budget the whole pass is void (the host takes the other route): the waves that
  // start later see the flag and leave instead of scanning their 4 KiB each (M candidates x 4096 steps otherwise)
  if (__hip_atomic_load(A.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2u) {
    A.c_len[i] = 0u;
    return;
  }
  while (q < stop_at) {
    st = anc[st + s_cls[d[q]]];
    if (st == 0) break;
    ++q;
    if (st >= P.rx_anc_acc) end = q;
  }
  if (st != 0 && q < ch.length) atomicOr(A.flags, 2u);
  A.c_len[i] = end ? (uint32_t)(end - p0) : 0u;
}

// ---- the walk over the occurrences, in parallel ------------------------------------------------------------------
// An occurrence is reported iff it starts at or behind the end of the last reported one.  Sequential as written, but
// an occurrence that starts at or behind the end of EVERY earlier occurrence (reported or not) is reported whatever
// happened before it, and the walk behind it does not depend on anything before it: a HEAD.  So: the running maximum
// of the ends (one max-scan over all candidates; keys carry the chunk number in their upper bits, so a chunk's
// occurrences never cover those of the next), heads from it, and the thread of each head walks its chain -- the
// occurrences up to the next head, usually none or one.
constexpr int kRxScanItems = 8;
constexpr uint64_t kRxScanBlock = (uint64_t)kBlock * kRxScanItems;
__device__ __forceinline__ uint64_t rx_key(uint32_t chunk, uint64_t off) { return ((uint64_t)chunk << 40) | off; }  // off < 2^40
__device__ __forceinline__ uint64_t rx_end_key(const RxPreArgs& A, uint64_t i) {
  const uint32_t len = A.c_len[i];
  return len ? rx_key(A.c_chunk[i], A.c_pos[i] + len) : 0ull;
}
__device__ __forceinline__ uint64_t rx_block_max(uint64_t v, uint64_t* sh) {  // max over the block, valid in every thread
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const uint64_t o = (uint64_t)__shfl_xor((long long)v, s);
    v = o > v ? o : v;
  }
  if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  uint64_t m = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) m = sh[w] > m ? sh[w] : m;
  __syncthreads();
  return m;
}
__global__ __launch_bounds__(kBlock) void k_rx_block_max(const RxPreArgs A, uint64_t* block_max) {
  __shared__ uint64_t sh[kWaves];
  const uint64_t b0 = (uint64_t)blockIdx.x * kRxScanBlock + (uint64_t)threadIdx.x * kRxScanItems;
  uint64_t v = 0;
#pragma unroll
  for (int k = 0; k < kRxScanItems; ++k)
    if (b0 + k < A.n) {
      const uint64_t e = rx_end_key(A, b0 + k);
      v = e > v ? e : v;
    }
  const uint64_t m = rx_block_max(v, sh);
  if (threadIdx.x == 0) block_max[blockIdx.x] = m;
}
// exclusive running maximum of one value per thread over the block (in thread order); *total = the block's maximum
__device__ __forceinline__ uint64_t rx_block_excl_max(uint64_t v, uint64_t* sh, uint64_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint64_t incl = v;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const uint64_t o = (uint64_t)__shfl_up((long long)incl, s);
    if (lane >= (uint32_t)s) incl = o > incl ? o : incl;
  }
  uint64_t excl = (uint64_t)__shfl_up((long long)incl, 1);
  if (lane == 0) excl = 0;
  if (lane == 63u) sh[wave] = incl;
  __syncthreads();
  uint64_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    if ((uint32_t)w < wave) before = sh[w] > before ? sh[w] : before;
    all = sh[w] > all ? sh[w] : all;
  }
  __syncthreads();
  *total = all;
  return excl > before ? excl : before;
}
__global__ __launch_bounds__(kBlock) void k_rx_block_max_scan(uint64_t* block_max, uint64_t nb) {  // in place, one workgroup
  __shared__ uint64_t sh[kWaves];
  uint64_t carry = 0;
  for (uint64_t i0 = 0; i0 < nb; i0 += kBlock) {
    const uint64_t i = i0 + threadIdx.x;
    const uint64_t v = i < nb ? block_max[i] : 0ull;
    uint64_t tot;
    const uint64_t ex = rx_block_excl_max(v, sh, &tot);
    if (i < nb) block_max[i] = ex > carry ? ex : carry;
    carry = tot > carry ? tot : carry;
  }
}
// heads: flagged in `head` (1 / 0); c_keep is cleared for the chain walk
__global__ __launch_bounds__(kBlock) void k_rx_heads(const RxPreArgs A, const uint64_t* block_max, uint32_t* head) {
  __shared__ uint64_t sh[kWaves];
  const uint64_t b0 = (uint64_t)blockIdx.x * kRxScanBlock + (uint64_t)threadIdx.x * kRxScanItems;
  uint64_t e[kRxScanItems], tmax = 0;
#pragma unroll
  for (int k = 0; k < kRxScanItems; ++k) {
    e[k] = b0 + k < A.n ? rx_end_key(A, b0 + k) : 0ull;
    tmax = e[k] > tmax ? e[k] : tmax;
  }
  uint64_t tot;
  uint64_t run = rx_block_excl_max(tmax, sh, &tot);
  const uint64_t bm = block_max[blockIdx.x];
  run = bm > run ? bm : run;
#pragma unroll
  for (int k = 0; k < kRxScanItems; ++k) {
    if (b0 + k < A.n) {
      head[b0 + k] = (e[k] != 0 && rx_key(A.c_chunk[b0 + k], A.c_pos[b0 + k]) >= run) ? 1u : 0u;
      A.c_keep[b0 + k] = 0u;
    }
    run = e[k] > run ? e[k] : run;
  }
}
__global__ void k_rx_chains(const RxPreArgs A, const uint32_t* head) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= A.n || !head[i]) return;
  uint64_t last_end = rx_end_key(A, i);
  A.c_keep[i] = 1u;
  for (uint64_t j = i + 1; j < A.n && !head[j]; ++j) {  // up to the next head: its own thread's business
    const uint64_t e = rx_end_key(A, j);
    if (e != 0 && rx_key(A.c_chunk[j], A.c_pos[j]) >= last_end) {
      A.c_keep[j] = 1u;
      last_end = e;
    }
  }
}

__global__ void k_rx_compact(const RxPreArgs A) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= A.n || !A.c_keep[i]) return;
  const uint64_t dst = A.c_pre[i];
  A.m_pos[dst] = A.c_pos[i];
  A.m_chunk[dst] = A.c_chunk[i];
}

hipError_t launch_rx_verify_keep(const RxPreArgs& a, hipStream_t s) {
  if (a.n) {
    const size_t dyn = ((size_t)a.pat.rx_anc_n * 2u + 15u) & ~(size_t)15u;
    hipLaunchKernelGGL(k_rx_verify, dim3((unsigned)((a.n + kBlock - 1) / kBlock)), dim3(kBlock), dyn, s, a);
    const uint64_t nb = (a.n + kRxScanBlock - 1) / kRxScanBlock;
    hipLaunchKernelGGL(k_rx_block_max, dim3((unsigned)nb), dim3(kBlock), 0, s, a, a.scan_tmp);
    uint32_t* head = reinterpret_cast<uint32_t*>(const_cast<uint64_t*>(a.c_pre));  // free until the scan of c_keep writes it
    hipLaunchKernelGGL(k_rx_block_max_scan, dim3(1), dim3(kBlock), 0, s, a.scan_tmp, nb);
    hipLaunchKernelGGL(k_rx_heads, dim3((unsigned)nb), dim3(kBlock), 0, s, a, (const uint64_t*)a.scan_tmp, head);
    hipLaunchKernelGGL(k_rx_chains, dim3((unsigned)((a.n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a, (const uint32_t*)head);
  }
  return hipGetLastError();
}
__global__ void k_rx_mark_tiles(const ListArgs A, uint32_t* mask) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= A.M || !A.keep[i]) return;  // kept = the first factor occurrence of its line; m_ls = where that line starts
  mask[A.chunk_tile0[A.m_chunk[i]] + A.m_ls[i] / kRxTile] = 1u;
}
hipError_t launch_rx_mark_tiles(const ListArgs& a, uint32_t* mask, hipStream_t s) {
  if (a.M) hipLaunchKernelGGL(k_rx_mark_tiles, dim3((unsigned)((a.M + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a, mask);
  return hipGetLastError();
}

hipError_t launch_rx_compact(const RxPreArgs& a, hipStream_t s) {
  if (a.n) hipLaunchKernelGGL(k_rx_compact, dim3((unsigned)((a.n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

static dim3 rx_grid(uint64_t ntiles) {
  const uint64_t maxx = 1u << 30;
  if (ntiles <= maxx) return dim3((unsigned)ntiles, 1, 1);
  return dim3((unsigned)maxx, (unsigned)((ntiles + maxx - 1) / maxx), 1);
}

static size_t rx_dyn_lds(const ScanArgs& a) { return ((size_t)a.pat.rx_fwd_n * 2u + 15u) & ~(size_t)15u; }

hipError_t launch_rx_count(const ScanArgs& a, bool want_nl, bool want_lines, hipStream_t s) {
  if (a.ntiles == 0) return hipSuccess;
  if (a.tile_bytes != kRxTile) return hipErrorInvalidValue;
  const dim3 grid = rx_grid(a.ntiles);
  if (a.pat.rx_multiline) {
    if (want_lines) return hipErrorInvalidValue;  // refused upstream (has_newline)
    if (want_nl) hipLaunchKernelGGL(k_rx_newlines, grid, dim3(kBlock), 0, s, a);
    const uint64_t nchunks = a.nchunks;
    hipLaunchKernelGGL((k_rx_chunk<false>), dim3((unsigned)((nchunks + kBlock - 1) / kBlock)), dim3(kBlock), rx_dyn_lds(a), s,
                       a, nchunks);
    return hipGetLastError();
  }
  // the count passes: one wave per 4 KiB span (k_rx_count); XSG_RX_WAVE=0: the tile-cooperative kernel of round 2/3a
  static const bool by_wave = [] { const char* e = getenv("XSG_RX_WAVE"); return !(e && *e == '0'); }();
  if (by_wave) {
    if (want_nl) {  // the waves of a tile add their newline counts
      const hipError_t e = hipMemsetAsync(a.tile_nl, 0, 4 * a.ntiles, s);
      if (e != hipSuccess) return e;
    }
    if (want_lines)
      hipLaunchKernelGGL((k_rx_count<true>), grid, dim3(kBlock), rx_dyn_lds(a), s, a, want_nl ? 1u : 0u);
    else
      hipLaunchKernelGGL((k_rx_count<false>), grid, dim3(kBlock), rx_dyn_lds(a), s, a, want_nl ? 1u : 0u);
    return hipGetLastError();
  }
  if (want_lines)
    hipLaunchKernelGGL((k_rx_scan<false, true>), grid, dim3(kBlock), rx_dyn_lds(a), s, a, want_nl ? 1u : 0u);
  else
    hipLaunchKernelGGL((k_rx_scan<false, false>), grid, dim3(kBlock), rx_dyn_lds(a), s, a, want_nl ? 1u : 0u);
  return hipGetLastError();
}

hipError_t launch_rx_emit(const ScanArgs& a, hipStream_t s) {
  if (a.ntiles == 0) return hipSuccess;
  if (a.tile_bytes != kRxTile) return hipErrorInvalidValue;
  if (a.pat.rx_multiline) {
    const uint64_t nchunks = a.nchunks;
    hipLaunchKernelGGL((k_rx_chunk<true>), dim3((unsigned)((nchunks + kBlock - 1) / kBlock)), dim3(kBlock), rx_dyn_lds(a), s, a,
                       nchunks);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((k_rx_scan<true, false>), rx_grid(a.ntiles), dim3(kBlock), rx_dyn_lds(a), s, a, 0u);
  return hipGetLastError();
}


// Loading a code object costs milliseconds the first time one of its kernels is launched: xsg_ctx_create launches this
// empty kernel of every kernel file, so that the first search of a process does not pay for it (5.5 ms of the first
// xsg_count, scripts/first_call.py).
__global__ void k_warm_rx() {}
hipError_t warm_rx_kernels(hipStream_t s) {
  hipLaunchKernelGGL(k_warm_rx, dim3(1), dim3(1), 0, s);
  return hipGetLastError();
}

}  // namespace xsg
