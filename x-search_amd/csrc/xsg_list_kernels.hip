// xsg_list_kernels.hip -- everything behind the bulk scan for the list tags (xs::match_byte_offsets,
// xs::line_byte_offsets, xs::line_indices, xs::lines): ranks of the per-tile counts, which occurrences the reference
// walk reports, the end-of-chunk walk, and the outputs.  What they replace in the reference:
//   search_wrappers.h:29-52 (_byte_offsets), :111-123 (previous_new_line_offset_relative_to_match),
//   :136-154 (byte_offsets_match / _line), :187-207 (line); simd_search.cpp:324-336 (greedy non-overlap).
#include <algorithm>

#include "xsg_devutil.h"
#include "xsg_tail.h"

namespace xsg {

// ---------------------------------------------------------------------------
// exclusive scan (uint32 or uint64 in -> uint64 out), three small kernels
// ---------------------------------------------------------------------------
constexpr int kScanItems = 8;
constexpr uint64_t kScanBlockElems = (uint64_t)kBlock * kScanItems;

uint64_t scan_tmp_elems(uint64_t n) { return (n + kScanBlockElems - 1) / kScanBlockElems + 1; }

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, uint32_t lane) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    uint64_t o = (uint64_t)__shfl_up((long long)v, s);
    if (lane >= (uint32_t)s) v += o;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, *total = block sum
__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t* sh, uint64_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t incl = wave_incl_scan_u64(v, lane);
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    if ((uint32_t)w < wave) base += sh[w];
    tot += sh[w];
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// UINT64_MAX in a uint64 input is the "dropped line" marker of k_line_lengths: it scans as 0
__device__ __forceinline__ uint64_t scan_val(uint32_t v) { return v; }
__device__ __forceinline__ uint64_t scan_val(uint64_t v) { return v == UINT64_MAX ? 0 : v; }

template <typename T>
__global__ __launch_bounds__(kBlock) void k_scan_block_sums(const T* in, uint64_t n, uint64_t* block_sums) {
  __shared__ uint64_t sh[kWaves];
  const uint64_t b0 = (uint64_t)blockIdx.x * kScanBlockElems + (uint64_t)threadIdx.x * kScanItems;
  uint64_t v = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (b0 + k < n) v += scan_val(in[b0 + k]);
  uint64_t tot;
  block_excl_scan_u64(v, sh, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block_sums in place (nb entries), total appended at [nb]
__global__ __launch_bounds__(kBlock) void k_scan_top(uint64_t* block_sums, uint64_t nb) {
  __shared__ uint64_t sh[kWaves];
  uint64_t carry = 0;
  for (uint64_t i0 = 0; i0 < nb; i0 += kBlock) {
    const uint64_t i = i0 + threadIdx.x;
    const uint64_t v = i < nb ? block_sums[i] : 0;
    uint64_t tot;
    const uint64_t ex = block_excl_scan_u64(v, sh, &tot);
    if (i < nb) block_sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) block_sums[nb] = carry;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_scan_write(const T* in, uint64_t n, const uint64_t* block_sums,
                                                       uint64_t* out) {
  __shared__ uint64_t sh[kWaves];
  const uint64_t b0 = (uint64_t)blockIdx.x * kScanBlockElems + (uint64_t)threadIdx.x * kScanItems;
  uint64_t x[kScanItems];
  uint64_t v = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    x[k] = b0 + k < n ? scan_val(in[b0 + k]) : 0;
    v += x[k];
  }
  uint64_t tot;
  uint64_t run = block_sums[blockIdx.x] + block_excl_scan_u64(v, sh, &tot);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (b0 + k < n) out[b0 + k] = run;
    run += x[k];
  }
  // the grand total goes to out[n]
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) out[n] = block_sums[gridDim.x];
}

__global__ void k_set_u64(uint64_t* p, uint64_t v) { *p = v; }

template <typename T>
static hipError_t exclusive_scan_impl(const T* in, uint64_t* out, uint64_t n, uint64_t* tmp, hipStream_t s) {
  if (n == 0) {
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, s, out, (uint64_t)0);
    return hipGetLastError();
  }
  const uint64_t nb = (n + kScanBlockElems - 1) / kScanBlockElems;
  hipLaunchKernelGGL((k_scan_block_sums<T>), dim3((unsigned)nb), dim3(kBlock), 0, s, in, n, tmp);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kBlock), 0, s, tmp, nb);
  hipLaunchKernelGGL((k_scan_write<T>), dim3((unsigned)nb), dim3(kBlock), 0, s, in, n, tmp, out);
  return hipGetLastError();
}

hipError_t launch_exclusive_scan_u32(const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, hipStream_t s) {
  return exclusive_scan_impl<uint32_t>(in, out, n, tmp, s);
}
hipError_t launch_exclusive_scan_u64(const uint64_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, hipStream_t s) {
  return exclusive_scan_impl<uint64_t>(in, out, n, tmp, s);
}

// ---------------------------------------------------------------------------
// The same prefix sums in TWO launches and with the element count on the device (the one-sync list route): the
// last workgroup of the first launch to arrive (a ticket) scans the block sums and publishes the total -- to a
// device word the following kernels read, and to a pinned host mirror the host reads after its one sync.  With
// HITS the second launch also writes the ordered list of the indices whose entry is not zero (the tiles that
// hold a match: the emit pass visits only those).
// ---------------------------------------------------------------------------
uint64_t scan2_tmp_elems(uint64_t n_cap) { return 2 * ((n_cap + kScanBlockElems - 1) / kScanBlockElems + 1); }

__device__ __forceinline__ uint64_t scan2_count(const Scan2Args& A) {
  if (!A.n_dev) return A.n_cap;
  const uint64_t n = *A.n_dev;
  return n < A.n_cap ? n : A.n_cap;
}

// Both kernels stride over VIRTUAL blocks of kScanBlockElems entries (as many as the count on the device needs), so the
// grid can be much smaller than the capacity asks for: a list of a few thousand entries in arrays of a million is
// four virtual blocks, and a ticket over 64 workgroups is drawn faster than one over 512.
template <typename T, bool HITS>
__global__ __launch_bounds__(kBlock) void k_scan2_a(const Scan2Args A) {
  __shared__ uint64_t sh[kWaves];
  __shared__ uint32_t s_is_last;
  const T* in = static_cast<const T*>(A.in);
  const uint64_t n = scan2_count(A);
  const uint64_t nb_cap = (A.n_cap + kScanBlockElems - 1) / kScanBlockElems;  // layout of the scratch
  const uint64_t nvb = (n + kScanBlockElems - 1) / kScanBlockElems;
  uint64_t* blk_sum = A.blk;
  uint64_t* blk_hit = A.blk + (nb_cap + 1);
  for (uint64_t vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    const uint64_t b0 = vb * kScanBlockElems + (uint64_t)threadIdx.x * kScanItems;
    uint64_t v = 0, h = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (b0 + k < n) {
        const uint64_t x = scan_val(in[b0 + k]);
        v += x;
        if (HITS) h += x != 0;
      }
    }
    uint64_t tot;
    block_excl_scan_u64(v, sh, &tot);
    uint64_t toth = 0;
    if (HITS) block_excl_scan_u64(h, sh, &toth);
    if (threadIdx.x == 0) {
      blk_sum[vb] = tot;
      if (HITS) blk_hit[vb] = toth;
    }
  }
  if (threadIdx.x == 0) {
    __threadfence();
    s_is_last = atomicAdd(A.ticket, 1u) == gridDim.x - 1u;
  }
  __syncthreads();
  if (!s_is_last) return;
  __threadfence();
  // exclusive scan of the block sums in place, the totals behind them
  uint64_t carry = 0, carryh = 0;
  for (uint64_t i0 = 0; i0 < nvb; i0 += kBlock) {
    const uint64_t i = i0 + threadIdx.x;
    const uint64_t x = i < nvb ? __hip_atomic_load(blk_sum + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    uint64_t t;
    const uint64_t ex = block_excl_scan_u64(x, sh, &t);
    if (i < nvb) blk_sum[i] = carry + ex;
    carry += t;
    if (HITS) {
      const uint64_t y = i < nvb ? __hip_atomic_load(blk_hit + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      const uint64_t exh = block_excl_scan_u64(y, sh, &t);
      if (i < nvb) blk_hit[i] = carryh + exh;
      carryh += t;
    }
  }
  if (threadIdx.x == 0) {
    A.out[n] = carry;
    if (A.tot_dev) *A.tot_dev = carry;
    if (A.tot_host) *A.tot_host = carry;
    if (HITS) {
      if (A.hits_dev) *A.hits_dev = carryh;
      if (A.hits_host) *A.hits_host = carryh;
    }
    if (A.ovf_dev) {
      const uint64_t over = (carry > A.total_cap || (HITS && carryh > A.hit_cap)) ? A.ovf_bit : 0u;
      const uint64_t w = A.ovf_init ? over : (*A.ovf_dev | over);
      *A.ovf_dev = w;
      if (A.ovf_host) *A.ovf_host = w;
    }
    *A.ticket = 0u;  // at rest for the next launch (stream order)
  }
}

template <typename T, bool HITS>
__global__ __launch_bounds__(kBlock) void k_scan2_b(const Scan2Args A) {
  __shared__ uint64_t sh[kWaves];
  const T* in = static_cast<const T*>(A.in);
  const uint64_t n = scan2_count(A);
  const uint64_t nb_cap = (A.n_cap + kScanBlockElems - 1) / kScanBlockElems;
  const uint64_t nvb = (n + kScanBlockElems - 1) / kScanBlockElems;
  for (uint64_t vb = blockIdx.x; vb < nvb; vb += gridDim.x) {  // (workgroup-uniform)
    const uint64_t b0 = vb * kScanBlockElems + (uint64_t)threadIdx.x * kScanItems;
    uint64_t x[kScanItems];
    uint64_t v = 0, h = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      x[k] = b0 + k < n ? scan_val(in[b0 + k]) : 0;
      v += x[k];
      if (HITS) h += x[k] != 0;
    }
    uint64_t tot;
    uint64_t run = A.blk[vb] + block_excl_scan_u64(v, sh, &tot);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (b0 + k < n) A.out[b0 + k] = run;
      run += x[k];
    }
    if (HITS) {
      uint64_t hr = A.blk[(nb_cap + 1) + vb] + block_excl_scan_u64(h, sh, &tot);
#pragma unroll
      for (int k = 0; k < kScanItems; ++k) {
        if (x[k] != 0) {
          if (hr < A.hit_cap) A.hit_idx[hr] = (uint32_t)(b0 + k);
          ++hr;
        }
      }
    }
  }
}

// grid: one workgroup per virtual block when the count is the host's (n_dev null: the per-tile arrays), a bounded
// number when it lives on the device and is, on this route, far below the capacity
template <typename T, bool HITS>
static hipError_t scan2_impl(const Scan2Args& a, hipStream_t s) {
  uint64_t nb = std::max<uint64_t>(1, (a.n_cap + kScanBlockElems - 1) / kScanBlockElems);
  nb = std::min<uint64_t>(nb, a.n_dev ? 64 : 2048);
  hipLaunchKernelGGL((k_scan2_a<T, HITS>), dim3((unsigned)nb), dim3(kBlock), 0, s, a);
  hipLaunchKernelGGL((k_scan2_b<T, HITS>), dim3((unsigned)nb), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_scan2_u32(const Scan2Args& a, bool hits, hipStream_t s) {
  return hits ? scan2_impl<uint32_t, true>(a, s) : scan2_impl<uint32_t, false>(a, s);
}
hipError_t launch_scan2_u64(const Scan2Args& a, hipStream_t s) { return scan2_impl<uint64_t, false>(a, s); }

// ---------------------------------------------------------------------------
// list post-processing: one thread per raw match / per chunk.  Matches are
// sparse at text densities (~5e-7 per byte), so these are latency-, not
// bandwidth-bound and deliberately simple.
// ---------------------------------------------------------------------------
static inline dim3 grid_for(uint64_t n) {
  uint64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

// grids of the one-sync route cover a CAPACITY (the count is on the device): a bounded number of workgroups that
// stride over the entries there are
static inline dim3 grid_capped(uint64_t n) {
  const dim3 g = grid_for(n);
  return dim3(g.x < 256u ? g.x : 256u);
}

// number of raw entries the list kernels work on: the host's value, or the device's bounded by the arrays' capacity
__device__ __forceinline__ uint64_t list_count(const ListArgs& A) {
  if (!A.M_dev) return A.M;
  const uint64_t m = *A.M_dev;
  return m < A.M ? m : A.M;
}

__global__ void k_keep_all(const ListArgs A) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < list_count(A)) A.keep[i] = 1u;
}

constexpr uint64_t kChainBudget = 4096;  // entries one thread walks before the chain is handed to the parallel passes
constexpr uint32_t kNoLink = 0xffffffffu;

// Greedy non-overlap (shift = match + plen, simd_search.cpp:333 /
// search_wrappers.h:42): only patterns with a border can overlap themselves.
// An occurrence >= plen after its predecessor is always kept and starts a
// chain; the thread owning a chain head walks its chain.
__global__ void k_greedy_keep(const ListArgs A) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t M = list_count(A);
  if (i >= M) return;
  const uint32_t plen = A.pat.plen;
  const uint32_t c = A.m_chunk[i];
  const bool head = i == 0 || A.m_chunk[i - 1] != c || A.m_pos[i] - A.m_pos[i - 1] >= plen;
  if (!head) return;
  A.keep[i] = 1u;
  uint64_t last = A.m_pos[i], prev = last;
  // The walk is a chain of dependent decisions, but not of dependent LOADS: eight entries are fetched at a time
  // (independent loads, one latency) and decided from registers -- a long chain (a run of one byte searched for
  // `aa` is one chain per chunk) moves at ~6 ns per occurrence instead of ~50.
  // With A.long_flag set the thread gives up after kChainBudget entries and raises the flag: what it has marked
  // stands (every mark is a reported occurrence), the rest of the chain is resolved by pointer jumping
  // (k_greedy_links / k_greedy_jump) -- a multi-megabyte run of one byte is ONE chain and would be one lane's work.
  for (uint64_t j = i + 1; j < M;) {
    if (A.long_flag && j - i > kChainBudget) {
      *A.long_flag = 1u;
      break;
    }
    uint64_t p[8];
    uint32_t ch[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint64_t idx = j + (uint64_t)u < M ? j + (uint64_t)u : M - 1;
      p[u] = A.m_pos[idx];
      ch[u] = A.m_chunk[idx];
    }
    bool done = false;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (done || j + (uint64_t)u >= M) continue;
      if (ch[u] != c || p[u] - prev >= plen) {  // the chain ends: the next occurrence is a head of its own
        done = true;
        continue;
      }
      const bool k = p[u] >= last + plen;
      A.keep[j + (uint64_t)u] = k ? 1u : 0u;
      if (k) last = p[u];
      prev = p[u];
    }
    if (done) break;
    j += 8;
  }
}

// ---- long chains, in parallel ------------------------------------------------------------------------------
// nxt(i) = the first occurrence of i's chunk that starts at or behind the END of occurrence i: if i is reported, nxt(i)
// is the next one reported (shift = match + plen).  The reported set is the closure of the chain heads under nxt.
// Pointer jumping: J = nxt; every round marks J(i) for every marked i and squares J (J2 = J o J), so after round k the
// closure holds nxt^m(head) for all m < 2^(k+1): log2(chain length) rounds of two coalesced passes each instead of
// one lane walking the chain.  A mark is only ever set on an element of the closure (a marked element's image under
// any power of nxt is reported too), so marks of an unfinished sequential walk, of earlier rounds and of this round
// mix freely, and a round that marks nothing new means the closure is complete.
__global__ void k_greedy_links(const ListArgs A, uint32_t* J) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t M = list_count(A);
  if (i >= M) return;
  const uint32_t plen = A.pat.plen;
  const uint32_t c = A.m_chunk[i];
  const uint64_t want = A.m_pos[i] + plen;
  // occurrences are distinct offsets, so nxt(i) <= i + plen: lower bound in (i, i + plen] of "another chunk, or far enough"
  uint64_t lo = i + 1, hi = i + 1 + plen < M ? i + 1 + plen : M;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    const bool f = A.m_chunk[mid] != c || A.m_pos[mid] >= want;
    if (f) hi = mid; else lo = mid + 1;
  }
  J[i] = (lo < M && A.m_chunk[lo] == c) ? (uint32_t)lo : kNoLink;
}

__global__ void k_greedy_jump(const ListArgs A, const uint32_t* J, uint32_t* J2, uint32_t* changed) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= list_count(A)) return;
  const uint32_t j = J[i];
  uint32_t jj = kNoLink;
  if (j != kNoLink) {
    if (A.keep[i] && !A.keep[j]) {
      A.keep[j] = 1u;
      *changed = 1u;
    }
    jj = J[j];
  }
  J2[i] = jj;
}

// ---- the line tags of a literal that contains '\n' -------------------------------------------------------------
// The reference's walk (search_wrappers.h:29-50, 163-207 with skip_to_nl) is defined for any std::string: report the
// leftmost occurrence at or behind `shift`, then shift = the byte behind the first '\n' at or behind the match's END.
// When the pattern holds no newline that is "the first occurrence of every line" (k_line_starts_keep).  When it does,
// an occurrence reaches into the next line(s) and the walk is a chain: the chunk's first occurrence is reported, and
// nxt(i) = the first occurrence that starts behind the first '\n' at or behind i's end.  Here: every raw occurrence
// gets its line start as the reference computes it (behind the last '\n' BEFORE the match -- or, when the match's first
// byte is itself a '\n', one byte behind that: previous_new_line_offset_relative_to_match tests the match's first
// byte first and wraps, :111-123), the chunk heads are marked, and J = nxt; the closure under nxt is then taken by the
// pointer jumping that resolves long greedy chains (k_greedy_jump: log2(chain) rounds).
__global__ __launch_bounds__(kBlock) void k_nlpat_links(const ListArgs A, uint32_t* J) {
  const uint64_t M = list_count(A);
  const uint32_t lane = threadIdx.x & 63u;
  for (uint64_t i0 = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); i0 < M; i0 += (uint64_t)gridDim.x * kBlock) {
    const uint64_t i = i0 + lane;
    const bool live = i < M;  // no early exit of a lane: the wave finishes long newline searches together
    const uint32_t c = live ? A.m_chunk[i] : 0u;
    const ChunkDev ch = A.chunks[c];
    const uint8_t* d = A.base + ch.offset;
    const uint64_t pos = live ? A.m_pos[i] : 0;
    const int64_t before = newline_query<false>(live && !A.pat.nl_first, d, 0, pos, lane);
    const int64_t behind = newline_query<true>(live, d, pos + A.pat.plen, ch.length, lane);
    if (!live) continue;
    A.m_ls[i] = A.pat.nl_first ? pos + 1u : before < 0 ? 0u : (uint64_t)before + 1u;
    A.keep[i] = (i == 0 || A.m_chunk[i - 1] != c) ? 1u : 0u;
    uint32_t nxt = kNoLink;
    if (behind >= 0) {  // (no newline behind the match: the walk ends with it)
      uint64_t r1 = A.tile_off[A.chunk_tile0[c + 1]];
      r1 = r1 < M ? r1 : M;
      uint64_t lo = i + 1, hi = r1;
      while (lo < hi) {  // first occurrence of the chunk that starts behind that newline
        const uint64_t mid = (lo + hi) >> 1;
        if (A.m_pos[mid] > (uint64_t)behind) hi = mid; else lo = mid + 1;
      }
      if (lo < r1) nxt = (uint32_t)lo;
    }
    J[i] = nxt;
  }
}

hipError_t launch_nlpat_links(const ListArgs& a, uint32_t* J, hipStream_t s) {
  if (!a.M) return hipSuccess;
  hipLaunchKernelGGL(k_nlpat_links, grid_for(a.M), dim3(kBlock), 0, s, a, J);
  return hipGetLastError();
}

hipError_t launch_greedy_links(const ListArgs& a, uint32_t* J, hipStream_t s) {
  if (!a.M) return hipSuccess;
  hipLaunchKernelGGL(k_greedy_links, grid_for(a.M), dim3(kBlock), 0, s, a, J);
  return hipGetLastError();
}
hipError_t launch_greedy_jump(const ListArgs& a, const uint32_t* J, uint32_t* J2, uint32_t* changed, hipStream_t s) {
  if (!a.M) return hipSuccess;
  hipLaunchKernelGGL(k_greedy_jump, grid_for(a.M), dim3(kBlock), 0, s, a, J, J2, changed);
  return hipGetLastError();
}

// Which raw matches survive the skip_to_nl walk, and their line starts
// (search_wrappers.h:111-123,149-154).  A match is the first of its line iff a
// '\n' lies between the previous match and it, so every thread scans back only as
// far as the previous match (the first match of a chunk: to the chunk start).  The
// scans of one chunk are disjoint: O(chunk) bytes in total however long the lines
// are (a walk back to the line start per match would be quadratic on one huge line).
__global__ __launch_bounds__(kBlock) void k_line_starts_keep(const ListArgs A) {
  const uint64_t M = list_count(A);
  // grid-stride by whole waves (the one-sync route sizes the grid by the arrays' capacity, not by M); no early
  // return for a lane without an entry: the wave finishes long scans together (newline_query)
  for (uint64_t i0 = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); i0 < M; i0 += (uint64_t)gridDim.x * kBlock) {
    const uint64_t i = i0 + (threadIdx.x & 63u);
    const bool live = i < M;
    const uint32_t c = live ? A.m_chunk[i] : 0u;
    const uint8_t* d = A.base + A.chunks[c].offset;
    const bool first_in_chunk = live && (i == 0 || A.m_chunk[i - 1] != c);
    const uint64_t lo = (!live || first_in_chunk) ? 0 : A.m_pos[i - 1];
    const uint64_t hi = live ? A.m_pos[i] : 0;
    // the newline that opens the match's line, if it lies in [lo, match)
    const int64_t nl = newline_query<false>(live, d, lo, hi, threadIdx.x & 63u);
    if (!live) continue;
    const bool found_nl = nl >= 0;
    A.keep[i] = (found_nl || first_in_chunk) ? 1u : 0u;
    A.m_ls[i] = found_nl ? (uint64_t)nl + 1u : lo;  // meaningful for kept matches only (0 for a first match on the chunk's first line)
  }
}

// per chunk: where the walk enters the tail zone, from the last kept bulk match
__global__ __launch_bounds__(kBlock) void k_chunk_shift0(const ListArgs A) {
  const uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = c < A.nchunks;
  uint64_t last_end = 0;
  ChunkDev ch{};
  if (live) {
    const uint64_t M = list_count(A);
    uint64_t r0 = A.tile_off[A.chunk_tile0[c]], r1 = A.tile_off[A.chunk_tile0[c + 1]];
    r0 = r0 < M ? r0 : M;  // bounded emission: entries beyond the capacity do not exist (the total is refused later)
    r1 = r1 < M ? r1 : M;
    // last kept raw match of the chunk (kept ones are never far from the end of a chain)
    for (uint64_t i = r1; i > r0; --i) {
      if (A.keep[i - 1]) {
        last_end = A.m_pos[i - 1] + A.pat.plen;
        break;
      }
    }
    ch = A.chunks[c];
  }
  // line modes: the walk continues at the start of the line after the last kept match (UINT64_MAX: no such line)
  const bool need_nl = live && last_end != 0 && A.line_mode != 0;
  const int64_t nl = newline_query<true>(need_nl, A.base + ch.offset, last_end, ch.length, threadIdx.x & 63u);
  if (!live) return;
  A.chunk_shift0[c] = last_end == 0 ? 0 : !A.line_mode ? last_end : nl < 0 ? UINT64_MAX : (uint64_t)nl + 1u;
}

__global__ void k_tail_list(const ListArgs A) {
  const uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (c >= A.nchunks) return;
  uint32_t n = 0;
  if (!A.pat.exact_tail && A.pat.plen > 1) {
    const ChunkDev ch = A.chunks[c];
    n = tail_walk(A.base + ch.offset, ch.length, A.pat.d_pat, A.pat.plen, A.chunk_shift0[c], A.line_mode != 0,
                  A.tail_pos + c * A.tail_cap, A.tail_cap, A.pat.icase != 0);
  }
  A.tail_cnt[c] = n;
}

// ---- one-sync route: k_chunk_shift0 + k_tail_list + the prefix of the tail counts, one launch ------------------
// One WAVE per chunk: where the reference walk stands when it reaches the chunk's tail zone (behind the last raw
// occurrence it reports; line modes: at the start of the line after it), then the zone itself on bit masks
// (wave_tail_counts: the zone's bytes in LDS, one position per lane, the walk on the scalar unit -- the
// one-thread-per-chunk walk through global memory, k_tail_list, takes 35-40 us whatever the number of chunks).
// The last workgroup to arrive turns tail_cnt into its exclusive prefix and publishes kTotKept / kTotFinal.
__global__ __launch_bounds__(kBlock) void k_chunk_tail(const ListArgs A) {
  __shared__ uint64_t sh[kWaves];
  __shared__ __attribute__((aligned(16))) uint8_t s_zone[kWaves][kZoneStage];
  __shared__ uint8_t s_pat[kTailMaskMaxPlen + 3];
  __shared__ uint32_t s_is_last;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t plen = A.pat.plen;
  const bool need_tail = !A.pat.exact_tail && plen > 1;
  const bool mask_tail = need_tail && plen <= kTailMaskMaxPlen;
  if (mask_tail) {
    if (threadIdx.x < plen) s_pat[threadIdx.x] = A.pat.d_pat[threadIdx.x];
    __syncthreads();
  }
  const uint64_t M = list_count(A);
  const uint64_t nwaves = (uint64_t)gridDim.x * kWaves;
  for (uint64_t c = (uint64_t)blockIdx.x * kWaves + wave; c < A.nchunks; c += nwaves) {  // wave-uniform
    uint64_t r0 = A.tile_off[A.chunk_tile0[c]], r1 = A.tile_off[A.chunk_tile0[c + 1]];
    r0 = r0 < M ? r0 : M;  // entries beyond the capacity do not exist (the search is repeated on the exact route)
    r1 = r1 < M ? r1 : M;
    const ChunkDev ch = A.chunks[c];
    const uint8_t* d = A.base + ch.offset;
    // last raw occurrence of the chunk the walk reports, 64 entries a step from the back
    uint64_t last_end = 0;
    if (A.keep_all) {
      if (r1 > r0) last_end = A.m_pos[r1 - 1] + plen;
    } else {
      for (uint64_t hi = r1; hi > r0 && last_end == 0;) {
        const uint64_t lo = hi - r0 >= 64 ? hi - 64 : r0;
        const uint64_t i = lo + lane;
        const unsigned long long kb = __ballot(i < hi && A.keep[i] != 0);
        if (kb) last_end = A.m_pos[lo + (uint64_t)(63 - __clzll((long long)kb))] + plen;
        hi = lo;
      }
    }
    const bool lm = A.line_mode != 0;
    const uint64_t entry = lm ? wave_walk_entry(d, ch.length, last_end, true, lane) : last_end;
    uint32_t n = 0;
    if (need_tail && ch.length) {
      uint64_t* out = A.tail_pos + c * A.tail_cap;
      if (mask_tail) {
        uint32_t nm = 0, nl = 0;
        wave_tail_counts(d, ch.length, s_pat, plen, A.pat.icase != 0, s_zone[wave], lane, !lm, entry, lm, entry, &nm, &nl,
                         lane == 0 ? out : nullptr, A.tail_cap);
        n = lm ? nl : nm;
      } else {
        // long patterns: the zone does not fit one position per lane; lane 0 walks it after a coalesced sweep of
        // the whole wave has pulled it into the caches
        const uint64_t Lr = (ch.length + 15u) & ~(uint64_t)15u;
        for (uint64_t off = (tail_zone_begin(ch.length, plen) & ~(uint64_t)15u) + (uint64_t)lane * kUnit; off < Lr; off += kWaveLoad) {
          const uint4 v = *reinterpret_cast<const uint4*>(d + off);
          asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        }
        if (lane == 0) n = tail_walk(d, ch.length, A.pat.d_pat, plen, entry, lm, out, A.tail_cap, A.pat.icase != 0);
        n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
      }
    }
    if (lane == 0) {
      A.tail_cnt[c] = n;
      A.chunk_shift0[c] = entry;
    }
  }
  // ---- ticket: the last workgroup scans the tail counts (every wave's stores first, then the workgroup's ticket)
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_is_last = atomicAdd(A.ticket, 1u) == gridDim.x - 1u;
  __syncthreads();
  if (!s_is_last) return;
  __threadfence();
  uint64_t* tail_pre = const_cast<uint64_t*>(A.tail_pre);
  uint64_t carry = 0;
  for (uint64_t i0 = 0; i0 < A.nchunks; i0 += kBlock) {
    const uint64_t i = i0 + threadIdx.x;
    const uint64_t x = i < A.nchunks ? __hip_atomic_load(A.tail_cnt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    uint64_t t;
    const uint64_t ex = block_excl_scan_u64(x, sh, &t);
    if (i < A.nchunks) tail_pre[i] = carry + ex;
    carry += t;
  }
  if (threadIdx.x == 0) {
    tail_pre[A.nchunks] = carry;
    const uint64_t kept = A.keep_all ? M : A.keep_pre[M];
    const uint64_t fin = kept + carry;
    A.tot_dev[kTotKept] = kept;
    A.tot_dev[kTotFinal] = fin;
    A.tot_host[kTotKept] = kept;
    A.tot_host[kTotFinal] = fin;
    if (fin > A.f_cap) {
      A.tot_dev[kTotOverflow] |= 2u;
      A.tot_host[kTotOverflow] = A.tot_dev[kTotOverflow];
    }
    *A.ticket = 0u;
  }
}

// ---- one-sync route: k_assemble + k_globalize in one, plus the pinned mirror of the result -----------------------
// final list = per chunk: kept bulk matches, then the tail walk's matches.  Thread i handles raw entry i
// (grid-stride: the grid covers the capacity) and, as chunk i's thread, that chunk's tail matches.
__global__ __launch_bounds__(kBlock) void k_list_out(const ListArgs A) {
  const uint64_t M = list_count(A);
  const uint64_t fin = A.tot_dev[kTotFinal];
  if (fin > A.f_cap) return;  // refused: the host repeats the search on the exact route
  const bool lm = A.line_mode != 0;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t n = M > A.nchunks ? M : A.nchunks;
  for (uint64_t i0 = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); i0 < n; i0 += (uint64_t)gridDim.x * kBlock) {
    const uint64_t i = i0 + lane;
    {
      const bool kept = i < M && (A.keep_all || A.keep[i]);
      const uint32_t c = kept ? A.m_chunk[i] : 0u;
      const uint64_t mpos = kept ? A.m_pos[i] : 0;
      const ChunkDev ch = A.chunks[c];
      // xs::lines: the line's length right here (k_line_lengths' work: [line start, next '\n' behind the match)) --
      // one launch and one chain of memory latencies less; the wave shares long scans, so no lane leaves early
      int64_t e = -1;
      if (A.line_len) e = newline_query<true>(kept, A.base + ch.offset, mpos + A.pat.plen, ch.length, lane);
      if (kept) {
        const uint64_t dst = (A.keep_all ? i : A.keep_pre[i]) + A.tail_pre[c];
        const uint64_t pos = lm ? A.m_ls[i] : mpos;
        if (A.want_f) {
          A.f_pos[dst] = pos;
          A.f_match[dst] = mpos;
          A.f_chunk[dst] = c;
        }
        if (A.out_u64) {
          const uint64_t g = ch.global_offset + pos;
          A.out_u64[dst] = g;
          if (A.out_host) A.out_host[dst] = g;
        }
        if (A.line_len) {
          const uint64_t len = e < 0 ? UINT64_MAX : (uint64_t)e - pos;  // UINT64_MAX: no terminating newline -> dropped
          A.line_len[dst] = len;
          if (A.line_len_host) A.line_len_host[dst] = len;
        }
      }
    }
    // the tail walk's matches of chunk i; their line starts may lie a whole huge line back, so the loop runs
    // wave-uniformly (up to the largest count in the wave) and the wave shares long scans
    const bool has_chunk = i < A.nchunks;
    const uint64_t c = has_chunk ? i : 0;
    const uint32_t nt = has_chunk ? A.tail_cnt[c] : 0u;
    const uint32_t nmax = wave_max_u32(nt);
    if (nmax) {
      uint64_t r1 = has_chunk ? A.tile_off[A.chunk_tile0[c + 1]] : 0;
      r1 = r1 < M ? r1 : M;
      const uint64_t dst0 = has_chunk ? (A.keep_all ? r1 : A.keep_pre[r1]) + A.tail_pre[c] : 0;
      const ChunkDev ch = A.chunks[c];
      const uint8_t* d = A.base + ch.offset;
      for (uint32_t k = 0; k < nmax; ++k) {
        const bool live = k < nt && k < A.tail_cap;
        const uint64_t m = live ? A.tail_pos[c * A.tail_cap + k] : 0;
        const int64_t nl = newline_query<false>(live && lm, d, 0, m, lane);
        int64_t e = -1;
        if (A.line_len) e = newline_query<true>(live, d, m + A.pat.plen, ch.length, lane);
        if (live) {
          const uint64_t pos = !lm ? m : A.pat.nl_first ? m + 1u : nl < 0 ? 0u : (uint64_t)nl + 1u;
          if (A.want_f) {
            A.f_pos[dst0 + k] = pos;
            A.f_match[dst0 + k] = m;
            A.f_chunk[dst0 + k] = (uint32_t)c;
          }
          if (A.out_u64) {
            const uint64_t g = ch.global_offset + pos;
            A.out_u64[dst0 + k] = g;
            if (A.out_host) A.out_host[dst0 + k] = g;
          }
          if (A.line_len) {
            const uint64_t len = e < 0 ? UINT64_MAX : (uint64_t)e - pos;
            A.line_len[dst0 + k] = len;
            if (A.line_len_host) A.line_len_host[dst0 + k] = len;
          }
        }
      }
    }
  }
}

hipError_t launch_chunk_tail(const ListArgs& a, hipStream_t s) {
  uint64_t blocks = (a.nchunks + kWaves - 1) / kWaves;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_chunk_tail, dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_list_out(const ListArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_list_out, grid_capped(std::max<uint64_t>(a.M, a.nchunks)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

// final list = per chunk: kept bulk matches, then the tail walk's matches
__global__ void k_assemble(const ListArgs A) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < A.M && A.keep[i]) {
    const uint32_t c = A.m_chunk[i];
    const uint64_t dst = A.keep_pre[i] + A.tail_pre[c];
    A.f_pos[dst] = A.line_mode ? A.m_ls[i] : A.m_pos[i];
    A.f_match[dst] = A.m_pos[i];
    A.f_chunk[dst] = c;
  }
  // the tail walk's matches, one thread per chunk; their line starts may lie a whole huge line back, so the
  // loop runs wave-uniformly (up to the largest count in the wave) and the wave shares long scans
  const bool has_chunk = i < A.nchunks;
  const uint64_t c = has_chunk ? i : 0;
  const uint32_t n = has_chunk ? A.tail_cnt[c] : 0u;
  const uint32_t nmax = wave_max_u32(n);
  if (nmax) {
    const uint64_t r1 = has_chunk ? A.tile_off[A.chunk_tile0[c + 1]] : 0;
    const uint64_t dst0 = has_chunk ? A.keep_pre[r1] + A.tail_pre[c] : 0;
    const uint8_t* d = A.base + A.chunks[c].offset;
    for (uint32_t k = 0; k < nmax; ++k) {
      const bool live = k < n;
      const uint64_t m = live ? A.tail_pos[c * A.tail_cap + k] : 0;
      const int64_t nl = newline_query<false>(live && A.line_mode != 0, d, 0, m, threadIdx.x & 63u);
      if (live) {
        A.f_pos[dst0 + k] = !A.line_mode ? m : A.pat.nl_first ? m + 1u : nl < 0 ? 0u : (uint64_t)nl + 1u;
        A.f_match[dst0 + k] = m;
        A.f_chunk[dst0 + k] = (uint32_t)c;
      }
    }
  }
}

// xsg_count_async for patterns that can overlap themselves: the greedy walk's count without a trip to the host
__global__ __launch_bounds__(kBlock) void k_bordered_total(const ListArgs A, uint64_t* counters) {
  __shared__ uint64_t sh[kWaves];
  const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t gsz = (uint64_t)gridDim.x * kBlock;
  const uint64_t M = list_count(A);
  uint64_t n = 0;
  for (uint64_t i = gid; i < M; i += gsz) n += A.keep[i];
  for (uint64_t c = gid; c < A.nchunks; c += gsz) n += A.tail_cnt[c];
  const uint64_t t = block_sum_u64(n, sh);
  if (threadIdx.x == 0 && t) atomicAdd((unsigned long long*)&counters[XSG_CTR_MATCHES], (unsigned long long)t);
}
__global__ void k_bordered_seal(const ListArgs A, uint64_t* counters, uint64_t total_bytes, uint32_t* flags, uint64_t* status) {
  const bool overflow = A.M_dev && *A.M_dev > A.M;
  const bool refuse = (A.pat.kind == kClass || A.pat.kind == kDfa) && A.pat.ascii_only && (*flags & 1u);
  *flags = 0u;
  if (status) *status = (overflow ? (uint64_t)XSG_STATUS_OVERFLOW : 0ull) | (refuse ? (uint64_t)XSG_STATUS_NONASCII : 0ull);
  if (overflow || refuse) {
    for (int k = 0; k < XSG_NUM_COUNTERS; ++k) counters[k] = status ? 0ull : UINT64_MAX;
  } else {
    counters[XSG_CTR_BYTES] = total_bytes;
  }
}

hipError_t launch_bordered_total(const ListArgs& a, uint64_t* counters, uint64_t total_bytes, uint32_t* flags, uint64_t* status,
                                 hipStream_t s) {
  uint64_t blocks = (std::max<uint64_t>(a.M, a.nchunks) + (uint64_t)kBlock * 8 - 1) / ((uint64_t)kBlock * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_bordered_total, dim3((unsigned)blocks), dim3(kBlock), 0, s, a, counters);
  hipLaunchKernelGGL(k_bordered_seal, dim3(1), dim3(1), 0, s, a, counters, total_bytes, flags, status);
  return hipGetLastError();
}

hipError_t launch_keep_all(const ListArgs& a, hipStream_t s) {
  if (!a.M) return hipSuccess;
  hipLaunchKernelGGL(k_keep_all, grid_for(a.M), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_greedy_keep(const ListArgs& a, hipStream_t s) {
  if (!a.M) return hipSuccess;
  hipLaunchKernelGGL(k_greedy_keep, grid_for(a.M), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_line_starts_keep(const ListArgs& a, hipStream_t s) {
  if (!a.M) return hipSuccess;
  hipLaunchKernelGGL(k_line_starts_keep, a.M_dev ? grid_capped(a.M) : grid_for(a.M), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_chunk_shift0(const ListArgs& a, hipStream_t s) {
  if (!a.nchunks) return hipSuccess;
  hipLaunchKernelGGL(k_chunk_shift0, grid_for(a.nchunks), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_tail_list(const ListArgs& a, hipStream_t s) {
  if (!a.nchunks) return hipSuccess;
  hipLaunchKernelGGL(k_tail_list, grid_for(a.nchunks), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_assemble(const ListArgs& a, hipStream_t s) {
  const uint64_t n = a.M > a.nchunks ? a.M : a.nchunks;
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_assemble, grid_for(n), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// outputs of the list modes
// ---------------------------------------------------------------------------
__global__ void k_globalize(const LineOutArgs A) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < A.total) A.out_u64[i] = A.chunks[A.f_chunk[i]].global_offset + A.f_pos[i];
}

// xs::line_indices: number of '\n' before the line start
// = newlines in all tiles before the line's tile (exclusive scan of tile_nl) + newlines between the tile start
// and the line start.  Counting from the tile start for every line would cost O(lines x tile) when most lines
// match (66 M lines in 10 GiB: 90 ms); instead every list entry counts only the gap back to the previous entry
// (the previous entry of the same tile, else both ends from their tile starts), and one prefix sum over the
// entries turns the differences into counts: O(shard) whatever the density.
__device__ __forceinline__ uint32_t unit_newlines(const uint8_t* p) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  return (uint32_t)__popc(nl_flags(v.x)) + (uint32_t)__popc(nl_flags(v.y)) + (uint32_t)__popc(nl_flags(v.z)) +
         (uint32_t)__popc(nl_flags(v.w));
}
// newlines among bytes [lo, hi) of the aligned 16-byte unit at p
__device__ __forceinline__ uint32_t unit_newlines_masked(const uint8_t* p, uint32_t lo, uint32_t hi) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const uint32_t f[4] = {nl_flags(v.x), nl_flags(v.y), nl_flags(v.z), nl_flags(v.w)};
  uint32_t n = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t b0 = 4u * q;
    uint32_t keep = 0xffffffffu;
    if (lo > b0) keep &= lo >= b0 + 4 ? 0u : 0xffffffffu << (8u * (lo - b0));
    if (hi < b0 + 4) keep &= hi <= b0 ? 0u : 0xffffffffu >> (8u * (b0 + 4 - hi));
    n += (uint32_t)__popc(f[q] & keep);
  }
  return n;
}
// Newlines in d[from, to) with aligned 16-byte loads only (a gap is ~30 bytes when most lines match: byte loads
// would be 10x the memory operations); the units at both ends are masked.  Chunk buffers are padded to 16.
// The interior runs four independent loads per step: a walk over a whole tile is latency-bound otherwise.
__device__ __forceinline__ uint64_t count_newlines(const uint8_t* d, uint64_t from, uint64_t to) {
  if (from >= to) return 0;
  uint64_t n = 0;
  uint64_t p = from & ~(uint64_t)15;
  if (p < from) {  // leading partial unit
    const uint64_t end = p + kUnit < to ? p + kUnit : to;
    n += unit_newlines_masked(d + p, (uint32_t)(from - p), (uint32_t)(end - p));
    p += kUnit;
  }
  for (; p + 4 * kUnit <= to; p += 4 * kUnit)
    n += unit_newlines(d + p) + unit_newlines(d + p + kUnit) + unit_newlines(d + p + 2 * kUnit) +
         unit_newlines(d + p + 3 * kUnit);
  for (; p + kUnit <= to; p += kUnit) n += unit_newlines(d + p);
  if (p < to) n += unit_newlines_masked(d + p, 0u, (uint32_t)(to - p));  // trailing partial unit
  return n;
}

// shard-wide newline count before entry i's line start, from its tile's prefix (up to one tile of counting)
__device__ __forceinline__ uint64_t newlines_before_entry(const LineOutArgs& A, uint64_t i) {
  const uint32_t c = A.f_chunk[i];
  const uint8_t* d = A.base + A.chunks[c].offset;
  const uint64_t b = A.f_pos[i];
  const uint32_t sh = 31u - (uint32_t)__clz(A.tile_bytes);  // tiles are a power of two
  const uint64_t tl = b >> sh;
  return A.tile_nl_off[A.chunk_tile0[c] + tl] + count_newlines(d, tl << sh, b);
}

// An entry is "near" if the previous entry is in the same chunk and at most a quarter tile back: its gap is
// counted directly -- when most lines match the gaps are tens of bytes and no thread walks a tile while its wave
// waits.  Farther apart, the entry counts from its tile start (at most 4x the bytes of its gap).
__device__ __forceinline__ bool entry_is_near(const LineOutArgs& A, uint64_t i) {
  return i > 0 && A.f_chunk[i - 1] == A.f_chunk[i] && A.f_pos[i] - A.f_pos[i - 1] <= A.tile_bytes / 4u;
}

// pass 1: the far entries get their absolute count (into out_u64, overwritten by k_line_indices later)
__global__ void k_line_nl_abs(const LineOutArgs A) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= A.total) return;
  if (!entry_is_near(A, i)) A.out_u64[i] = newlines_before_entry(A, i);
}

// pass 2: line_len[i] = (newlines before entry i) - (newlines before entry i-1); entry 0: its own count.
__global__ void k_line_nl_delta(const LineOutArgs A) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= A.total) return;
  uint64_t delta;
  if (entry_is_near(A, i)) {
    const uint8_t* d = A.base + A.chunks[A.f_chunk[i]].offset;
    delta = count_newlines(d, A.f_pos[i - 1], A.f_pos[i]);  // line starts ascend inside a chunk
  } else {
    const uint64_t prev = i == 0 ? 0u : entry_is_near(A, i - 1) ? newlines_before_entry(A, i - 1) : A.out_u64[i - 1];
    delta = A.out_u64[i] - prev;
  }
  A.line_len[i] = delta;
}

// line_out_off = exclusive prefix sums of the deltas (total + 1 entries): entry i's count is line_out_off[i + 1]
__global__ void k_line_indices(const LineOutArgs A) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= A.total) return;
  const uint32_t c = A.f_chunk[i];
  const ChunkDev ch = A.chunks[c];
  const uint64_t n = A.line_out_off[i + 1];
  if (ch.line_base == XSG_LINE_BASE_AUTO)
    A.out_u64[i] = A.shard_line_base + n;
  else
    A.out_u64[i] = ch.line_base + (n - A.tile_nl_off[A.chunk_tile0[c]]);
}

// length of the final list: the host's value, or (one-sync route) the device's bounded by the arrays' capacity
__device__ __forceinline__ uint64_t out_count(const LineOutArgs& A) {
  if (!A.tot_dev) return A.total;
  const uint64_t n = A.tot_dev[kTotFinal];
  return n <= A.total ? n : 0;  // over capacity: nothing is produced, the host repeats the search on the exact route
}

// xs::line_indices on the one-sync route: the list is short (it fits the route's capacity), so instead of the
// difference / prefix-sum passes above every entry gets a WAVE: newlines before the line's tile from the prefix of
// the per-tile counts, plus the newlines between the tile start and the line start counted by 64 lanes, 4 KiB a step
// (one thread walking up to a whole tile took 150 us for 7 000 entries).
__global__ __launch_bounds__(kBlock) void k_line_index_waves(const LineOutArgs A) {
  const uint64_t total = out_count(A);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t sh = 31u - (uint32_t)__clz(A.tile_bytes);  // tiles are a power of two
  const uint64_t nwaves = (uint64_t)gridDim.x * kWaves;
  for (uint64_t i = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); i < total; i += nwaves) {  // wave-uniform
    const uint32_t c = A.f_chunk[i];
    const ChunkDev ch = A.chunks[c];
    const uint8_t* d = A.base + ch.offset;
    const uint64_t b = A.f_pos[i];
    const uint64_t tl = b >> sh;
    const uint64_t from = tl << sh;
    uint32_t cnt = 0;
    for (uint64_t p = from + (uint64_t)lane * kUnit; p < b; p += 4u * kWaveLoad) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint64_t q = p + (uint64_t)u * kWaveLoad;
        if (q + kUnit <= b)
          cnt += unit_newlines(d + q);
        else if (q < b)
          cnt += unit_newlines_masked(d + q, 0u, (uint32_t)(b - q));
      }
    }
    cnt = wave_sum_u32(cnt);
    if (lane == 0) {
      const uint64_t n = A.tile_nl_off[A.chunk_tile0[c] + tl] + cnt;
      const uint64_t v = ch.line_base == XSG_LINE_BASE_AUTO ? A.shard_line_base + n
                                                            : ch.line_base + (n - A.tile_nl_off[A.chunk_tile0[c]]);
      A.out_u64[i] = v;
      if (A.out_host) A.out_host[i] = v;
    }
  }
}

// xs::lines: [line start, next '\n' after the match); a line without '\n' is
// dropped (search_wrappers.h:199-202)
__global__ __launch_bounds__(kBlock) void k_line_lengths(const LineOutArgs A) {
  const uint64_t total = out_count(A);
  for (uint64_t i0 = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); i0 < total; i0 += (uint64_t)gridDim.x * kBlock) {
    const uint64_t i = i0 + (threadIdx.x & 63u);
    const bool live = i < total;  // no early exit of a lane: see newline_query
    const ChunkDev ch = A.chunks[live ? A.f_chunk[i] : 0u];
    const uint8_t* d = A.base + ch.offset;
    const int64_t e = newline_query<true>(live, d, live ? A.f_match[i] + A.pat.plen : 0, ch.length, threadIdx.x & 63u);
    if (!live) continue;
    const uint64_t len = e < 0 ? UINT64_MAX : (uint64_t)e - A.f_pos[i];
    const uint64_t g = ch.global_offset + A.f_pos[i];
    A.line_len[i] = len;
    A.out_u64[i] = g;
    if (A.line_len_host) __builtin_nontemporal_store(len, A.line_len_host + i);
    if (A.out_host) __builtin_nontemporal_store(g, A.out_host + i);
    if (e < 0 && A.dropped) atomicAdd(A.dropped, 1u);  // (at most one per chunk: its last line)
  }
}

// xs::lines: the bytes of every reported line, packed.  One THREAD per line: a line of text is a few dozen bytes,
// which a lane moves with one or two 16-byte loads and stores (unaligned global accesses are native on gfx950) --
// a wave per line, the first version, kept 64 lanes busy with 30 bytes (19 ms for the 66 M lines that contain
// `She` in 10 GiB).  Lines over 256 bytes wait until the wave has finished its short ones and are then copied by
// all 64 lanes together, 1 KiB a step.  One-sync route: the packed bytes also go to their pinned mirror, and
// nothing is written if they exceed the capacity (line_out_off[total] is their number).
typedef unsigned int uint4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
__global__ __launch_bounds__(kBlock) void k_line_gather(const LineOutArgs A) {
  const uint64_t total = out_count(A);
  const uint32_t lane = threadIdx.x & 63u;
  if (A.line_bytes_cap && A.line_out_off[total] > A.line_bytes_cap) return;
  uint8_t* const mirror = A.line_bytes_host;
  const uint64_t first = A.slice_end ? A.slice_begin : 0, stop = A.slice_end ? (A.slice_end < total ? A.slice_end : total) : total;
  for (uint64_t i0 = first + (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); i0 < stop; i0 += (uint64_t)gridDim.x * kBlock) {
    const uint64_t i = i0 + lane;
    uint64_t len = i < stop ? A.line_len[i] : UINT64_MAX;
    const bool live = len != UINT64_MAX;  // UINT64_MAX: no terminating newline -> not reported
    const uint8_t* src = live ? A.base + A.chunks[A.f_chunk[i]].offset + A.f_pos[i] : nullptr;
    const uint64_t doff = live ? A.line_out_off[i] : 0;
    uint8_t* dst = live ? A.line_bytes + doff : nullptr;
    if (!live) len = 0;
    const bool big = len > 256;
    if (!big) {
      uint64_t k = 0;
      for (; k + 16 <= len; k += 16) {
        const uint4_unaligned v = *reinterpret_cast<const uint4_unaligned*>(src + k);
        *reinterpret_cast<uint4_unaligned*>(dst + k) = v;
        if (mirror) *reinterpret_cast<uint4_unaligned*>(mirror + doff + k) = v;
      }
      for (; k < len; ++k) {
        dst[k] = src[k];
        if (mirror) mirror[doff + k] = src[k];
      }
    }
    unsigned long long pend = __ballot(big);
    while (pend) {  // wave-uniform
      const int L = __builtin_ctzll(pend);
      pend &= pend - 1ull;
      const uint8_t* s2 = reinterpret_cast<const uint8_t*>((uintptr_t)__shfl((long long)(uintptr_t)src, L));
      const uint64_t o2 = (uint64_t)__shfl((long long)doff, L);
      uint8_t* d2 = A.line_bytes + o2;
      const uint64_t n2 = (uint64_t)__shfl((long long)len, L);
      const uint64_t whole = n2 & ~(uint64_t)15;
      for (uint64_t k = (uint64_t)lane * 16u; k < whole; k += 64u * 16u) {
        const uint4_unaligned v = *reinterpret_cast<const uint4_unaligned*>(s2 + k);
        *reinterpret_cast<uint4_unaligned*>(d2 + k) = v;
        if (mirror) *reinterpret_cast<uint4_unaligned*>(mirror + o2 + k) = v;
      }
      if (whole + lane < n2) {
        d2[whole + lane] = s2[whole + lane];
        if (mirror) mirror[o2 + whole + lane] = s2[whole + lane];
      }
    }
  }
}

hipError_t launch_globalize(const LineOutArgs& a, hipStream_t s) {
  if (!a.total) return hipSuccess;
  hipLaunchKernelGGL(k_globalize, grid_for(a.total), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_line_nl_delta(const LineOutArgs& a, hipStream_t s) {
  if (!a.total) return hipSuccess;
  hipLaunchKernelGGL(k_line_nl_abs, grid_for(a.total), dim3(kBlock), 0, s, a);
  hipLaunchKernelGGL(k_line_nl_delta, grid_for(a.total), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_line_indices(const LineOutArgs& a, hipStream_t s) {
  if (!a.total) return hipSuccess;
  hipLaunchKernelGGL(k_line_indices, grid_for(a.total), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_line_lengths(const LineOutArgs& a, hipStream_t s) {
  if (!a.total) return hipSuccess;
  hipLaunchKernelGGL(k_line_lengths, a.tot_dev ? grid_capped(a.total) : grid_for(a.total), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_line_index_waves(const LineOutArgs& a, hipStream_t s) {
  if (!a.total) return hipSuccess;
  uint64_t blocks = (a.total + kWaves - 1) / kWaves;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_line_index_waves, dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}
// The same gather, OUTPUT-centric, for results of millions of lines (the exact route): a thread per line writes 16 bytes at
// whatever alignment its line starts at, then up to 15 single bytes -- 66 M lines of ~30 bytes (3 GB of `She` lines on
// 10 GiB) left at 35 GB/s, 58 ms of a 93 ms search (profiles/r04_dense_timeline.txt).  Here a workgroup takes kBlock
// consecutive lines = one contiguous span of the packed output, parks their output offsets and source addresses in LDS,
// and every thread fills ALIGNED 16-byte units of that span: which line a unit starts in by a binary search in LDS, its
// bytes picked one by one across the line boundaries inside it (source lines are contiguous text: the byte loads hit
// lines the neighbouring lanes just touched), one 16-byte store.  The span's first and last unit are shared with the
// neighbouring workgroups and written byte by byte.
__global__ __launch_bounds__(kBlock) void k_line_gather_span(const LineOutArgs A) {
  __shared__ uint64_t s_off[kBlock + 1];
  __shared__ const uint8_t* s_src[kBlock];
  const uint64_t total = out_count(A);
  if (A.line_bytes_cap && A.line_out_off[total] > A.line_bytes_cap) return;
  uint8_t* const mirror = A.line_bytes_host;
  const uint64_t first = A.slice_end ? A.slice_begin : 0, stop = A.slice_end ? (A.slice_end < total ? A.slice_end : total) : total;
  for (uint64_t w0 = first + (uint64_t)blockIdx.x * kBlock; w0 < stop; w0 += (uint64_t)gridDim.x * kBlock) {
    const uint32_t n = (uint32_t)(stop - w0 < (uint64_t)kBlock ? stop - w0 : (uint64_t)kBlock);
    __syncthreads();  // (the previous round's readers are done with the arrays)
    if (threadIdx.x < n) {
      const uint64_t i = w0 + threadIdx.x;
      s_off[threadIdx.x] = A.line_out_off[i];
      s_src[threadIdx.x] = A.base + A.chunks[A.f_chunk[i]].offset + A.f_pos[i];
    }
    if (threadIdx.x == 0) s_off[n] = A.line_out_off[w0 + n];
    __syncthreads();
    const uint64_t B = s_off[0], E = s_off[n];  // a dropped line (no terminating newline) has length 0 here: scan_val
    for (uint64_t U = (B & ~(uint64_t)15) + 16u * threadIdx.x; U < E; U += 16u * kBlock) {
      const uint64_t lo = U > B ? U : B, hi = U + 16u < E ? U + 16u : E;
      // the line that holds output byte lo: the last l with s_off[l] <= lo (lo < E = s_off[n], so l < n and s_off[l + 1] > lo)
      uint32_t a = 0, b = n;  // invariant: s_off[a] <= lo < s_off[b]
      while (b - a > 1) {
        const uint32_t m = (a + b) >> 1;
        if (s_off[m] <= lo) a = m; else b = m;
      }
      uint32_t l = a;
      uint64_t beg = s_off[l], end = s_off[l + 1];
      const uint8_t* src = s_src[l];
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      for (uint64_t o = lo; o < hi; ++o) {
        while (o >= end) {  // (zero-length lines in between are stepped over)
          ++l;
          beg = end;
          end = s_off[l + 1];
          src = s_src[l];
        }
        const uint32_t k = (uint32_t)(o - U);
        w[k >> 2] |= (uint32_t)src[o - beg] << (8u * (k & 3u));
      }
      if (hi - lo == 16u) {
        const uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
        if (A.line_bytes) *reinterpret_cast<uint4*>(A.line_bytes + U) = v;
        if (mirror) {  // (streaming store: the unit is never read back on the device)
          typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
          u32x4_t nv = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_t*>(mirror + U));
        }
      } else {
        for (uint64_t o = lo; o < hi && A.line_bytes; ++o) {
          const uint32_t k = (uint32_t)(o - U);
          A.line_bytes[o] = (uint8_t)(w[k >> 2] >> (8u * (k & 3u)));
        }
        if (mirror) {
          // a unit shared with the neighbouring workgroup: single bytes over the link are single transactions (two such
          // units x ~8 bytes x 258 000 workgroups held the gather of 2 GB at 36 GB/s where the link moves 57).  The
          // bytes meet in a device array, one entry per BOUNDARY between workgroups (head -> this one's, tail -> the next
          // one's); k_line_gather_edges writes every such unit once, as 16 bytes.
          const uint64_t bnd = (w0 - first) / kBlock + (lo > U ? 0u : 1u);
          if (A.edge_units && !A.slice_end) {  // (k_line_gather_edges covers whole results only)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (w[q]) atomicOr(A.edge_units + 4 * bnd + q, w[q]);
          } else {
            for (uint64_t o = lo; o < hi; ++o) {
              const uint32_t k = (uint32_t)(o - U);
              mirror[o] = (uint8_t)(w[k >> 2] >> (8u * (k & 3u)));
            }
          }
        }
      }
    }
  }
}

// the units k_line_gather_span's workgroups share: boundary b lies at output byte B_b = line_out_off[first + b * kBlock];
// consecutive boundaries inside one 16-byte unit (workgroups whose whole span is a few bytes) are merged by the first
__global__ __launch_bounds__(kBlock) void k_line_gather_edges(const LineOutArgs A, const uint64_t nbnd) {
  const uint64_t total = out_count(A);
  const uint64_t first = 0, stop = total;
  const uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b > nbnd) return;
  auto at = [&](uint64_t k) { const uint64_t i = first + k * kBlock; return A.line_out_off[i < stop ? i : stop]; };
  const uint64_t Bb = at(b);
  if ((Bb & 15u) == 0) return;  // the boundary falls between two units: nobody shares one
  const uint64_t U = Bb & ~(uint64_t)15;
  if (b > 0) {
    const uint64_t Bp = at(b - 1);
    if ((Bp & 15u) != 0 && (Bp & ~(uint64_t)15) == U) return;  // an earlier boundary of the same unit writes it
  }
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = A.edge_units[4 * b + q];
  for (uint64_t k = b + 1; k <= nbnd; ++k) {
    const uint64_t Bk = at(k);
    if ((Bk & 15u) == 0 || (Bk & ~(uint64_t)15) != U) break;
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] |= A.edge_units[4 * k + q];
  }
  *reinterpret_cast<uint4*>(A.line_bytes_host + U) = make_uint4(w[0], w[1], w[2], w[3]);  // (the mirror holds 16 bytes beyond the result)
}

hipError_t launch_line_gather(const LineOutArgs& a, hipStream_t s) {
  if (!a.total) return hipSuccess;
  const uint64_t n = a.slice_end ? a.slice_end - a.slice_begin : a.total;
  if (!n) return hipSuccess;
  if (!a.tot_dev && (n >= (1u << 16) || !a.line_bytes)) {  // the exact route with a result worth the set-up (or only a pinned destination)
    const uint64_t nwg = (n + kBlock - 1) / kBlock;
    const uint64_t blocks = std::min<uint64_t>(nwg, 1u << 20);
    hipLaunchKernelGGL(k_line_gather_span, dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
    if (a.edge_units && a.line_bytes_host && !a.slice_end)  // (edge_units: 16 zeroed bytes per boundary, nwg + 1 of them)
      hipLaunchKernelGGL(k_line_gather_edges, dim3((unsigned)((nwg + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a, nwg);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_line_gather, a.tot_dev ? grid_capped(n) : grid_for(n), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}


// Loading a code object costs milliseconds the first time one of its kernels is launched: xsg_ctx_create launches this
// empty kernel of every kernel file, so that the first search of a process does not pay for it (5.5 ms of the first
// xsg_count, scripts/first_call.py).
__global__ void k_warm_list() {}
hipError_t warm_list_kernels(hipStream_t s) {
  hipLaunchKernelGGL(k_warm_list, dim3(1), dim3(1), 0, s);
  return hipGetLastError();
}

}  // namespace xsg
