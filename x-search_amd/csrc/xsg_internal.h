// xsg_internal.h -- shared between the HIP kernels (xsg_kernels.hip) and the
// C-ABI host code (xsg_api.cpp).  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/xsg.h"

namespace xsg {

// ---- tile geometry ---------------------------------------------------------
// A workgroup of 4 wave64s scans one tile.  Each lane reads kLoads 16-byte
// units; a wave-instruction therefore covers 1 KiB of consecutive bytes
// (fully coalesced global_load_dwordx4), a wave covers a contiguous span of
// kLoads KiB and the tile is the 4 spans back to back.
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr uint32_t kUnit = 16;
constexpr uint32_t kWaveLoad = 64 * kUnit;
// kLoads (units per lane) is a template parameter of k_scan: the tile is
// kLoads KiB per wave x 4 waves = 16 KiB (kLoads 4) or 32 KiB (kLoads 8).
constexpr uint32_t kDefaultTileBytes = 16384;
constexpr uint32_t kDefaultStagger = 16;  // see k_scan / pick_stagger (round 2: xsg_shard_tune picked 16 in every 50 GiB run)
constexpr uint32_t kTuneAuto = 0xffffffffu;  // ScanArgs::tune: let launch_scan pick the stagger per variant

// Same layout as xsg_chunk (include/xsg.h).
struct ChunkDev {
  uint64_t offset;
  uint64_t length;
  uint64_t global_offset;
  uint64_t line_base;
};

// Candidate filter shapes (how many of the first 8 pattern bytes the
// in-register window compare covers; longer patterns are verified from memory).
enum FilterKind : int {
  kMask1 = 0,  // plen 1..3 : one masked dword compare
  kOne = 1,    // plen 4    : one dword compare
  kMask2 = 2,  // plen 5..7 : one dword + one masked dword
  kTwo = 3,    // plen 8    : two dword compares (exact)
  kLong = 4,   // plen > 8  : two dword compares on the 8-byte filter window (koff), then memory for the rest
  kClass = 5,  // class sequence (xsg_classseq.h): two masked dword compares over the literal bytes of the window,
               // then every position against its 256-bit set (d_pat holds the sets, 32 bytes per position)
  kClassFast = 7,  // instantiation only (PatternDev::kind stays kClass): a class sequence whose window takes the exact
                   // 16 + 32 bit compare (PatternDev::cls_fast), window filter
  kDfa = 6     // variable-length expression (xsg_regex.h): k_rx_scan walks every line with a byte-class DFA; d_pat holds
               // class_of[256], the forward table, the reverse table (uint16 row offsets)
};

struct PatternDev {
  uint32_t plen;
  uint32_t kind;
  uint32_t p0, m0, p1, m1;  // the 8 pattern bytes of the filter window (pattern[koff..koff+8)) as dwords + byte masks
  uint32_t q0, q1;          // ignore_case hot filter: (p0 | 0x20202020) & m0, (p1 | 0x20202020) & m1 (k_scan, LAZY)
  uint32_t koff;            // kLong, kClass: offset of the 8-byte filter window inside the pattern (0 for the other kinds)
  const uint8_t* d_pat;     // device copy of the pattern
  uint32_t exact_tail;      // XSG_FLAG_EXACT_TAIL
  uint32_t nl_first;        // literal whose FIRST byte is '\n': the reference's line start of a match then lies one byte behind the
                            // match's first byte (previous_new_line_offset_relative_to_match looks at that byte first, :111-123)
  uint32_t has_newline;     // pattern contains '\n'
  uint32_t icase;           // XSG_FLAG_IGNORE_CASE: data bytes are ASCII-lowered before every compare (pattern is lowered on the host)
  uint32_t lazy_exact;      // ignore_case: every byte of the filter window is a letter, so the hot filter on (data | 0x20) IS the
                            // exact folded compare and its results stand (no second, properly folded pass over the window)
  uint32_t hot;             // window kinds (kTwo, kLong, kClass): 1 = aligned-dword trigger, 0 = window filter (k_scan<..., ALIGNED>)
  uint32_t cls_fast;        // kClass: m1 is all ones and the low half of m0 too: the window filter compares 16 + 32 exact bits
  uint32_t cls_inreg;       // kClass, plen <= 8 (koff = 0): candidates are decided in registers (k_scan: cls_verify_at) ...
  uint32_t cls_chk;         // ... looking up only these positions (bit k) in the sets: the ones the hot filter's compare does
                            // not already decide exactly
  uint32_t cls_exact;       // kClass, one alternative, plen <= 8, and the masked window compare decides every position exactly:
                            // a candidate IS a match, nothing is looked up
  uint32_t nalt;            // kClass: alternatives (d_pat holds nalt x plen sets, alternative-major); 0/1 otherwise
  uint32_t ascii_only;      // kClass: the expression is exact on ASCII data only ('.', negated classes): k_scan raises
                            // ScanArgs::flags bit 0 when it meets a byte >= 0x80
  // kDfa (ascii_only as for kClass; plen = the shortest match; exact_tail = 1)
  uint32_t rx_ncls;                // byte classes
  uint32_t rx_fwd_n, rx_rev_n;     // table entries (states x classes) of the forward / reverse automaton
  uint32_t rx_fwd_start, rx_fwd_acc;  // ROW OFFSETS (state x ncls): start state, first accepting state
  uint32_t rx_rev_start, rx_rev_acc;
  uint32_t rx_anc_n, rx_anc_start, rx_anc_acc;  // the ANCHORED forward automaton (behind the reverse table in d_pat): k_rx_verify
  uint32_t rx_multiline;           // a set of the expression accepts '\n': the chunk, not the line, is the unit (k_rx_chunk)
  uint32_t rx_ntrig, rx_trig4;     // rx_skip and at most four trigger byte values besides '\n' (packed in rx_trig4): k_rx_scan tests
                                   // the tile for them on the 16-byte loads and leaves a tile that holds none without staging it
  uint32_t rx_skip;                // bit 7 of every class_of[] entry flags a TRIGGER byte: one that moves the forward automaton
                                   // out of its start state, or '\n' (needs ncls <= 128; XSG_RX_SKIP=0 switches it off)
};

// Per-tile line summaries (XSG_COUNT_LINES): see xsg_linesum.h.

struct ScanArgs {
  const uint8_t* base;          // shard buffer
  const ChunkDev* chunks;       // device chunk table
  const uint32_t* tile_chunk;   // tile -> chunk (null when the shard has one chunk)
  const uint64_t* chunk_tile0;  // first tile of every chunk (nchunks + 1 entries)
  uint64_t ntiles;
  uint64_t nchunks;             // entries of `chunks` (k_rx_chunk: one lane per chunk)
  uint32_t tile_bytes;          // 16384 (selects the k_scan instantiation)
  uint32_t tune;                // bits 0-7: wave stagger in units of s_sleep(1) = 64 clocks; kTuneAuto = per variant (XSG_TUNE overrides)
  uint32_t epoch;               // 1..0xffff: tag of this pass in the upper half of the tile_last words
  PatternDev pat;
  // outputs of the counting pass.  Only waves that found something write (no store on the common path), so
  // "nothing found" must already be in place: tile_cnt == 0 and tile_sum == 0 (k_count_finish restores both as
  // it consumes them), tile_last from an older epoch (never reset: a newer tag wins the atomicMax).
  uint32_t* tile_cnt;                  // matches starting in the tile (o < limit)
  uint32_t* tile_nl;                   // '\n' in the tile              (WANT_NL; every tile is written)
  uint32_t* tile_sum;                  // line summary PER WAVE (4 per tile), stored XOR kSumNl (WANT_LINES)
  uint32_t* tile_last;                 // (epoch << 16) | max (match offset + plen) in the tile, relative to the tile start
  uint32_t dense_hint;                 // an earlier count of this pattern over this shard found it dense: 1 = more than one result
                                       // per 8 KiB (dense_bytes_route), 2 = more than one per 2 KiB (pick_stagger too)
  uint32_t lines_only;                 // count pass with line summaries, matches not asked for (xs::count_lines): a kMask1
                                       // needle skips its match counting (and, where no end-of-chunk walk exists, tile_cnt /
                                       // tile_last altogether)
  uint32_t* flags;                     // one word per shard, zero at rest: bit 0 = "non-ASCII byte under an ascii_only expression"
  const uint32_t* tile_mask;           // k_rx_scan: if set, only tiles with a non-zero word can hold the start of a line with a
                                       // match (the factor prefilter, xsg_api.cpp: ensure_factor_mask); null: every tile
  // inputs/outputs of the emit pass
  uint64_t m_cap;            // entries m_pos / m_chunk can hold (ranks beyond are dropped; 0 = as many as there are)
  const uint64_t* tile_off;  // exclusive prefix of tile_cnt
  uint64_t* m_pos;           // chunk-local offset of every match, ascending
  uint32_t* m_chunk;         // its chunk
  // emit pass over the tiles that hold a match only (the one-sync list route): the ordered list of those tiles and
  // where its length lives; null: one workgroup per tile of the shard, each leaving at once if its count is 0
  const uint32_t* hit_tiles;
  const uint64_t* n_hits_dev;
  uint64_t hit_cap;
  // one byte per tile (four tiles a word): bit w = wave w of the tile found something.  Set by a count pass that is
  // followed by an emit pass over the hit list (null otherwise), read and cleared by that emit pass; a stale bit is
  // harmless (the wave reads its 4 KiB and finds nothing), so the array is only ever zeroed when it is (re)allocated
  uint32_t* tile_wmask;
};

constexpr int kFinishBlocks = 2048;  // upper bound of k_count_finish's grid (size of FinishArgs::partials / 3)

struct FinishArgs {
  const uint8_t* base;
  const ChunkDev* chunks;
  const uint64_t* chunk_tile0;
  uint64_t nchunks;
  uint64_t ntiles;
  PatternDev pat;
  uint32_t* tile_cnt;         // read, then zeroed again
  const uint32_t* tile_nl;
  uint32_t* tile_sum;         // read, then zeroed again
  const uint32_t* tile_last;  // valid where the upper half equals `epoch`
  uint32_t epoch;
  uint32_t tile_bytes;
  uint64_t* counters;       // device, XSG_NUM_COUNTERS: overwritten (no zeroing needed)
  uint64_t* host_counters;  // optional: pinned host mirror of the same four values
  uint64_t* status;         // optional (xsg_count_async_status): receives XSG_STATUS_* bits; a refusal then ZEROES the counters
                            // instead of poisoning them
  uint64_t* partials;       // scratch: 3 x kFinishBlocks
  uint32_t* ticket;         // zero at rest: the last workgroup to arrive does the final sum and resets it
  uint32_t* flags;          // ScanArgs::flags: read and cleared by that workgroup; bit 0 poisons the counters (UINT64_MAX)
  uint64_t total_bytes;     // sum of the chunk lengths (host-side knowledge)
  uint32_t want_nl;
  uint32_t want_lines;
  uint32_t want_matches;
  uint32_t cnt_is_lines;  // kDfa, XSG_COUNT_LINES: tile_cnt holds matching lines, its sum is reported as XSG_CTR_LINES
};

// ---- launchers (xsg_kernels.hip) --------------------------------------------
hipError_t launch_scan_count(const ScanArgs& a, bool want_nl, bool want_lines, hipStream_t s);
hipError_t launch_scan_emit(const ScanArgs& a, hipStream_t s);
hipError_t launch_count_finish(const FinishArgs& a, hipStream_t s);
// kDfa (xsg_rx_kernels.hip): same ScanArgs, same per-tile outputs.  count: tile_cnt = matches (or, want_lines, matching
// lines) of the lines that START in the tile, tile_nl if want_nl; emit: the matches at their ranks.
hipError_t launch_rx_count(const ScanArgs& a, bool want_nl, bool want_lines, hipStream_t s);
hipError_t launch_rx_emit(const ScanArgs& a, hipStream_t s);
// the prefilter route of kDfa (xsg_rx_kernels.hip): candidates of the scan kernel -> verified, walked, packed
struct RxPreArgs {
  const uint8_t* base;
  const ChunkDev* chunks;
  const uint64_t* chunk_tile0;
  uint64_t nchunks;
  PatternDev pat;            // the kDfa pattern (tables in d_pat)
  uint64_t n;                // candidates, ascending per chunk
  const uint64_t* tile_off;  // their ranks by tile: a chunk's candidates are [tile_off[tile0[c]], tile_off[tile0[c+1]])
  const uint64_t* c_pos;
  const uint32_t* c_chunk;
  uint32_t* c_len;           // length of the match that starts at the candidate, 0: none
  uint32_t* c_keep;          // reported by the reference's walk
  uint64_t* scan_tmp;        // n / 2048 + 1 words
  uint32_t* flags;           // ScanArgs::flags; bit 1: a candidate outran k_rx_verify's budget (the host takes the other route)
  const uint64_t* c_pre;     // exclusive prefix of c_keep
  uint64_t* m_pos;           // the reported ones, packed
  uint32_t* m_chunk;
};
hipError_t launch_rx_verify_keep(const RxPreArgs& a, hipStream_t s);
hipError_t launch_rx_compact(const RxPreArgs& a, hipStream_t s);
// "xsg::k_scan<KIND, WANT_NL, WANT_LINES, EMIT, LOADS, ICASE>" + the stagger launch_scan would use, for reports
void describe_scan(const ScanArgs& a, bool want_nl, bool want_lines, bool emit, char* out, size_t cap);

// exclusive scan: out[i] = sum_{k<i} in[k] for i in [0, n]; out has n+1 entries.
// tmp must hold scan_tmp_elems(n) uint64 values.
uint64_t scan_tmp_elems(uint64_t n);
hipError_t launch_exclusive_scan_u32(const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, hipStream_t s);
hipError_t launch_exclusive_scan_u64(const uint64_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, hipStream_t s);

// ---- the one-sync list route (xsg_api.cpp: run_list_fast) ---------------------------------------------------
// Every size the host used to fetch between the stages of a list search (raw occurrences, kept ones, tail matches,
// line bytes) stays on the device: arrays have CAPACITIES, kernels read the counts they need from this block of
// device words and bound themselves by the capacities, and the kernels that produce a count also store it in a
// pinned host mirror.  The host syncs once, at the end, and reads the mirror; kTotOverflow != 0 means a capacity
// was too small and the search is repeated on the exact route (which sizes every array from a fetched count).
enum FastTot : int {
  kTotRaw = 0,        // raw bulk occurrences = sum of tile_cnt
  kTotHits = 1,       // tiles that hold at least one
  kTotKept = 2,       // raw occurrences the reference walk reports
  kTotFinal = 3,      // kept + end-of-chunk matches = length of the list
  kTotNewlines = 4,   // '\n' in the shard (xs::line_indices)
  kTotLineBytes = 5,  // xs::lines: packed bytes
  kTotOverflow = 6,   // != 0: some capacity was exceeded (bit 0 raw, 1 list, 2 line bytes)
  kTotWords = 8
};

// exclusive prefix sums in two launches with the count on the device: k_scan2_a (block sums; the last workgroup to
// arrive scans them and publishes the total), k_scan2_b (writes out[0..n)); out[n] = total.
struct Scan2Args {
  const void* in;          // uint32 or uint64 (UINT64_MAX scans as 0: dropped lines)
  uint64_t* out;           // n + 1 entries
  uint64_t n_cap;          // the grids cover this many entries
  const uint64_t* n_dev;   // optional: n = min(*n_dev, n_cap); null: n = n_cap
  uint64_t* blk;           // scratch: 2 x (blocks + 1) words
  uint32_t* ticket;        // zero at rest
  uint64_t* tot_dev;       // optional: the grand total (device word) ...
  uint64_t* tot_host;      // ... and its pinned mirror
  // HITS: the ordered list of indices with a non-zero entry
  uint32_t* hit_idx;
  uint64_t hit_cap;
  uint64_t* hits_dev;
  uint64_t* hits_host;
  uint64_t* ovf_dev;       // optional: |= ovf_bit when the total exceeds total_cap (or the hits exceed hit_cap)
  uint64_t* ovf_host;
  uint64_t total_cap;
  uint64_t ovf_bit;
  uint32_t ovf_init;       // the first producer of a search: the overflow word is STORED (bit or 0), not OR-ed into
};
uint64_t scan2_tmp_elems(uint64_t n_cap);
hipError_t launch_scan2_u32(const Scan2Args& a, bool hits, hipStream_t s);
hipError_t launch_scan2_u64(const Scan2Args& a, hipStream_t s);

struct ListArgs {
  const uint8_t* base;
  const ChunkDev* chunks;
  const uint64_t* chunk_tile0;
  uint64_t nchunks;
  PatternDev pat;
  uint64_t M;                // raw matches (bulk, o < limit), ascending per chunk
  const uint64_t* M_dev;     // if set: the number lives on the device (tile_off[ntiles]); M is then the CAPACITY of the
                             // arrays and the kernels work on min(*M_dev, M) entries (xsg_count_async, bordered patterns)
  const uint64_t* m_pos;     // chunk-local offsets
  const uint32_t* m_chunk;
  const uint64_t* tile_off;  // -> first raw index of each chunk = tile_off[chunk_tile0[c]]
  uint64_t* m_ls;            // line start per raw match (line modes)
  uint32_t* keep;            // 1 = survives the walk (greedy / first in line)
  const uint64_t* keep_pre;  // exclusive prefix of keep (M + 1 entries)
  uint64_t* chunk_shift0;    // per chunk: where the reference walk enters the tail zone
  uint32_t* tail_cnt;        // per chunk
  uint64_t* tail_pos;        // per chunk x tail_cap: chunk-local match offsets from the tail walk
  uint32_t tail_cap;
  const uint64_t* tail_pre;  // exclusive prefix of tail_cnt (nchunks + 1)
  uint32_t line_mode;        // 0: matches, 1: lines (skip_to_nl)
  // final list
  uint64_t* f_pos;    // chunk-local: match offset (match mode) or line start (line modes)
  uint64_t* f_match;  // chunk-local offset of the (first) match of that line
  uint32_t* f_chunk;
  uint64_t total;
  // one-sync route
  uint32_t keep_all;         // every raw occurrence is reported (no keep[] / keep_pre[]: entry i stays entry i)
  uint64_t* tot_dev;         // FastTot words
  uint64_t* tot_host;        // pinned mirror
  uint32_t* ticket;          // zero at rest
  uint64_t f_cap;            // entries f_pos / f_match / f_chunk / out arrays can hold
  uint64_t* out_u64;         // k_list_out: global offset of every final entry (match / line-start tags) ...
  uint64_t* out_host;        // ... and its pinned mirror (may be null)
  uint32_t want_f;           // k_list_out: also write f_pos / f_match / f_chunk (line indices, lines)
  uint32_t* long_flag;       // k_greedy_keep: if set, a chain over its budget is left unfinished and this word raised
  uint64_t* line_len;        // k_list_out, xs::lines: length of every final entry's line (UINT64_MAX: unterminated -> dropped) ...
  uint64_t* line_len_host;   // ... and its pinned mirror
};

// mask[tile of the line start] = 1 for every kept entry of a candidate list (ListArgs after launch_line_starts_keep)
hipError_t launch_rx_mark_tiles(const ListArgs& a, uint32_t* mask, hipStream_t s);
hipError_t launch_greedy_keep(const ListArgs& a, hipStream_t s);
// long chains: J[i] = nxt(i) (uint32 index or 0xffffffff), then rounds of "mark J(marked), square J" until *changed stays 0
hipError_t launch_greedy_links(const ListArgs& a, uint32_t* J, hipStream_t s);
hipError_t launch_greedy_jump(const ListArgs& a, const uint32_t* J, uint32_t* J2, uint32_t* changed, hipStream_t s);
// line tags of a literal that contains '\n': line start of every raw occurrence, the chunk heads, and the links of the
// reference's walk (next = the first occurrence behind the first '\n' at or behind this one's end)
hipError_t launch_nlpat_links(const ListArgs& a, uint32_t* J, hipStream_t s);
hipError_t launch_line_starts_keep(const ListArgs& a, hipStream_t s);
hipError_t launch_keep_all(const ListArgs& a, hipStream_t s);
hipError_t launch_chunk_shift0(const ListArgs& a, hipStream_t s);
hipError_t launch_tail_list(const ListArgs& a, hipStream_t s);
hipError_t launch_assemble(const ListArgs& a, hipStream_t s);
// counters[XSG_CTR_MATCHES] = sum of keep[0..M) + sum of tail_cnt, counters[XSG_CTR_BYTES] = total_bytes, the other two 0;
// all four UINT64_MAX if *M_dev exceeds the capacity (the caller falls back to the synchronous route) or the scan
// raised flags bit 0 (ascii_only expression on non-ASCII data).  counters must be zero on entry.
// status (optional): as FinishArgs::status
hipError_t launch_bordered_total(const ListArgs& a, uint64_t* counters, uint64_t total_bytes, uint32_t* flags, uint64_t* status,
                                 hipStream_t s);
// one-sync route: chunk_shift0 + tail walk by one wave per chunk, the prefix of the tail counts and kTotFinal by the
// last workgroup to arrive; then the final list (k_assemble + k_globalize in one)
hipError_t launch_chunk_tail(const ListArgs& a, hipStream_t s);
hipError_t launch_list_out(const ListArgs& a, hipStream_t s);

struct LineOutArgs {
  const uint8_t* base;
  const ChunkDev* chunks;
  const uint64_t* chunk_tile0;
  uint64_t nchunks;
  PatternDev pat;
  uint64_t total;
  const uint64_t* f_pos;
  const uint64_t* f_match;
  const uint32_t* f_chunk;
  uint64_t* out_u64;  // global offsets or line indices
  // line indices
  const uint64_t* tile_nl_off;  // exclusive prefix of tile_nl over all tiles of the shard
  uint32_t tile_bytes;
  uint64_t shard_line_base;
  // lines
  uint64_t* line_len;  // length without '\n'; UINT64_MAX marks "no terminating newline" (dropped)
  const uint64_t* line_out_off;
  uint8_t* line_bytes;
  // one-sync route: `total` is the capacity, the length of the list is tot_dev[kTotFinal]
  const uint64_t* tot_dev;
  uint64_t* out_host;        // pinned mirror of out_u64 (may be null)
  uint64_t* line_len_host;   // pinned mirror of line_len (xs::lines)
  uint8_t* line_bytes_host;  // pinned mirror of line_bytes
  uint64_t line_bytes_cap;   // bytes line_bytes (and its mirror) can hold
  uint32_t* dropped;         // k_line_lengths: incremented per line without a terminating newline (may be null)
  uint64_t slice_begin, slice_end;  // k_line_gather: only the entries [slice_begin, slice_end) (slice_end == 0: all of them)
  uint32_t* edge_units;      // k_line_gather_span into a pinned mirror: 4 zeroed words per workgroup boundary (total / kBlock + 2), or null
};
hipError_t launch_globalize(const LineOutArgs& a, hipStream_t s);
hipError_t launch_line_nl_delta(const LineOutArgs& a, hipStream_t s);
hipError_t launch_line_indices(const LineOutArgs& a, hipStream_t s);
hipError_t launch_line_lengths(const LineOutArgs& a, hipStream_t s);
hipError_t launch_line_gather(const LineOutArgs& a, hipStream_t s);
// one empty launch per kernel file (code object): see xsg_kernels.hip
hipError_t warm_scan_kernels(hipStream_t s);
hipError_t warm_list_kernels(hipStream_t s);
hipError_t warm_rx_kernels(hipStream_t s);
// one-sync route: the line index of every entry by one wave per entry (sparse lists: no prefix pass over the entries)
hipError_t launch_line_index_waves(const LineOutArgs& a, hipStream_t s);

}  // namespace xsg
