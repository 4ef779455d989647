// xsg_objects.h -- private object definitions shared by xsg_api.cpp (device-
// resident searches) and xsg_file.cpp (the file pipeline).  Not installed.
#pragma once
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "xsg_internal.h"

namespace xsg {
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const char* last_error_message();
// XSG_TRACE=1: timestamped marks on stderr ("[xsg +12.345 ms] label"), milliseconds since the library was loaded --
// where the time of a process's first search goes (HIP start-up, code objects, pinned ring, device buffers, first chunk).
bool trace_on();
void trace(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
}  // namespace xsg
// An XSG_* tuning / A-B toggle: read from the environment ONCE per process (getenv next to a host thread that writes the
// environment is a data race, and the one-sync list route exists to save microseconds).  XSG_TEST_HOOKS=1, set before the
// library is loaded (tests/conftest.py, scripts/fuzz_campaign.py), re-reads it at every use: the route tests switch
// toggles between searches.
namespace xsg {
bool test_hooks();
}
#define XSG_TOGGLE(name) \
  ([]() -> const char* { static const char* const v = getenv(name); return xsg::test_hooks() ? getenv(name) : v; }())

#define XSG_TRACE(...)                      \
  do {                                      \
    if (xsg::trace_on()) xsg::trace(__VA_ARGS__); \
  } while (0)

#define HIP_TRY(expr)                                                                                        \
  do {                                                                                                       \
    hipError_t _e = (expr);                                                                                  \
    if (_e != hipSuccess)                                                                                    \
      return xsg::fail(XSG_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define XSG_TRY(expr)            \
  do {                           \
    int _r = (expr);             \
    if (_r != XSG_OK) return _r; \
  } while (0)

// ---------------------------------------------------------------------------
// device buffers (grow-only)
// ---------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  // *grew (optional) is set when the buffer was (re)allocated: its old contents are gone
  int ensure(size_t bytes, bool* grew = nullptr) {
    if (bytes <= cap && p) return XSG_OK;
    if (grew) *grew = true;
    if (bytes == 0) bytes = 16;
    // grow geometrically so that repeated searches with slowly growing results do not re-allocate
    size_t want = std::max(bytes, cap + cap / 2);
    want = (want + 255) & ~(size_t)255;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return xsg::fail(XSG_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    return XSG_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct xsg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<uint8_t> pattern;
  uint32_t flags = 0;
  bool bordered = false;  // the pattern can overlap itself
  // ... a literal one, in these ways: for every border b the word P[0 .. plen - b) + P, the text of two occurrences
  // plen - b apart.  No such word in the data <=> no two occurrences overlap <=> the greedy walk keeps every occurrence
  // (xsg_api.cpp: ensure_overlap_check).  Empty: not checkable (a regex, more than three borders, words too long).
  std::vector<std::vector<uint8_t>> overlap_words;
  DevBuf d_aux_pat;  // device copies of the overlap words of the pattern (ensure_overlap_check), side by side
  uint64_t aux_serial = 0;  // ... of which pattern_serial
  xsg::PatternDev pat{};
  DevBuf d_pat;
  // kDfa with a selective start (xsg_regex.h: RegexDfa::prefix): the class-sequence pattern that finds the candidates
  bool rx_pre = false;
  bool rx_pre_forced = false;  // XSG_RX_PRE=1: on shards of any size
  // kDfa without a selective start but with a factor every match contains (RegexDfa::factor): the class-sequence pattern
  // that finds the lines worth walking
  bool rx_fac = false, rx_fac_forced = false;
  xsg::PatternDev fac_pat{};
  DevBuf d_fac;
  xsg::PatternDev pre_pat{};
  DevBuf d_pre;
  uint32_t tile_bytes = xsg::kDefaultTileBytes;  // geometry new shards get (XSG_TILE_KIB)
  uint32_t tune = xsg::kTuneAuto;                  // wave stagger: per kernel variant (XSG_TUNE overrides)
  uint64_t probe_min_bytes = 64ull << 20;          // shards below this keep the defaults (XSG_PROBE_MIN_BYTES; tests set 0)
  int hot_env = -1;                                // XSG_HOT=0|1 pins the hot filter of the window kinds; -1: measured per shard
  uint64_t pattern_serial = 0;                     // bumped by every xsg_set_pattern
  // the pattern_serial whose lists did not fit the one-sync list route's capacity on SOME binding of this context: later
  // bindings (the file pipeline re-binds its shard for every chunk) go to the exact route at once instead of running the
  // whole one-sync attempt first and then the exact route, chunk after chunk.  A choice, not a result.
  uint64_t fast_dense_serial = 0;
  std::vector<uint32_t> koff_cands;                // long literal patterns: filter windows worth measuring (first: the heuristic's)
  char arch[128] = "";
  int cus = 0;
  uint64_t hbm = 0;
  // What the probe (choose_hot_filter) measured last, kept with the buffer it was measured on: a NEW binding of the same
  // buffer and size with the same pattern (a caller that creates a shard per search) takes it over instead of paying the
  // probe again; xsg_shard_rebind / xsg_shard_invalidate on that buffer drop it (the bytes changed), and so does a
  // binding of the same address and size whose first or last 16 bytes differ (a freed buffer handed out again for other
  // data).  A choice, never a result: a stale one costs speed only.
  struct ProbeMemo {
    uint64_t serial = 0;
    const uint8_t* base = nullptr;
    uint64_t total_bytes = 0, nchunks = 0;
    uint64_t tag[4] = {0, 0, 0, 0};  // the first and the last 16 bytes of the text that was measured
    uint8_t hot_v[4] = {0, 0, 0, 0}, hot_known = 0;  // as xsg_shard::hot_v / hot_known
    uint32_t koff = 0, tune = xsg::kTuneAuto;
    bool koff_chosen = false, tune_probe = false;
  } memo;
};


struct xsg_shard {
  xsg_ctx* ctx = nullptr;
  const uint8_t* base = nullptr;
  uint64_t capacity = 0;
  std::vector<xsg_chunk> chunks;
  std::vector<uint64_t> chunk_tile0;
  uint64_t ntiles = 0;
  uint32_t tile_bytes = xsg::kDefaultTileBytes;
  uint64_t total_bytes = 0;
  uint64_t shard_line_base = 0;
  uint32_t tune = xsg::kTuneAuto;  // wave stagger chosen by xsg_shard_tune (kTuneAuto: per variant / ctx override)
  uint64_t tune_serial = 0;        // ... for this ctx->pattern_serial (0: not bound to a pattern)
  uint64_t overlap_serial = 0;     // ctx->pattern_serial for which `overlap_free` was established on this binding
  bool overlap_free = false;       // ... no two occurrences of the (bordered) pattern overlap anywhere in these chunks
  uint64_t density_serial = 0;     // ctx->pattern_serial for which `dense` was observed (a synchronous count's result)
  uint32_t dense = 0;              // ... 1: more than one result per 8 KiB of this data, 2: more than one per 2 KiB
  bool tune_probe = false;         // `tune` came from choose_hot_filter's two-way probe (re-measured after a re-bind), not from xsg_shard_tune
  // hot filter of the window kinds for (this binding, the ctx's current pattern): measured once PER KERNEL VARIANT the
  // caller's mode launches (bit 0 of the index: the pass also counts newlines; bit 1: it builds line summaries) -- the
  // aligned trigger wins by 12 % where the kernel is VALU-bound (count + newlines) and loses 2 % where it waits for
  // memory (the plain count); see choose_hot_filter
  uint8_t hot_v[4] = {0, 0, 0, 0};
  uint8_t hot_known = 0;  // bit v: hot_v[v] was measured (or settled without measuring) for hot_serial's pattern
  uint32_t koff = 0;  // long patterns: the filter window measured best here
  bool koff_chosen = false;
  uint64_t hot_serial = 0;  // the ctx->pattern_serial `hot` was measured for (0: never)

  DevBuf d_chunks, d_tile_chunk, d_chunk_tile0;
  DevBuf d_tile_cnt, d_tile_nl, d_tile_sum, d_tile_last, d_counters;
  DevBuf d_finish;  // k_count_finish scratch: 3 x kFinishBlocks partial sums, then the ticket word
  DevBuf d_tile_off, d_tile_nl_off, d_scan_tmp;
  DevBuf d_m_pos, d_m_chunk, d_m_ls, d_keep, d_keep_pre;
  DevBuf d_chunk_shift0, d_tail_cnt, d_tail_pos, d_tail_pre;
  DevBuf d_f_pos, d_f_match, d_f_chunk, d_out_u64, d_line_len, d_line_off, d_line_bytes, d_dropped;
  DevBuf d_c_pos, d_c_chunk, d_c_len, d_c_keep, d_c_pre;  // prefilter route of kDfa: the candidates
  DevBuf d_tile_mask;             // factor prefilter of kDfa: tiles in which a line with a factor occurrence starts
  uint64_t mask_serial = 0;       // ... valid for this ctx->pattern_serial on this binding (0: not built)
  uint64_t mask_dense_serial = 0; // ... found useless for this pattern here (factor occurrences too dense)

  // State of the per-tile arrays between passes (host-side bookkeeping; see ScanArgs).  A count pass leaves
  // tile_cnt / tile_sum clean (k_count_finish zeroes what it read); a list pass or a timing loop leaves them
  // dirty and the next pass pays one memset.  tile_last is never cleaned: passes are told apart by `epoch`.
  bool cnt_clean = false, sum_clean = false, last_valid = false;
  uint32_t epoch = 0;
  // The newline counts per tile do not depend on the pattern: computed by the first pass that needs them
  // (k_scan<WANT_NL>), kept until the shard is re-bound; later XSG_LINE_INDICES / XSG_WITH_NEWLINES passes run
  // the plain kernel.
  bool nl_cached = false, nl_off_cached = false;

  // chunk table upload without a host sync (one-chunk shards: the file pipeline re-binds per chunk)
  void* h_stage = nullptr;  // pinned
  hipEvent_t table_ev = nullptr;
  bool table_pending = false;  // an upload has been enqueued on the ctx stream since the last sync
  void* h_result = nullptr;  // pinned: xsg_result_u64_view
  size_t h_result_cap = 0;
  uint64_t* h_counters = nullptr;  // pinned mirror of the four counters (xsg_count reads it after the stream sync)
  bool begin_sync_result = false;  // xsg_count_begin had to run synchronously: _end hands out begin_counters
  uint64_t begin_counters[XSG_NUM_COUNTERS] = {0, 0, 0, 0};

  // The one-sync list route (xsg_api.cpp: run_list_fast): capacities instead of fetched sizes, totals and results
  // mirrored in pinned host memory by the kernels, one stream sync per search.
  DevBuf d_tot;                    // FastTot words, then u32 ticket words
  DevBuf d_hit;                    // tiles that hold a match, in order
  DevBuf d_scan2;                  // scratch of the two-launch scans
  DevBuf d_wmask;                  // ScanArgs::tile_wmask
  uint64_t* h_tot = nullptr;       // pinned mirror of the FastTot words (+ one word for the scan flags)
  uint64_t* hp_line_len = nullptr; // pinned: xs::lines lengths (UINT64_MAX = dropped)
  size_t hp_line_len_cap = 0;      // entries
  uint8_t* hp_line_bytes = nullptr;
  size_t hp_line_bytes_cap = 0;
  uint64_t fast_dense_serial = 0;  // ctx->pattern_serial whose result did not fit the route's capacity on this binding (0: none)
  bool fast_result = false;        // the pending result lives in the pinned mirrors (h_result, hp_line_*)
  bool line_len_on_device = false; // xs::lines on the exact route: the lengths have not been copied to hp_line_len yet (fetch_line_lengths)
  uint32_t h_dropped = 0;          // ... lines without a terminating newline among them (counted by k_line_lengths)
  uint64_t fast_raw_lines = 0;     // xs::lines on that route: entries of hp_line_len (dropped lines included)
  uint64_t nl_total = 0;           // '\n' in the shard, valid while nl_off_cached

  uint64_t pre_dense_serial = 0;  // the ctx->pattern_serial whose prefilter candidates were found dense on this binding (0: none)
  bool pre_off = false;        // run_list: this call must not take the prefilter route (its verification budget ran out)
  bool want_nl_total = false;  // run_list: also leave the shard's newline total in last_newlines (xsg_count on the prefilter route)
  int last_mode = -1;
  uint64_t last_raw_matches = 0;  // raw occurrences of the last list pass (capacity hint for xsg_count_async, bordered patterns)
  uint64_t total = 0;       // elements of the last list search
  uint64_t line_bytes = 0;  // XSG_LINES: packed bytes
  uint64_t last_newlines = 0;  // XSG_LINE_INDICES: '\n' in the shard (for chaining line bases)

  void release_all() {
    DevBuf* all[] = {&d_chunks, &d_tile_chunk, &d_chunk_tile0, &d_tile_cnt, &d_tile_nl, &d_tile_sum, &d_tile_last,
                     &d_counters, &d_finish, &d_tile_off, &d_tile_nl_off, &d_scan_tmp, &d_m_pos, &d_m_chunk, &d_m_ls,
                     &d_keep, &d_keep_pre, &d_chunk_shift0, &d_tail_cnt, &d_tail_pos, &d_tail_pre, &d_f_pos, &d_f_match,
                     &d_f_chunk, &d_out_u64, &d_line_len, &d_line_off, &d_line_bytes, &d_dropped, &d_c_pos, &d_c_chunk, &d_c_len, &d_c_keep,
                     &d_c_pre, &d_tile_mask, &d_tot, &d_hit, &d_scan2, &d_wmask};
    for (DevBuf* b : all) b->release();
    if (h_stage) (void)hipHostFree(h_stage);
    if (h_counters) (void)hipHostFree(h_counters);
    if (h_result) (void)hipHostFree(h_result);
    h_result = nullptr;
    h_result_cap = 0;
    if (h_tot) (void)hipHostFree(h_tot);
    if (hp_line_len) (void)hipHostFree(hp_line_len);
    if (hp_line_bytes) (void)hipHostFree(hp_line_bytes);
    h_tot = nullptr;
    hp_line_len = nullptr;
    hp_line_bytes = nullptr;
    hp_line_len_cap = hp_line_bytes_cap = 0;
    if (table_ev) (void)hipEventDestroy(table_ev);
    h_stage = nullptr;
    h_counters = nullptr;
    table_ev = nullptr;
  }
};

