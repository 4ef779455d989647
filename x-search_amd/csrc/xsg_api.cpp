// xsg_api.cpp -- the C ABI of include/xsg.h: contexts, shards, count and list
// searches on device-resident chunks.  Host glue only; the kernels are in
// xsg_kernels.hip.  There is no CPU fallback anywhere in this file: every
// compute entry point needs a HIP device and fails with XSG_ENODEV/XSG_EHIP
// otherwise.
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "xsg_classseq.h"
#include "xsg_regex.h"
#include "xsg_objects.h"
#include "xsg_linesum.h"
#include "xsg_tail.h"

using namespace xsg;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

namespace xsg {
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
const char* last_error_message() { return g_err; }

static const std::chrono::steady_clock::time_point g_loaded = std::chrono::steady_clock::now();
bool test_hooks() {
  static const bool on = [] { const char* e = getenv("XSG_TEST_HOOKS"); return e && *e == '1'; }();
  return on;
}
bool trace_on() {
  static const bool on = [] { const char* e = getenv("XSG_TRACE"); return e && *e && *e != '0'; }();
  return on;
}
void trace(const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  const auto now = std::chrono::steady_clock::now();
  const double ms = std::chrono::duration<double, std::milli>(now - g_loaded).count();
  // (the second figure is CLOCK_MONOTONIC in seconds: a launcher that prints the same clock before it starts the
  // process and after it has exited -- scripts/cli_trace.py -- shows what lies before the library is loaded and behind
  // the last mark: the process's start and its teardown)
  fprintf(stderr, "[xsg +%10.3f ms | %.6f] %s\n", ms, std::chrono::duration<double>(now.time_since_epoch()).count(), buf);
}
}  // namespace xsg

extern "C" int xsg_abi_version(void) { return XSG_ABI_VERSION; }

extern "C" const char* xsg_strerror(int code) {
  switch (code) {
    case XSG_OK: return "ok";
    case XSG_EINVAL: return "invalid argument";
    case XSG_ENODEV: return "no usable HIP device";
    case XSG_EHIP: return "HIP runtime error";
    case XSG_ENOMEM: return "out of memory";
    case XSG_ENOTSUP: return "not supported by this entry point";
    case XSG_EIO: return "I/O error";
    case XSG_ESTATE: return "call sequence error";
    default: return "unknown error";
  }
}
extern "C" const char* xsg_last_error(void) { return g_err; }

extern "C" int xsg_device_count(int* count) {
  if (!count) return fail(XSG_EINVAL, "count is null");
  int n = 0;
  XSG_TRACE("hipGetDeviceCount ...");
  hipError_t e = hipGetDeviceCount(&n);
  XSG_TRACE("hipGetDeviceCount -> %d", n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(XSG_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return XSG_OK;
}

extern "C" int xsg_ctx_create(int device, xsg_ctx** out) {
  if (!out) return fail(XSG_EINVAL, "out is null");
  *out = nullptr;
  int n = 0;
  XSG_TRY(xsg_device_count(&n));
  if (n <= 0) return fail(XSG_ENODEV, "no HIP device visible");
  if (device < 0 || device >= n) return fail(XSG_ENODEV, "device %d out of range (0..%d)", device, n - 1);
  HIP_TRY(hipSetDevice(device));
  XSG_TRACE("ctx_create: hipSetDevice(%d) done", device);
  xsg_ctx* c = new (std::nothrow) xsg_ctx();
  if (!c) return fail(XSG_ENOMEM, "host allocation failed");
  c->device = device;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  XSG_TRACE("ctx_create: device properties");
  if (e != hipSuccess) {
    delete c;
    return fail(XSG_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  }
  snprintf(c->arch, sizeof c->arch, "%s", prop.gcnArchName);
  c->cus = prop.multiProcessorCount;
  c->hbm = prop.totalGlobalMem;
  if (strncmp(c->arch, "gfx950", 6) != 0) {
    // the code objects in this library are gfx950 only
    std::string a = c->arch;
    delete c;
    return fail(XSG_ENODEV, "device %d is %s; this library carries gfx950 (MI355X) code only", device, a.c_str());
  }
  if (const char* tn = getenv("XSG_TUNE")) c->tune = (uint32_t)strtoul(tn, nullptr, 0);
  if (const char* hf = getenv("XSG_HOT")) c->hot_env = (*hf == '0' || *hf == '1') ? *hf - '0' : -1;
  if (const char* pm = getenv("XSG_PROBE_MIN_BYTES")) c->probe_min_bytes = strtoull(pm, nullptr, 0);
  if (const char* tk = getenv("XSG_TILE_KIB")) {
    const int v = atoi(tk);
    if (v == 16) c->tile_bytes = (uint32_t)v * 1024u;
  }
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(XSG_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  XSG_TRACE("ctx_create: stream");
  // The code object of the scan kernels (3.5 MB, ~5 ms the first time in a process) is loaded now rather than inside
  // the first search -- every search launches one of them.  The list kernels' (1.1 ms) and the automaton route's
  // (0.5 ms) are loaded by the first search that needs them: a count of a literal needs neither
  // (profiles/r04_cli_start.txt; XSG_WARM_ALL=1 loads all three here, as round 3 did).
  e = warm_scan_kernels(c->stream);
  XSG_TRACE("ctx_create: scan kernels launched");
  static const bool warm_all = [] { const char* w = getenv("XSG_WARM_ALL"); return w && *w == '1'; }();
  if (warm_all) {
    if (e == hipSuccess) e = warm_list_kernels(c->stream);
    if (e == hipSuccess) e = warm_rx_kernels(c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  XSG_TRACE("ctx_create: warm-up synchronised");
  if (e != hipSuccess) {
    (void)hipStreamDestroy(c->stream);
    delete c;
    return fail(XSG_EHIP, "loading the kernels failed: %s", hipGetErrorString(e));
  }
  *out = c;
  return XSG_OK;
}

extern "C" void xsg_ctx_destroy(xsg_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  c->d_pat.release();
  c->d_pre.release();
  c->d_fac.release();
  c->d_aux_pat.release();
  delete c;
}

extern "C" int xsg_ctx_info(xsg_ctx* c, char* arch, size_t arch_cap, int* compute_units, uint64_t* hbm_bytes) {
  if (!c) return fail(XSG_EINVAL, "ctx is null");
  if (arch && arch_cap) snprintf(arch, arch_cap, "%s", c->arch);
  if (compute_units) *compute_units = c->cus;
  if (hbm_bytes) *hbm_bytes = c->hbm;
  return XSG_OK;
}

static uint32_t le32(const uint8_t* p, size_t n) {
  uint32_t v = 0;
  for (size_t i = 0; i < 4 && i < n; ++i) v |= (uint32_t)p[i] << (8 * i);
  return v;
}
// Which 8 bytes of a long pattern should the hot loop look for?  The slow path runs for
// every wave-load that holds the window somewhere, so the window should be rare in
// text: one that spans a word boundary is (a pair of words is far rarer than either
// word), then upper case / digits / non-ASCII, then the rarer letters.  Static
// heuristic, no look at the data; `detective street` -> "ective s".
static int byte_rarity(uint8_t c, int pos_in_window) {
  const bool lower = c >= 'a' && c <= 'z', upper = c >= 'A' && c <= 'Z', digit = c >= '0' && c <= '9';
  if (c >= 0x80) return 30;
  if (!lower && !upper && !digit) return (pos_in_window >= 1 && pos_in_window <= 6) ? 40 : 10;
  if (upper || digit) return 12;
  if (strchr("jqxzvkwbypgf", c)) return 6;
  return 1;
}

static uint32_t pick_filter_window(const uint8_t* p, size_t plen) {
  if (plen <= 8) return 0;
  uint32_t best = 0;
  int best_score = -1;
  for (size_t k = 0; k + 8 <= plen; ++k) {
    int score = 0;
    for (int i = 0; i < 8; ++i) score += byte_rarity(p[k + i], i);
    if (score > best_score) {
      best_score = score;
      best = (uint32_t)k;
    }
  }
  return best;
}

static uint32_t mask32(size_t n) { return n >= 4 ? 0xffffffffu : (n == 0 ? 0u : ((1u << (8 * n)) - 1u)); }

// the window-dependent fields of a literal pattern: the 8 bytes at p[koff..] as compare dwords and masks
static void window_fields(const uint8_t* p, size_t plen, uint32_t koff, PatternDev* P) {
  const uint8_t* w = p + koff;
  const size_t wlen = plen - koff;  // >= 8 when koff > 0
  P->koff = koff;
  P->p0 = le32(w, wlen);
  P->m0 = mask32(wlen);
  P->p1 = wlen > 4 ? le32(w + 4, wlen - 4) : 0u;
  P->m1 = wlen > 4 ? mask32(wlen - 4) : 0u;
  P->q0 = (P->p0 | 0x20202020u) & P->m0;
  P->q1 = (P->p1 | 0x20202020u) & P->m1;
  // (x | 0x20) == (p | 0x20) holds exactly for x in {p, p - 32} when p is a lower-case letter (the pattern is
  // already lowered): a window of letters only needs no second look under ignore_case
  bool letters = true;
  for (size_t i = 0; i < 8 && i < wlen; ++i) letters &= w[i] >= 'a' && w[i] <= 'z';
  P->lazy_exact = letters ? 1u : 0u;
}

// Long patterns: the windows worth MEASURING on the data (choose_hot_filter): the static heuristic's pick first, then
// the next best-looking ones -- every position for patterns up to 20 bytes, the eight best scores beyond.
static std::vector<uint32_t> window_candidates(const uint8_t* p, size_t plen) {
  std::vector<uint32_t> out;
  if (plen <= 8) return out;
  std::vector<std::pair<int, uint32_t>> scored;
  for (size_t k = 0; k + 8 <= plen; ++k) {
    int score = 0;
    for (int i = 0; i < 8; ++i) score += byte_rarity(p[k + i], i);
    scored.push_back({-score, (uint32_t)k});
  }
  std::stable_sort(scored.begin(), scored.end());
  const size_t n = plen <= 20 ? scored.size() : std::min<size_t>(scored.size(), 8);
  for (size_t i = 0; i < n; ++i) out.push_back(scored[i].second);
  return out;
}

// The window-filter fields of a class expression and the device image of its sets (alternative-major, 32 bytes per
// set) -- for a pattern of its own (set_class_pattern) and for the prefilter of the automaton route (set_dfa_pattern).
static void class_fields(const xsg::ClassExpr& ex, bool icase, PatternDev* Pout, std::vector<uint8_t>* blob) {
  const size_t plen = ex.npos;
  const std::vector<xsg::ByteSet> seq = xsg::union_sets(ex);
  // What the window compare can know about a position: the bits all members of its set agree on (a literal: all
  // eight; [Ss]: seven; [0-9]: the upper four; [a-z]: the upper three).  (x & agree) == (member & agree) holds for
  // every member x, so it is a superset filter at no cost -- the compare is masked per byte anyway -- and the
  // exact decision against the sets follows for the rare candidate.  With several alternatives the sets are the
  // position-wise unions.
  std::vector<uint8_t> agree(plen), value(plen);
  for (size_t k = 0; k < plen; ++k) {
    int first = -1;
    uint32_t diff = 0;
    for (uint32_t b = 0; b < 256; ++b)
      if (xsg::set_has(seq[k], b)) {
        if (first < 0) first = (int)b;
        diff |= b ^ (uint32_t)first;
      }
    agree[k] = (uint8_t)~diff;
    value[k] = (uint8_t)((uint32_t)first & ~diff);
  }
  // the window that pins the most bits (rarer literal bytes break ties)
  uint32_t koff = 0;
  int best = -1;
  for (size_t k = 0; k < plen; ++k) {
    int score = 0;
    for (int i = 0; i < 8 && k + i < plen; ++i) {
      score += 16 * __builtin_popcount(agree[k + i]);
      if (agree[k + i] == 0xff) score += byte_rarity(value[k + i], i);
    }
    // a window whose positions 0, 1 and 4..7 are single bytes takes the exact 16 + 32 bit filter (PatternDev::cls_fast)
    if (k + 8 <= plen && agree[k] == 0xff && agree[k + 1] == 0xff && agree[k + 4] == 0xff && agree[k + 5] == 0xff &&
        agree[k + 6] == 0xff && agree[k + 7] == 0xff)
      score += 48;
    if (score > best) best = score, koff = (uint32_t)k;
  }
  uint32_t pw[2] = {0, 0}, mw[2] = {0, 0};
  for (int i = 0; i < 8 && koff + i < plen; ++i) {
    pw[i >> 2] |= (uint32_t)value[koff + i] << (8 * (i & 3));
    mw[i >> 2] |= (uint32_t)agree[koff + i] << (8 * (i & 3));
  }
  constexpr size_t kSetBytes = xsg::kMaxAltSets * sizeof(xsg::ByteSet);
  blob->assign(std::max<size_t>(XSG_MAX_REGEX, kSetBytes) + 16, 0);
  for (size_t a = 0; a < ex.alts.size(); ++a)
    memcpy(blob->data() + a * plen * sizeof(xsg::ByteSet), ex.alts[a].data(), plen * sizeof(xsg::ByteSet));
  PatternDev& P = *Pout;
  P = PatternDev{};
  P.plen = (uint32_t)plen;
  P.kind = kClass;
  P.koff = koff;
  P.p0 = pw[0], P.m0 = mw[0], P.p1 = pw[1], P.m1 = mw[1];
  P.q0 = (P.p0 | 0x20202020u) & P.m0, P.q1 = (P.p1 | 0x20202020u) & P.m1;
  {
    const char* cf = XSG_TOGGLE("XSG_CLS_FAST");
    P.cls_fast = (P.m1 == 0xffffffffu && (P.m0 & 0xffffu) == 0xffffu && !(cf && *cf == '0')) ? 1u : 0u;
  }
  P.exact_tail = 1u;
  P.icase = icase ? 1u : 0u;
  P.nalt = (uint32_t)ex.alts.size();
  // Up to 8 positions: the filter window is the whole expression and candidates are decided in registers.  A position
  // needs no look at its set when the hot filter's compare already decides it exactly: one alternative, and the set
  // is precisely the bytes that agree with `value` under `agree` (a literal; [Ss]; [a-z] is not: 0x60-0x7f pass the
  // compare) -- and, under the 16 + 32 bit filter, the position is not one of the two that filter leaves out.
  // With several alternatives only a position that is one byte in all of them is decided (the compare sees the union).
  {
    const char* ir = XSG_TOGGLE("XSG_CLS_INREG");
    P.cls_chk = 0;
    for (size_t k = 0; k < plen && k < 8; ++k) {
      bool decided = true;
      for (uint32_t b = 0; b < 256 && decided; ++b)
        decided = xsg::set_has(seq[k], b) == ((b & agree[k]) == value[k]);
      if (ex.alts.size() > 1 && agree[k] != 0xff) decided = false;
      if (P.cls_fast && (k == 2 || k == 3)) decided = false;  // (the aligned trigger's slow path uses the full masks, but one table serves both)
      if (!decided) P.cls_chk |= 1u << k;
    }
    // Measured on whole calls (scripts/ab_inreg.py, 20 GiB): one alternative with something left to look up wins
    // (`She[r ]lock` 4.38 -> 4.05 ms); several alternatives lose (the candidate scan of `Sherlock|Holmes`, two
    // alternatives, dense candidates: 33 -> 61 ms: a scalar loop per alternative and position); and an expression the
    // compare decides completely (one alternative, nothing to look up: `Sher`, `[Ss]herlock`) needs no verification
    // at all -- its candidate bits ARE its matches (cls_exact).
    const bool one = ex.alts.size() == 1 && plen <= 8 && koff == 0;
    P.cls_inreg = (one && P.cls_chk != 0 && !(ir && *ir == '0')) ? 1u : 0u;
    P.cls_exact = (one && P.cls_chk == 0 && !P.cls_fast && !(ir && *ir == '0')) ? 1u : 0u;
  }
  P.ascii_only = ex.ascii_only ? 1u : 0u;
  P.has_newline = 0;
  for (const xsg::ByteSet& st : seq) P.has_newline |= xsg::set_has(st, '\n') ? 1u : 0u;
}

// layout of the device copy of a RegexDfa: class_of[256], then the forward table, then (16-byte aligned) the reverse one
static size_t rx_rev_offset(uint32_t fwd_entries) { return (256 + 2 * (size_t)fwd_entries + 15) & ~(size_t)15; }

// XSG_FLAG_REGEX, second route: an expression of variable length, as a pair of byte-class DFAs for k_rx_scan
// (xsg_regex.h).  `why_not_class`: what the class-sequence compiler said, for the message if this route refuses too.
static int set_dfa_pattern(xsg_ctx* c, const uint8_t* re, size_t n, uint32_t flags, const std::string& why_not_class) {
  xsg::RegexDfa dfa;
  std::string err;
  const bool icase = (flags & XSG_FLAG_IGNORE_CASE) != 0;
  if (!xsg::compile_regex_dfa(re, n, icase, &dfa, &err))
    return fail(XSG_ENOTSUP, "regex not supported by the GPU matchers: %s [as a fixed-length expression: %s]", err.c_str(),
                why_not_class.c_str());
  HIP_TRY(hipSetDevice(c->device));
  c->pattern.assign(re, re + n);
  c->flags = flags;
  ++c->pattern_serial;
  c->koff_cands.clear();
  c->bordered = false;  // the kernel walks every line as the reference does: what it reports is already non-overlapping
  c->overlap_words.clear();
  const size_t rev_off = rx_rev_offset((uint32_t)dfa.fwd.size());
  const size_t anc_off = (rev_off + 2 * dfa.rev.size() + 15) & ~(size_t)15;
  const size_t bytes = anc_off + 2 * dfa.anc.size() + 16;
  std::vector<uint8_t> blob(bytes, 0);
  memcpy(blob.data() + anc_off, dfa.anc.data(), 2 * dfa.anc.size());
  memcpy(blob.data(), dfa.class_of, 256);
  // Trigger bytes: those that move the forward automaton out of its start state (a byte that cannot begin a match
  // leaves it there), and '\n'.  Flagged in bit 7 of the class table; k_rx_scan's walks jump from trigger to trigger.
  const char* skip_env = XSG_TOGGLE("XSG_RX_SKIP");
  bool skip = dfa.ncls <= 128 && !dfa.multiline && !(skip_env && *skip_env == '0');
  if (skip && !(skip_env && *skip_env == '1')) {
    // skipping pays when triggers are rare in the data; an expression that can begin with most letters (`\\w+ing`)
    // triggers at every word and the jumps cost more than the steps they replace (measured: 156 against 201 GB/s)
    uint32_t common = 0;
    for (uint32_t b = 'a'; b <= 'z'; ++b)
      common += dfa.fwd[(size_t)dfa.fwd_start * dfa.ncls + dfa.class_of[b]] != dfa.fwd_start * dfa.ncls;
    if (common >= 9) skip = false;
  }
  if (skip)
    for (uint32_t b = 0; b < 256; ++b)
      if (b == '\n' || dfa.fwd[(size_t)dfa.fwd_start * dfa.ncls + dfa.class_of[b]] != dfa.fwd_start * dfa.ncls) blob[b] |= 0x80u;
  memcpy(blob.data() + 256, dfa.fwd.data(), 2 * dfa.fwd.size());
  memcpy(blob.data() + rev_off, dfa.rev.data(), 2 * dfa.rev.size());
  XSG_TRY(c->d_pat.ensure(std::max<size_t>(bytes, XSG_MAX_REGEX + 16)));
  HIP_TRY(hipMemcpyAsync(c->d_pat.p, blob.data(), bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  PatternDev& P = c->pat;
  P = PatternDev{};
  P.plen = dfa.minlen;  // what the list kernels may skip behind a match start before they look for the line's end
  P.kind = kDfa;
  P.d_pat = c->d_pat.as<uint8_t>();
  P.exact_tail = 1u;
  P.icase = 0u;  // the sets are closed under case; the data is not folded
  P.ascii_only = dfa.ascii_only ? 1u : 0u;
  P.has_newline = dfa.multiline ? 1u : 0u;  // a match may span lines: the line tags are refused, as for a literal with '\n'
  P.rx_multiline = dfa.multiline ? 1u : 0u;
  P.rx_ncls = dfa.ncls;
  P.rx_fwd_n = (uint32_t)dfa.fwd.size();
  P.rx_rev_n = (uint32_t)dfa.rev.size();
  P.rx_fwd_start = dfa.fwd_start * dfa.ncls;
  P.rx_fwd_acc = dfa.fwd_first_acc * dfa.ncls;
  P.rx_rev_start = dfa.rev_start * dfa.ncls;
  P.rx_rev_acc = dfa.rev_first_acc * dfa.ncls;
  P.rx_skip = skip ? 1u : 0u;
  // few trigger byte values (`Sherlock|Holmes`: S, H; `Sher.*mes`: S; closed under case: up to four): k_rx_scan looks for
  // them with byte-parallel compares on its loads and does not stage a tile that holds none (XSG_RX_TRIG=0 switches it off)
  if (skip) {
    const char* te = XSG_TOGGLE("XSG_RX_TRIG");
    uint32_t n = 0, packed = 0;
    for (uint32_t b = 0; b < 256; ++b)
      if (b != '\n' && (blob[b] & 0x80u)) {
        if (n < 4) packed |= b << (8 * n);
        ++n;
      }
    if (n >= 1 && n <= 4 && !(te && *te == '0')) {
      for (uint32_t k = n; k < 4; ++k) packed |= (packed & 0xffu) << (8 * k);  // unused slots repeat the first value
      P.rx_ntrig = n;
      P.rx_trig4 = packed;
    }
  }
  P.rx_anc_n = (uint32_t)dfa.anc.size();
  P.rx_anc_start = dfa.anc_start * dfa.ncls;
  P.rx_anc_acc = dfa.anc_first_acc * dfa.ncls;
  // A selective start: the synchronous entry points find candidates with the class-sequence matcher and verify them
  // (rx_pre_matches); xsg_count_async, which may not wait for the host, keeps k_rx_scan.  XSG_RX_PRE=0 switches it off.
  const char* pre_env = XSG_TOGGLE("XSG_RX_PRE");
  c->rx_pre = dfa.prefix.npos != 0 && !(pre_env && *pre_env == '0');
  c->rx_pre_forced = pre_env && *pre_env == '1';  // on shards of any size (tests; by default only where it pays, use_prefilter)
  if (c->rx_pre) {
    std::vector<uint8_t> pblob;
    class_fields(dfa.prefix, false, &c->pre_pat, &pblob);  // the sets are closed under case already: no folding
    XSG_TRY(c->d_pre.ensure(pblob.size()));
    HIP_TRY(hipMemcpyAsync(c->d_pre.p, pblob.data(), pblob.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->pre_pat.d_pat = c->d_pre.as<uint8_t>();
    c->pre_pat.ascii_only = P.ascii_only;  // the candidate scan reads every byte: it raises the refusal flag
  }
  // No selective start, but a factor every match contains (`\\w+ing`: `\\wing`): lines without it have no match, and the
  // synchronous entry points first mark the tiles in which a line with an occurrence starts (ensure_factor_mask).
  const char* fac_env = XSG_TOGGLE("XSG_RX_FAC");
  c->rx_fac = !c->rx_pre && dfa.factor.npos != 0 && !(fac_env && *fac_env == '0');
  c->rx_fac_forced = fac_env && *fac_env == '1';
  if (c->rx_fac) {
    std::vector<uint8_t> fblob;
    class_fields(dfa.factor, false, &c->fac_pat, &fblob);
    XSG_TRY(c->d_fac.ensure(fblob.size()));
    HIP_TRY(hipMemcpyAsync(c->d_fac.p, fblob.data(), fblob.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->fac_pat.d_pat = c->d_fac.as<uint8_t>();
    c->fac_pat.ascii_only = P.ascii_only;
  }
  return XSG_OK;
}

// XSG_FLAG_REGEX: a fixed-length class sequence (xsg_classseq.h).  RE2 has no lossy tail, so the
// matching is exact up to the end of the chunk (as with XSG_FLAG_EXACT_TAIL).
static int set_class_pattern(xsg_ctx* c, const uint8_t* re, size_t n, uint32_t flags) {
  xsg::ClassExpr ex;
  std::string err;
  const bool icase = (flags & XSG_FLAG_IGNORE_CASE) != 0;
  if (!xsg::compile_class_expr(re, n, icase, &ex, &err)) return set_dfa_pattern(c, re, n, flags, err);
  const size_t plen = ex.npos;
  const std::vector<xsg::ByteSet> seq = xsg::union_sets(ex);  // what the filter, the overlap and '\n' tests look at
  bool literal = ex.alts.size() == 1;
  std::vector<uint8_t> lit(plen);
  for (size_t k = 0; k < plen; ++k) {
    const int b = xsg::set_single(seq[k]);
    literal &= b >= 0;
    lit[k] = (uint8_t)(b >= 0 ? b : 0);
  }
  if (literal)  // e.g. `a\.b`: an ordinary literal, minus the reference's scalar-tail quirk (RE2 has none)
    return xsg_set_pattern(c, lit.data(), plen, (flags & XSG_FLAG_IGNORE_CASE) | XSG_FLAG_EXACT_TAIL);

  HIP_TRY(hipSetDevice(c->device));
  c->pattern.assign(re, re + n);
  c->flags = flags;
  ++c->pattern_serial;
  c->koff_cands.clear();
  c->bordered = xsg::sequence_can_overlap(seq);
  c->overlap_words.clear();
  std::vector<uint8_t> blob;
  PatternDev P;
  class_fields(ex, icase, &P, &blob);
  XSG_TRY(c->d_pat.ensure(blob.size()));
  HIP_TRY(hipMemcpyAsync(c->d_pat.p, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  P.d_pat = c->d_pat.as<uint8_t>();
  c->pat = P;
  c->rx_pre = false;
  c->rx_fac = false;
  return XSG_OK;
}

static int dfa_route_serves(const void* expr, size_t n, uint32_t flags, const std::string& why_not_class, xsg::RegexDfa* out) {
  xsg::RegexDfa local;
  std::string err;
  if (!xsg::compile_regex_dfa(static_cast<const uint8_t*>(expr), n, (flags & XSG_FLAG_IGNORE_CASE) != 0, out ? out : &local, &err))
    return fail(XSG_ENOTSUP, "regex not supported by the GPU matchers: %s [as a fixed-length expression: %s]", err.c_str(),
                why_not_class.c_str());
  return XSG_OK;
}

extern "C" int xsg_regex_dfa_info(const void* expr, size_t n, uint32_t flags, xsg_regex_dfa* info, uint16_t* fwd,
                                  uint16_t* rev, size_t cap_entries) {
  if (!expr || n == 0) return fail(XSG_EINVAL, "empty expression");
  if (n > XSG_MAX_REGEX) return fail(XSG_EINVAL, "expression longer than %u bytes", XSG_MAX_REGEX);
  if (!info) return fail(XSG_EINVAL, "info is null");
  xsg::RegexDfa dfa;
  std::string err;
  if (!xsg::compile_regex_dfa(static_cast<const uint8_t*>(expr), n, (flags & XSG_FLAG_IGNORE_CASE) != 0, &dfa, &err))
    return fail(XSG_ENOTSUP, "regex not supported by the automaton route: %s", err.c_str());
  info->ncls = dfa.ncls, info->minlen = dfa.minlen, info->ascii_only = dfa.ascii_only ? 1u : 0u;
  info->multiline = dfa.multiline ? 1u : 0u;
  info->prefix_positions = dfa.prefix.npos;
  info->prefix_alternatives = (uint32_t)dfa.prefix.alts.size();
  info->factor_positions = dfa.factor.npos;
  info->fwd_states = dfa.fwd_states, info->fwd_start = dfa.fwd_start, info->fwd_first_acc = dfa.fwd_first_acc;
  info->rev_states = dfa.rev_states, info->rev_start = dfa.rev_start, info->rev_first_acc = dfa.rev_first_acc;
  memcpy(info->class_of, dfa.class_of, 256);
  if (fwd && cap_entries >= dfa.fwd.size()) memcpy(fwd, dfa.fwd.data(), 2 * dfa.fwd.size());
  if (rev && cap_entries >= dfa.rev.size()) memcpy(rev, dfa.rev.data(), 2 * dfa.rev.size());
  return XSG_OK;
}

extern "C" int xsg_regex_prefix(const void* expr, size_t n, uint32_t flags, uint32_t* positions, uint32_t* alternatives,
                                uint32_t* sets) {
  if (!expr || n == 0) return fail(XSG_EINVAL, "empty expression");
  if (n > XSG_MAX_REGEX) return fail(XSG_EINVAL, "expression longer than %u bytes", XSG_MAX_REGEX);
  xsg::RegexDfa dfa;
  std::string err;
  if (!xsg::compile_regex_dfa(static_cast<const uint8_t*>(expr), n, (flags & XSG_FLAG_IGNORE_CASE) != 0, &dfa, &err))
    return fail(XSG_ENOTSUP, "regex not supported by the automaton route: %s", err.c_str());
  if (positions) *positions = dfa.prefix.npos;
  if (alternatives) *alternatives = (uint32_t)dfa.prefix.alts.size();
  if (sets)
    for (size_t a = 0; a < dfa.prefix.alts.size(); ++a)
      memcpy(sets + a * dfa.prefix.npos * 8, dfa.prefix.alts[a].data(), dfa.prefix.npos * sizeof(xsg::ByteSet));
  return XSG_OK;
}

extern "C" int xsg_regex_factor(const void* expr, size_t n, uint32_t flags, uint32_t* positions, uint32_t* sets) {
  if (!expr || n == 0) return fail(XSG_EINVAL, "empty expression");
  if (n > XSG_MAX_REGEX) return fail(XSG_EINVAL, "expression longer than %u bytes", XSG_MAX_REGEX);
  xsg::RegexDfa dfa;
  std::string err;
  if (!xsg::compile_regex_dfa(static_cast<const uint8_t*>(expr), n, (flags & XSG_FLAG_IGNORE_CASE) != 0, &dfa, &err))
    return fail(XSG_ENOTSUP, "regex not supported by the automaton route: %s", err.c_str());
  if (positions) *positions = dfa.factor.npos;
  if (sets && dfa.factor.npos) memcpy(sets, dfa.factor.alts[0].data(), dfa.factor.npos * sizeof(xsg::ByteSet));
  return XSG_OK;
}

extern "C" int xsg_regex_check(const void* expr, size_t n, uint32_t flags, uint32_t* positions, uint32_t* sets) {
  if (!expr || n == 0) return fail(XSG_EINVAL, "empty expression");
  if (n > XSG_MAX_REGEX) return fail(XSG_EINVAL, "expression longer than %u bytes", XSG_MAX_REGEX);
  xsg::ClassExpr ex;
  std::string err;
  if (!xsg::compile_class_expr(static_cast<const uint8_t*>(expr), n, (flags & XSG_FLAG_IGNORE_CASE) != 0, &ex, &err)) {
    XSG_TRY(dfa_route_serves(expr, n, flags, err, nullptr));
    if (positions) *positions = 0;  // variable length: no position-wise sets; such an expression never accepts '\n'
    return XSG_OK;
  }
  const std::vector<xsg::ByteSet> seq = xsg::union_sets(ex);
  if (positions) *positions = (uint32_t)seq.size();
  if (sets) memcpy(sets, seq.data(), seq.size() * sizeof(xsg::ByteSet));
  return XSG_OK;
}

extern "C" int xsg_regex_info(const void* expr, size_t n, uint32_t flags, uint32_t* positions, uint32_t* alternatives,
                              uint32_t* ascii_only, uint32_t* sets) {
  if (!expr || n == 0) return fail(XSG_EINVAL, "empty expression");
  if (n > XSG_MAX_REGEX) return fail(XSG_EINVAL, "expression longer than %u bytes", XSG_MAX_REGEX);
  xsg::ClassExpr ex;
  std::string err;
  if (!xsg::compile_class_expr(static_cast<const uint8_t*>(expr), n, (flags & XSG_FLAG_IGNORE_CASE) != 0, &ex, &err)) {
    xsg::RegexDfa dfa;
    XSG_TRY(dfa_route_serves(expr, n, flags, err, &dfa));
    if (positions) *positions = 0;
    if (alternatives) *alternatives = 0;
    if (ascii_only) *ascii_only = dfa.ascii_only ? 1u : 0u;
    return XSG_OK;
  }
  if (positions) *positions = ex.npos;
  if (alternatives) *alternatives = (uint32_t)ex.alts.size();
  if (ascii_only) *ascii_only = ex.ascii_only ? 1u : 0u;
  if (sets)
    for (size_t a = 0; a < ex.alts.size(); ++a)
      memcpy(sets + a * ex.npos * 8, ex.alts[a].data(), ex.npos * sizeof(xsg::ByteSet));
  return XSG_OK;
}

extern "C" int xsg_set_pattern(xsg_ctx* c, const void* pattern, size_t plen, uint32_t flags) {
  if (!c) return fail(XSG_EINVAL, "ctx is null");
  if (!pattern || plen == 0) return fail(XSG_EINVAL, "empty pattern");
  if (plen > XSG_MAX_PATTERN) return fail(XSG_EINVAL, "pattern longer than %u bytes", XSG_MAX_PATTERN);
  if (flags & ~(XSG_FLAG_EXACT_TAIL | XSG_FLAG_IGNORE_CASE | XSG_FLAG_REGEX))
    return fail(XSG_EINVAL, "unknown pattern flags 0x%x", flags);
  if (flags & XSG_FLAG_REGEX) {
    if (plen > XSG_MAX_REGEX) return fail(XSG_EINVAL, "expression longer than %u bytes", XSG_MAX_REGEX);
    return set_class_pattern(c, static_cast<const uint8_t*>(pattern), plen, flags);
  }
  HIP_TRY(hipSetDevice(c->device));
  c->pattern.assign(static_cast<const uint8_t*>(pattern), static_cast<const uint8_t*>(pattern) + plen);
  if (flags & XSG_FLAG_IGNORE_CASE)  // simd::toLower on the pattern (string_utils.cpp:11-33)
    for (uint8_t& b : c->pattern)
      if (b >= 'A' && b <= 'Z') b = (uint8_t)(b + 32);
  const uint8_t* p = c->pattern.data();
  c->flags = flags;
  ++c->pattern_serial;
  // border <=> the pattern can overlap itself (KMP failure function of the last position > 0)
  std::vector<uint32_t> pi(plen, 0);
  for (size_t i = 1, k = 0; i < plen; ++i) {
    while (k > 0 && p[i] != p[k]) k = pi[k - 1];
    if (p[i] == p[k]) ++k;
    pi[i] = (uint32_t)k;
  }
  c->bordered = pi[plen - 1] > 0;
  c->overlap_words.clear();
  for (uint32_t b = pi[plen - 1]; b > 0; b = pi[b - 1]) {
    std::vector<uint8_t> w(p, p + (plen - b));
    w.insert(w.end(), p, p + plen);
    if (c->overlap_words.size() == 3 || w.size() > XSG_MAX_PATTERN) {  // `aaaa`: such a needle overlaps itself wherever it is dense
      c->overlap_words.clear();
      break;
    }
    c->overlap_words.push_back(std::move(w));
  }

  // padded device copy (the long-pattern verify and the tail walk read it)
  // (at least a KiB: the long-pattern kernel stages min(plen, 1 KiB) into LDS, the tail kernels read a few bytes past short patterns)
  const size_t pat_bytes = std::max<size_t>(plen, 1024) + 16;
  XSG_TRY(c->d_pat.ensure(pat_bytes));
  std::vector<uint8_t> padded(pat_bytes, 0);
  memcpy(padded.data(), p, plen);
  HIP_TRY(hipMemcpyAsync(c->d_pat.p, padded.data(), padded.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));

  PatternDev& P = c->pat;
  P = PatternDev{};
  P.plen = (uint32_t)plen;
  window_fields(p, plen, pick_filter_window(p, plen), &P);  // koff 0 unless plen > 8
  c->koff_cands = window_candidates(p, plen);
  c->rx_pre = false;
  c->rx_fac = false;
  P.kind = plen < 4 ? kMask1 : plen == 4 ? kOne : plen < 8 ? kMask2 : plen == 8 ? kTwo : kLong;
  P.d_pat = c->d_pat.as<uint8_t>();
  P.exact_tail = (flags & XSG_FLAG_EXACT_TAIL) ? 1u : 0u;
  P.has_newline = memchr(p, '\n', plen) != nullptr;
  P.nl_first = p[0] == '\n' ? 1u : 0u;
  P.icase = (flags & XSG_FLAG_IGNORE_CASE) ? 1u : 0u;
  return XSG_OK;
}

// ---------------------------------------------------------------------------
// shards
// ---------------------------------------------------------------------------
// whatever was derived from the bytes of a binding is void once they change (re-bind, xsg_shard_invalidate)
static void forget_derived(xsg_shard* s) {
  s->nl_cached = s->nl_off_cached = false;  // newline counts per tile and their prefix
  s->hot_serial = 0;                         // measured hot filter / filter window of a pattern
  s->koff_chosen = false;
  s->pre_dense_serial = 0;                   // regex prefilter: candidates found dense
  s->mask_serial = s->mask_dense_serial = 0; // regex factor prefilter: tile marks
  s->fast_dense_serial = 0;                  // a pattern whose lists did not fit the one-sync route
  s->fast_result = false;
  s->line_len_on_device = false;
  s->density_serial = 0;                     // a pattern found dense in this data (scan_args: stagger)
  s->overlap_serial = 0;                     // a bordered pattern whose occurrences do not overlap in this data
}

static int bind_shard(xsg_shard* s, const void* d_base, uint64_t capacity, const xsg_chunk* chunks, uint64_t nchunks) {
  xsg_ctx* c = s->ctx;
  if (nchunks && !chunks) return fail(XSG_EINVAL, "chunks is null");
  if (nchunks && !d_base) return fail(XSG_EINVAL, "d_base is null");
  if (((uintptr_t)d_base & 15u) != 0) return fail(XSG_EINVAL, "d_base is not 16-byte aligned");
  if (nchunks >= (1ull << 32)) return fail(XSG_EINVAL, "too many chunks");
  uint64_t prev_end = 0, ntiles = 0, total = 0;
  const uint32_t tile_bytes = c->tile_bytes;
  std::vector<uint64_t> tile0(nchunks + 1, 0);
  for (uint64_t i = 0; i < nchunks; ++i) {
    const xsg_chunk& k = chunks[i];
    if (k.offset & 15u) return fail(XSG_EINVAL, "chunk %llu: offset %llu is not a multiple of 16", (unsigned long long)i,
                                    (unsigned long long)k.offset);
    if (k.length >= (1ull << 40)) return fail(XSG_EINVAL, "chunk %llu: length too large", (unsigned long long)i);
    const uint64_t rl = (k.length + 15u) & ~(uint64_t)15u;
    if (k.offset < prev_end) return fail(XSG_EINVAL, "chunk %llu overlaps its predecessor or is out of order",
                                         (unsigned long long)i);
    if (k.offset + rl > capacity || k.offset + rl < k.offset)
      return fail(XSG_EINVAL, "chunk %llu: offset+round_up(length,16) exceeds the shard capacity",
                  (unsigned long long)i);
    prev_end = k.offset + k.length;
    tile0[i] = ntiles;
    ntiles += (k.length + tile_bytes - 1) / tile_bytes;
    total += k.length;
  }
  tile0[nchunks] = ntiles;
  HIP_TRY(hipSetDevice(c->device));

  s->base = static_cast<const uint8_t*>(d_base);
  s->capacity = capacity;
  s->chunks.assign(chunks, chunks + nchunks);
  s->chunk_tile0 = std::move(tile0);
  s->ntiles = ntiles;
  s->tile_bytes = tile_bytes;
  s->total_bytes = total;
  s->last_mode = -1;
  s->total = 0;

  forget_derived(s);
  bool grew = false;
  XSG_TRY(s->d_chunks.ensure(sizeof(ChunkDev) * std::max<uint64_t>(nchunks, 1)));
  XSG_TRY(s->d_chunk_tile0.ensure(8 * (nchunks + 1)));
  XSG_TRY(s->d_tile_last.ensure(4 * std::max<uint64_t>(ntiles, 1), &grew));
  if (grew) s->last_valid = false;
  grew = false;
  XSG_TRY(s->d_tile_cnt.ensure(4 * std::max<uint64_t>(ntiles, 1), &grew));
  if (grew) s->cnt_clean = false;
  XSG_TRY(s->d_counters.ensure(8 * XSG_NUM_COUNTERS));
  grew = false;
  // k_count_finish scratch: the partial sums, then u32 words: [0] ticket, [1] scan flags
  XSG_TRY(s->d_finish.ensure(8 * 3 * (size_t)kFinishBlocks + 128, &grew));
  if (grew) HIP_TRY(hipMemsetAsync(s->d_finish.p, 0, 8 * 3 * (size_t)kFinishBlocks + 128, c->stream));  // tickets = 0
  if (!s->h_counters) HIP_TRY(hipHostMalloc((void**)&s->h_counters, 8 * XSG_NUM_COUNTERS, hipHostMallocDefault));
  if (!s->table_ev) HIP_TRY(hipEventCreateWithFlags(&s->table_ev, hipEventDisableTiming));
  static_assert(sizeof(ChunkDev) == sizeof(xsg_chunk), "layout");
  if (nchunks <= 1) {
    // The file pipeline re-binds its one-chunk shard for every chunk it feeds: the 48 bytes of table go through
    // a pinned staging block and no host sync.  The block is reused only after the previous upload has run.
    if (!s->h_stage) HIP_TRY(hipHostMalloc(&s->h_stage, 64, hipHostMallocDefault));
    if (s->table_pending) HIP_TRY(hipEventSynchronize(s->table_ev));
    uint8_t* st = static_cast<uint8_t*>(s->h_stage);
    if (nchunks) memcpy(st, s->chunks.data(), sizeof(xsg_chunk));
    memcpy(st + 32, s->chunk_tile0.data(), 8 * (nchunks + 1));
    if (nchunks) HIP_TRY(hipMemcpyAsync(s->d_chunks.p, st, sizeof(xsg_chunk), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(s->d_chunk_tile0.p, st + 32, 8 * (nchunks + 1), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(s->table_ev, c->stream));
    s->table_pending = true;
    return XSG_OK;
  }
  HIP_TRY(hipMemcpyAsync(s->d_chunks.p, s->chunks.data(), sizeof(xsg_chunk) * nchunks, hipMemcpyHostToDevice,
                         c->stream));
  HIP_TRY(hipMemcpyAsync(s->d_chunk_tile0.p, s->chunk_tile0.data(), 8 * (nchunks + 1), hipMemcpyHostToDevice,
                         c->stream));
  std::vector<uint32_t> map(ntiles);
  for (uint64_t i = 0; i < nchunks; ++i)
    for (uint64_t t = s->chunk_tile0[i]; t < s->chunk_tile0[i + 1]; ++t) map[t] = (uint32_t)i;
  XSG_TRY(s->d_tile_chunk.ensure(4 * std::max<uint64_t>(ntiles, 1)));
  if (ntiles) HIP_TRY(hipMemcpyAsync(s->d_tile_chunk.p, map.data(), 4 * ntiles, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // `map` is a local
  s->table_pending = false;
  return XSG_OK;
}

extern "C" int xsg_shard_invalidate(xsg_shard* s) {
  if (!s) return fail(XSG_EINVAL, "shard is null");
  if (s->ctx->memo.base == s->base) s->ctx->memo = xsg_ctx::ProbeMemo{};
  forget_derived(s);
  s->last_mode = -1;
  s->total = 0;
  return XSG_OK;
}

extern "C" int xsg_shard_create(xsg_ctx* c, const void* d_base, uint64_t capacity, const xsg_chunk* chunks,
                                uint64_t nchunks, xsg_shard** out) {
  if (!c) return fail(XSG_EINVAL, "ctx is null");
  if (!out) return fail(XSG_EINVAL, "out is null");
  *out = nullptr;
  xsg_shard* s = new (std::nothrow) xsg_shard();
  if (!s) return fail(XSG_ENOMEM, "host allocation failed");
  s->ctx = c;
  int r = bind_shard(s, d_base, capacity, chunks, nchunks);
  if (r != XSG_OK) {
    s->release_all();
    delete s;
    return r;
  }
  *out = s;
  return XSG_OK;
}

extern "C" int xsg_shard_rebind(xsg_shard* s, const void* d_base, uint64_t capacity, const xsg_chunk* chunks,
                                uint64_t nchunks) {
  if (!s) return fail(XSG_EINVAL, "shard is null");
  if (s->ctx->memo.base == s->base || s->ctx->memo.base == d_base) s->ctx->memo = xsg_ctx::ProbeMemo{};  // (the bytes changed)
  return bind_shard(s, d_base, capacity, chunks, nchunks);
}

extern "C" void xsg_shard_destroy(xsg_shard* s) {
  if (!s) return;
  (void)hipSetDevice(s->ctx->device);
  (void)hipStreamSynchronize(s->ctx->stream);
  s->release_all();
  delete s;
}

extern "C" int xsg_shard_set_line_base(xsg_shard* s, uint64_t line_base) {
  if (!s) return fail(XSG_EINVAL, "shard is null");
  s->shard_line_base = line_base;
  return XSG_OK;
}

// ---------------------------------------------------------------------------
// counting
// ---------------------------------------------------------------------------
static uint32_t scan_variant(bool want_nl, bool want_lines) { return (want_nl ? 1u : 0u) | (want_lines ? 2u : 0u); }

// `variant`: which k_scan instantiation the arguments are for (scan_variant): it selects the measured hot filter
static ScanArgs scan_args(xsg_shard* s, uint32_t variant = 0) {
  ScanArgs a{};
  a.base = s->base;
  a.chunks = s->d_chunks.as<ChunkDev>();
  a.tile_chunk = s->chunks.size() > 1 ? s->d_tile_chunk.as<uint32_t>() : nullptr;
  a.chunk_tile0 = s->d_chunk_tile0.as<uint64_t>();
  a.ntiles = s->ntiles;
  a.nchunks = s->chunks.size();
  a.tile_bytes = s->tile_bytes;
  // XSG_TUNE, else xsg_shard_tune's choice -- for the pattern it was measured with: a stagger tuned for a literal would
  // cost an instruction-bound class-sequence scan 10 % -- else per variant
  a.tune = s->ctx->tune != kTuneAuto ? s->ctx->tune
                                     : (s->tune_serial == s->ctx->pattern_serial || s->tune_serial == 0) ? s->tune : kTuneAuto;
  a.epoch = s->epoch;
  a.dense_hint = s->density_serial == s->ctx->pattern_serial ? s->dense : 0u;
  a.pat = s->ctx->pat;
  a.pat.hot = s->ctx->hot_env >= 0 ? (uint32_t)s->ctx->hot_env
              : (s->hot_serial == s->ctx->pattern_serial && ((s->hot_known >> variant) & 1u)) ? s->hot_v[variant] : 0u;
  if (a.pat.kind == kLong && s->hot_serial == s->ctx->pattern_serial && s->koff_chosen)
    window_fields(s->ctx->pattern.data(), s->ctx->pattern.size(), s->koff, &a.pat);  // the window measured best on this shard
  a.tile_cnt = s->d_tile_cnt.as<uint32_t>();
  a.tile_nl = s->d_tile_nl.as<uint32_t>();
  a.tile_sum = s->d_tile_sum.as<uint32_t>();
  a.tile_last = s->d_tile_last.as<uint32_t>();
  a.flags = reinterpret_cast<uint32_t*>(s->d_finish.as<uint64_t>() + 3 * (size_t)kFinishBlocks) + 1;  // behind the ticket
  a.tile_mask = (a.pat.kind == kDfa && s->mask_serial != 0 && s->mask_serial == s->ctx->pattern_serial)
                    ? s->d_tile_mask.as<uint32_t>() : nullptr;
  return a;
}

static int check_ready(xsg_shard* s) {
  if (!s) return fail(XSG_EINVAL, "shard is null");
  if (s->ctx->pattern.empty()) return fail(XSG_ESTATE, "xsg_set_pattern has not been called");
  return XSG_OK;
}

// k_scan only writes where a wave found something: "nothing found" must be in place before it runs.  After a
// count pass it already is (k_count_finish cleaned up behind itself): the steady state enqueues no memset at all.
// Also opens a new epoch for tile_last, and orders a pending chunk-table upload before work on a foreign stream.
static int prepare_tiles(xsg_shard* s, bool want_lines, hipStream_t st) {
  xsg_ctx* c = s->ctx;
  const uint64_t nt = std::max<uint64_t>(s->ntiles, 1);
  if (s->table_pending && st != c->stream) HIP_TRY(hipStreamWaitEvent(st, s->table_ev, 0));
  // A clean-up always covers the WHOLE allocation (the buffers grow geometrically): a later binding with more
  // tiles that still fits must find the words beyond today's ntiles clean as well.
  if (!s->cnt_clean) {
    HIP_TRY(hipMemsetAsync(s->d_tile_cnt.p, 0, s->d_tile_cnt.cap, st));
    s->cnt_clean = true;
  }
  if (!s->last_valid || s->epoch >= 0xffffu) {
    HIP_TRY(hipMemsetAsync(s->d_tile_last.p, 0, s->d_tile_last.cap, st));
    s->last_valid = true;
    s->epoch = 0;
  }
  ++s->epoch;
  if (want_lines) {
    bool grew = false;
    XSG_TRY(s->d_tile_sum.ensure(4 * kWaves * nt, &grew));
    if (grew) s->sum_clean = false;
    if (!s->sum_clean) {
      HIP_TRY(hipMemsetAsync(s->d_tile_sum.p, 0, s->d_tile_sum.cap, st));
      s->sum_clean = true;
    }
  }
  return XSG_OK;
}

static int ensure_tile_nl(xsg_shard* s) {
  bool grew = false;
  XSG_TRY(s->d_tile_nl.ensure(4 * std::max<uint64_t>(s->ntiles, 1), &grew));
  if (grew) s->nl_cached = s->nl_off_cached = false;
  return XSG_OK;
}

static bool is_window_kind(uint32_t k) { return k == kTwo || k == kLong || k == kClass; }

// The window kinds have two hot filters (k_scan<..., ALIGNED>): the aligned-dword trigger does half the VALU work
// but looks at 4 bytes of the window where the window filter looks at 8, so text in which the window's 4-byte
// pieces are common (a window made of words of the text) sends it into the slow path all the time.  Which one is
// faster is a property of (pattern, data, kernel variant): measured once per binding, pattern and VARIANT on a prefix
// of the shard (up to 2 GiB, a few launches of a fraction of a millisecond and one sync), remembered until the shard
// is re-bound or the pattern changes.  The probe times the variant the caller's pass is about to launch (round 3
// timed the newline-counting variant for every mode: the VALU-heaviest shows the largest difference -- but the answer
// differs: on the bench corpus the aligned trigger wins the count + newlines pass by 12 % and loses the plain count,
// which waits for memory with either filter, by 1.7 %).  Shards under 64 MiB keep the window filter (their scans
// take microseconds either way); XSG_HOT pins the choice.
static int choose_hot_filter(xsg_shard* s, hipStream_t st, bool want_nl = false, bool want_lines = false) {
  xsg_ctx* c = s->ctx;
  if (!is_window_kind(c->pat.kind) || c->hot_env >= 0) return XSG_OK;
  const uint32_t v = scan_variant(want_nl, want_lines);
  const bool first = s->hot_serial != c->pattern_serial;  // nothing measured for this pattern on this binding yet
  if (first) {
    s->hot_known = 0;
    memset(s->hot_v, 0, sizeof s->hot_v);
    s->koff_chosen = false;
    s->hot_serial = c->pattern_serial;
  }
  if ((s->hot_known >> v) & 1u) return XSG_OK;
  if (s->total_bytes < c->probe_min_bytes || s->ntiles == 0) {
    s->hot_known = 0xfu;  // too small to measure: the window filter for every variant
    return XSG_OK;
  }
  // Another binding of this buffer may have measured this pattern already (a caller that creates a shard per search).
  // The memo is keyed by address, size, chunk count and a content tag -- the first and last 16 bytes of the text: an
  // allocator that hands a freed address out again for OTHER data of the same size does not inherit the choice.
  bool memo_hit = c->memo.serial == c->pattern_serial && c->memo.base == s->base && c->memo.total_bytes == s->total_bytes &&
                  c->memo.nchunks == s->chunks.size();
  uint64_t tag[4] = {0, 0, 0, 0};
  {
    const xsg_chunk& c0 = s->chunks.front();
    const xsg_chunk& c1 = s->chunks.back();
    HIP_TRY(hipMemcpyAsync(tag, s->base + c0.offset, std::min<uint64_t>(16, c0.length), hipMemcpyDeviceToHost, st));
    if (c1.length >= 16) HIP_TRY(hipMemcpyAsync(tag + 2, s->base + c1.offset + c1.length - 16, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memo_hit = memo_hit && memcmp(tag, c->memo.tag, sizeof tag) == 0;
  }
  if (memo_hit && first) {  // measured on this buffer for this pattern by another binding
    s->koff = c->memo.koff;
    s->koff_chosen = c->memo.koff_chosen;
    if (c->memo.tune_probe && (s->tune_serial != c->pattern_serial || s->tune_probe)) {
      s->tune = c->memo.tune;
      s->tune_serial = s->tune == kTuneAuto ? 0 : c->pattern_serial;
      s->tune_probe = true;
    }
  }
  if (memo_hit && ((c->memo.hot_known >> v) & 1u)) {
    s->hot_v[v] = c->memo.hot_v[v];
    s->hot_known |= (uint8_t)(1u << v);
    return XSG_OK;
  }
  const bool settle_window = first && !memo_hit;  // the filter window and the stagger of a long pattern: once per pattern
  if (want_nl) XSG_TRY(ensure_tile_nl(s));
  XSG_TRY(prepare_tiles(s, want_lines, st));
  s->cnt_clean = false;  // no finish kernel behind these launches
  if (want_lines) s->sum_clean = false;
  hipEvent_t ev[2];
  for (hipEvent_t& e : ev) HIP_TRY(hipEventCreate(&e));
  float ms[2] = {1e30f, 1e30f};
  int rc = XSG_OK;
  // Long patterns first settle WHICH 8 bytes the hot loop looks for: how often a window occurs in this text decides
  // how often the slow path runs (`detective street` on the bench corpus: "ective s" every 4 KiB, "ve stree" every
  // 11 KiB), and no static letter table knows the text.  Each candidate: the window filter on a 1 GiB prefix.
  if (settle_window && c->pat.kind == kLong && c->koff_cands.size() > 1) {
    // (the candidates are timed on the newline-counting instantiation -- the VALU-heaviest, which shows a window's slow-path
    // rate best -- whatever variant the caller is about to launch: its per-tile newline array must exist.  Since the probe
    // became per-variant in round 4 a plain count no longer allocated it, and a long pattern's first count on a fresh binding
    // stored through a null pointer: tests/test_gpu_parity.py::test_first_search_of_a_fresh_binding_with_a_long_pattern)
    XSG_TRY(ensure_tile_nl(s));
    float best = 1e30f;
    uint32_t best_koff = c->koff_cands[0];
    for (uint32_t koff : c->koff_cands) {
      ScanArgs a = scan_args(s);
      window_fields(c->pattern.data(), c->pattern.size(), koff, &a.pat);
      a.pat.hot = 0;
      a.tune = 0;
      a.ntiles = std::min<uint64_t>(a.ntiles, 65536);
      hipError_t e = launch_scan_count(a, true, false, st);  // warm-up, then one timed launch
      if (e == hipSuccess) e = hipEventRecord(ev[0], st);
      if (e == hipSuccess) e = launch_scan_count(a, true, false, st);
      if (e == hipSuccess) e = hipEventRecord(ev[1], st);
      if (e == hipSuccess) e = hipEventSynchronize(ev[1]);
      float t = 0;
      if (e == hipSuccess) e = hipEventElapsedTime(&t, ev[0], ev[1]);
      if (e != hipSuccess) {
        rc = fail(XSG_EHIP, "window probe failed: %s", hipGetErrorString(e));
        break;
      }
      if (t < 0.97f * best) {  // the first candidate is the heuristic's pick; a later one must beat the best clearly
        best = t;
        best_koff = koff;
      }
    }
    if (rc == XSG_OK) {
      s->koff = best_koff;
      s->koff_chosen = true;
    }
  }
  // A B A B A B, the best of three rounds each: the first launches after a quiet spell run on ramping clocks (measured
  // on fresh bindings of one 50 GiB shard, scripts/probe_check.py: `Sherlock` 0.658 against 0.600 ms in seven probes
  // of eight, 0.664 against 0.633 in the first -- two rounds and a 3 % bar picked the slower filter now and then)
  for (int round = 0; round < 3 && rc == XSG_OK; ++round) {
    for (uint32_t hot = 0; hot < 2 && rc == XSG_OK; ++hot) {
      ScanArgs a = scan_args(s);
      a.pat.hot = hot;
      // (with the wave stagger the variant's real launches use: round 3 timed with the stagger off, which is how the
      // newline-counting variant runs anyway -- but count_lines runs with 16, and there the aligned trigger wins by 3.7 %
      // where it ties with the stagger off: profiles/r04_dense_variants.txt)
      a.ntiles = std::min<uint64_t>(a.ntiles, 131072);
      // one timed launch per round (0.3 ms on the 2 GiB prefix); the first round warms up first (that pulls the code in)
      hipError_t e = round == 0 ? launch_scan_count(a, want_nl, want_lines, st) : hipSuccess;
      if (e == hipSuccess) e = hipEventRecord(ev[0], st);
      if (e == hipSuccess) e = launch_scan_count(a, want_nl, want_lines, st);
      if (e == hipSuccess) e = hipEventRecord(ev[1], st);
      if (e == hipSuccess) e = hipEventSynchronize(ev[1]);
      float t = 0;
      if (e == hipSuccess) e = hipEventElapsedTime(&t, ev[0], ev[1]);
      if (e != hipSuccess) rc = fail(XSG_EHIP, "hot-filter probe failed: %s", hipGetErrorString(e));
      ms[hot] = std::min(ms[hot], t);
    }
  }
  // The plain count waits for memory and the window filter is its better half at every size measured with warm clocks
  // (10 / 20 / 50 GiB: 0.925 / 0.933 / 0.938 of peak against 0.89-0.90): the aligned trigger has to win visibly.  The
  // variants that also count newlines or keep line summaries are VALU-bound, the trigger is half the filter work and wins
  // by 3-6 % at full size (count_lines 0.92-0.94 against 0.89; with newline counts 0.89 against 0.84) -- but on the 2 GiB
  // prefix the launch ramp dilutes that to ~1 %, inside the probe's noise (a 10 GiB shard: 0.2962 against 0.2993 ms, and a
  // 1.5 % bar kept the window filter): there the WINDOW filter has to win by 1.5 %, as it does when the trigger's pieces
  // are common in the text (profiles/r04_dense_variants.txt).
  s->hot_v[v] = (v == 0 ? ms[1] < 0.995f * ms[0] : ms[1] < 1.015f * ms[0]) ? 1u : 0u;
  s->hot_known |= (uint8_t)(1u << v);
  static const bool probe_log = getenv("XSG_PROBE_LOG") != nullptr;
  if (probe_log)
    fprintf(stderr, "[xsg] hot-filter probe (variant nl=%d lines=%d): window %.4f ms, aligned %.4f ms -> %u (koff %u)\n", (int)want_nl,
            (int)want_lines, ms[0], ms[1], (unsigned)s->hot_v[v], s->koff_chosen ? s->koff : 0u);
  // Long patterns also settle their wave stagger here: the default (16) is right for a scan that waits for memory and
  // costs one that waits for its slow path -- which of the two a long pattern is depends on how often its window occurs
  // in THIS text (`detective street` on the bench corpus: 5.4 TB/s with the default, 6.1 without; `Sherlock Holmes` the
  // other way round).  The plain count with the window and filter just chosen, stagger 0 against the default, on the
  // same prefix; a tie keeps the default.  xsg_shard_tune (all staggers, full size) overrides it.
  if (rc == XSG_OK && settle_window && c->pat.kind == kLong && c->tune == kTuneAuto && (s->tune_serial != c->pattern_serial || s->tune_probe)) {
    float tms[2] = {1e30f, 1e30f};
    static const uint32_t cand[2] = {kDefaultStagger, 0u};
    for (int round = 0; round < 3 && rc == XSG_OK; ++round) {
      for (int k = 0; k < 2 && rc == XSG_OK; ++k) {
        ScanArgs a = scan_args(s);
        a.pat.hot = s->hot_v[v];
        a.tune = cand[k];
        a.ntiles = std::min<uint64_t>(a.ntiles, 131072);
        hipError_t e = round == 0 ? launch_scan_count(a, false, false, st) : hipSuccess;
        if (e == hipSuccess) e = hipEventRecord(ev[0], st);
        if (e == hipSuccess) e = launch_scan_count(a, false, false, st);
        if (e == hipSuccess) e = hipEventRecord(ev[1], st);
        if (e == hipSuccess) e = hipEventSynchronize(ev[1]);
        float t = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, ev[0], ev[1]);
        if (e != hipSuccess) rc = fail(XSG_EHIP, "stagger probe failed: %s", hipGetErrorString(e));
        tms[k] = std::min(tms[k], t);
      }
    }
    if (rc == XSG_OK) {
      s->tune = tms[1] < 0.97f * tms[0] ? 0u : kTuneAuto;
      s->tune_serial = s->tune == kTuneAuto ? 0 : c->pattern_serial;
      s->tune_probe = true;
    }
  }
  for (hipEvent_t& e : ev) (void)hipEventDestroy(e);
  if (rc != XSG_OK) return rc;
  HIP_TRY(hipMemsetAsync(scan_args(s).flags, 0, 4, st));
  if (!memo_hit) {
    c->memo = xsg_ctx::ProbeMemo{};
    c->memo.serial = c->pattern_serial;
    c->memo.base = s->base;
    c->memo.total_bytes = s->total_bytes;
    c->memo.nchunks = s->chunks.size();
    memcpy(c->memo.tag, tag, sizeof tag);
  }
  c->memo.hot_v[v] = s->hot_v[v];
  c->memo.hot_known |= (uint8_t)(1u << v);
  c->memo.koff = s->koff, c->memo.koff_chosen = s->koff_chosen;
  c->memo.tune = s->tune, c->memo.tune_probe = s->tune_probe;
  return XSG_OK;
}

static int enqueue_count(xsg_shard* s, bool want_matches, bool want_lines, bool want_nl, hipStream_t st,
                         uint64_t* d_counters, uint64_t* host_counters, const PatternDev* other_pattern = nullptr,
                         uint64_t* d_status = nullptr) {
  if (want_nl) XSG_TRY(ensure_tile_nl(s));
  const uint64_t nchunks = s->chunks.size();
  // kDfa: k_rx_scan counts matching lines directly into tile_cnt (a line is one lane's work): no line summaries
  const bool rx_lines = want_lines && s->ctx->pat.kind == kDfa;
  if (rx_lines) want_lines = false;
  const bool scan_nl = want_nl && !s->nl_cached;  // the per-tile newline counts of this binding may already exist
  if (!other_pattern) XSG_TRY(choose_hot_filter(s, st, scan_nl, want_lines));  // measured for the variant this pass launches
  XSG_TRY(prepare_tiles(s, want_lines, st));
  ScanArgs a = scan_args(s, scan_variant(scan_nl, want_lines));
  if (other_pattern) {  // (ensure_overlap_check: a word derived from the ctx's pattern, nothing measured or remembered for it)
    a.pat = *other_pattern;
    a.dense_hint = 0;
    if (s->tune_serial != 0) a.tune = kTuneAuto;
  }
  // dirty until the finish kernel is in the queue behind the scan
  s->cnt_clean = false;
  if (want_lines) s->sum_clean = false;
  a.lines_only = want_lines && !want_matches;
  HIP_TRY(launch_scan_count(a, scan_nl, want_lines || rx_lines, st));
  FinishArgs f{};
  f.base = s->base;
  f.chunks = a.chunks;
  f.chunk_tile0 = a.chunk_tile0;
  f.nchunks = nchunks;
  f.ntiles = s->ntiles;
  f.pat = a.pat;
  f.tile_cnt = a.tile_cnt;
  f.tile_nl = a.tile_nl;
  f.tile_sum = a.tile_sum;
  f.tile_last = a.tile_last;
  f.epoch = a.epoch;
  f.tile_bytes = s->tile_bytes;
  f.counters = d_counters;
  f.host_counters = host_counters;
  f.status = d_status;
  f.partials = s->d_finish.as<uint64_t>();
  f.ticket = reinterpret_cast<uint32_t*>(s->d_finish.as<uint64_t>() + 3 * (size_t)kFinishBlocks);
  f.flags = a.flags;
  f.total_bytes = s->total_bytes;
  f.want_nl = want_nl;
  f.want_lines = want_lines;
  f.want_matches = want_matches || rx_lines;
  f.cnt_is_lines = rx_lines ? 1u : 0u;
  HIP_TRY(launch_count_finish(f, st));
  s->cnt_clean = true;
  if (want_lines) s->sum_clean = true;
  if (scan_nl) s->nl_cached = true;
  return XSG_OK;
}

// A literal pattern with a border CAN overlap itself; whether it DOES in the bound data is a property of that data, and in
// text it nearly never does (`that`: "thathat" would have to occur).  Two occurrences plen - b apart spell the word
// P[0 .. plen - b) + P, one word per border b: a count pass for each (exact to the end of the chunk, so pairs reaching into
// the end-of-chunk zone are seen too) settles it once per (binding, pattern).  No such word -> every occurrence is kept by
// the reference's walk, and the pattern is counted and listed like one without a border: one pass and no list where
// xs::count took the whole list route (`that` on 10 GiB: 5.3 ms -> one pass).
static bool overlap_free_known(const xsg_shard* s) {
  return s->overlap_serial == s->ctx->pattern_serial && s->overlap_free;
}
static int ensure_overlap_check(xsg_shard* s) {
  xsg_ctx* c = s->ctx;
  if (s->overlap_serial == c->pattern_serial) return XSG_OK;
  s->overlap_serial = c->pattern_serial;
  s->overlap_free = false;
  if (!c->bordered || c->overlap_words.empty() || s->ntiles == 0) return XSG_OK;
  hipStream_t st = c->stream;
  // the words go to the device once per PATTERN (every binding of the file pipeline asks again), side by side
  size_t stride = 0;
  for (const std::vector<uint8_t>& w : c->overlap_words) stride = std::max(stride, (std::max<size_t>(w.size(), 1024) + 16 + 255) & ~(size_t)255);
  if (c->aux_serial != c->pattern_serial) {
    std::vector<uint8_t> all(stride * c->overlap_words.size(), 0);
    for (size_t k = 0; k < c->overlap_words.size(); ++k) memcpy(all.data() + k * stride, c->overlap_words[k].data(), c->overlap_words[k].size());
    XSG_TRY(c->d_aux_pat.ensure(all.size()));
    HIP_TRY(hipMemcpyAsync(c->d_aux_pat.p, all.data(), all.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // (`all` leaves scope)
    c->aux_serial = c->pattern_serial;
  }
  for (size_t k = 0; k < c->overlap_words.size(); ++k) {
    const std::vector<uint8_t>& w = c->overlap_words[k];
    PatternDev P{};
    P.plen = (uint32_t)w.size();
    window_fields(w.data(), w.size(), pick_filter_window(w.data(), w.size()), &P);
    P.kind = w.size() < 4 ? kMask1 : w.size() == 4 ? kOne : w.size() < 8 ? kMask2 : w.size() == 8 ? kTwo : kLong;
    P.d_pat = c->d_aux_pat.as<uint8_t>() + k * stride;
    P.exact_tail = 1u;
    P.has_newline = c->pat.has_newline;
    P.icase = c->pat.icase;
    XSG_TRY(enqueue_count(s, true, false, false, st, s->d_counters.as<uint64_t>(), s->h_counters, &P));
    HIP_TRY(hipStreamSynchronize(st));
    s->table_pending = false;
    if (s->h_counters[XSG_CTR_MATCHES] != 0) return XSG_OK;  // they do overlap here: the list route decides which are kept
  }
  s->overlap_free = true;
  return XSG_OK;
}

// XSG_COUNT_MATCHES for a pattern that can overlap itself, without a trip to the host: the list route of xsg_count
// (count pass, ranks, ordered emission, greedy keep, end-of-chunk walk) with the one number it used to fetch --
// how many raw occurrences there are -- left on the device: the arrays get a CAPACITY (twice what the last such
// pass on this shard found, at least a million), emission is bounded by it, the list kernels read the count
// from tile_off[ntiles], and a final kernel adds up keep[] and the tail counts.  More occurrences than capacity
// -> all four counters UINT64_MAX (like the ascii_only refusal): the caller takes xsg_count, which also teaches
// the shard the size for next time.
static int enqueue_count_bordered(xsg_shard* s, hipStream_t st, uint64_t* d_counters, uint64_t* d_status) {
  xsg_ctx* c = s->ctx;
  const uint64_t nchunks = s->chunks.size();
  const uint64_t ntiles = s->ntiles;
  XSG_TRY(choose_hot_filter(s, st));
  XSG_TRY(prepare_tiles(s, false, st));
  ScanArgs a = scan_args(s);
  s->cnt_clean = false;  // the tile counts stay for the emit pass
  HIP_TRY(launch_scan_count(a, false, false, st));
  XSG_TRY(s->d_tile_off.ensure(8 * (ntiles + 1)));
  const uint64_t cap = std::max<uint64_t>(1u << 20, 2 * s->last_raw_matches);
  XSG_TRY(s->d_scan_tmp.ensure(8 * scan_tmp_elems(std::max<uint64_t>(ntiles, nchunks) + 1)));
  HIP_TRY(launch_exclusive_scan_u32(a.tile_cnt, s->d_tile_off.as<uint64_t>(), ntiles, s->d_scan_tmp.as<uint64_t>(), st));
  XSG_TRY(s->d_m_pos.ensure(8 * cap));
  XSG_TRY(s->d_m_chunk.ensure(4 * cap));
  XSG_TRY(s->d_keep.ensure(4 * cap));
  const uint32_t tail_cap = std::max<uint32_t>(tail_max_matches(c->pat.plen), 1u);
  XSG_TRY(s->d_chunk_shift0.ensure(8 * std::max<uint64_t>(nchunks, 1)));
  XSG_TRY(s->d_tail_cnt.ensure(4 * std::max<uint64_t>(nchunks, 1)));
  XSG_TRY(s->d_tail_pos.ensure(8 * std::max<uint64_t>(nchunks, 1) * tail_cap));
  a.tile_off = s->d_tile_off.as<uint64_t>();
  a.m_pos = s->d_m_pos.as<uint64_t>();
  a.m_chunk = s->d_m_chunk.as<uint32_t>();
  a.m_cap = cap;
  HIP_TRY(launch_scan_emit(a, st));
  ListArgs l{};
  l.base = s->base;
  l.chunks = a.chunks;
  l.chunk_tile0 = a.chunk_tile0;
  l.nchunks = nchunks;
  l.pat = a.pat;
  l.M = cap;
  l.M_dev = a.tile_off + ntiles;
  l.m_pos = a.m_pos;
  l.m_chunk = a.m_chunk;
  l.tile_off = a.tile_off;
  l.keep = s->d_keep.as<uint32_t>();
  l.chunk_shift0 = s->d_chunk_shift0.as<uint64_t>();
  l.tail_cnt = s->d_tail_cnt.as<uint32_t>();
  l.tail_pos = s->d_tail_pos.as<uint64_t>();
  l.tail_cap = tail_cap;
  l.line_mode = 0;
  HIP_TRY(hipMemsetAsync(l.keep, 0, 4 * cap, st));  // entries that are not chain heads or members are written; be safe
  HIP_TRY(launch_greedy_keep(l, st));
  HIP_TRY(launch_chunk_shift0(l, st));
  HIP_TRY(launch_tail_list(l, st));
  HIP_TRY(hipMemsetAsync(d_counters, 0, 8 * XSG_NUM_COUNTERS, st));
  HIP_TRY(launch_bordered_total(l, d_counters, s->total_bytes, a.flags, d_status, st));
  s->last_mode = -1;
  return XSG_OK;
}

static const char* const kNonAsciiMsg =
    "the expression uses '.', a negated class or \\D \\W \\S, which match whole code points in RE2; the data holds "
    "bytes >= 0x80, where one byte per position is not the same thing: refused, not approximated";

static int refuse_if_poisoned(const uint64_t counters[XSG_NUM_COUNTERS]) {
  if (counters[XSG_CTR_BYTES] == UINT64_MAX) return fail(XSG_ENOTSUP, "%s", kNonAsciiMsg);
  return XSG_OK;
}

static int count_async_impl(xsg_shard* s, uint32_t mode, void* stream, uint64_t* d_counters, uint64_t* d_status) {
  XSG_TRY(check_ready(s));
  if (!d_counters) return fail(XSG_EINVAL, "d_counters is null");
  const uint32_t m = mode & 0xffu;
  const bool want_nl = (mode & XSG_WITH_NEWLINES) != 0;
  if (mode & ~(0xffu | XSG_WITH_NEWLINES)) return fail(XSG_EINVAL, "unknown mode bits 0x%x", mode);
  xsg_ctx* c = s->ctx;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
  if (m == XSG_COUNT_MATCHES) {
    if (c->bordered && !overlap_free_known(s)) {  // (known from an earlier synchronous call: this entry point may not wait)
      if (want_nl)
        return fail(XSG_ENOTSUP, "pattern can overlap itself: XSG_WITH_NEWLINES next to its match count needs xsg_count()");
      return enqueue_count_bordered(s, st, d_counters, d_status);
    }
    return enqueue_count(s, true, false, want_nl, st, d_counters, nullptr, nullptr, d_status);
  }
  if (m == XSG_COUNT_LINES) {
    if (c->pat.has_newline)
      return fail(XSG_ENOTSUP, "count_lines of a pattern that contains '\\n' walks a chain of occurrences: the stream-ordered entry "
                               "point does not serve it, xsg_count() does");
    return enqueue_count(s, false, true, want_nl, st, d_counters, nullptr, nullptr, d_status);
  }
  return fail(XSG_EINVAL, "xsg_count_async: mode %u is not a count mode", m);
}

extern "C" int xsg_count_async(xsg_shard* s, uint32_t mode, void* stream, uint64_t* d_counters) {
  return count_async_impl(s, mode, stream, d_counters, nullptr);
}

extern "C" int xsg_count_async_status(xsg_shard* s, uint32_t mode, void* stream, uint64_t* d_counters, uint64_t* d_status) {
  if (!d_status) return fail(XSG_EINVAL, "d_status is null");
  return count_async_impl(s, mode, stream, d_counters, d_status);
}

static int run_list(xsg_shard* s, uint32_t mode, bool outputs);
// A LITERAL that contains '\n': its line tags are a chain of occurrences (k_nlpat_links), resolved on the exact list
// route.  (An EXPRESSION that can match '\n' keeps being refused by the line tags: RE2's walk over a re-sliced input is
// not restated for it.)
static bool newline_literal(const xsg_ctx* c) { return c->pat.has_newline && c->pat.kind != kClass && c->pat.kind != kDfa; }
static const char* const kNewlineExprMsg = "line modes do not accept an expression that can match '\\n'";
// The prefilter route is half a dozen kernels and three trips to the host where k_rx_scan is one pass: it pays on
// shards where a pass takes longer than that (the file pipeline's 16 MiB chunks are walked by k_rx_scan in tens of
// microseconds).
// The factor prefilter of the automaton route: for an expression without a selective start but with a class sequence
// every match contains, the occurrences of that factor are found by the scan kernel's class-sequence matcher (count +
// emit), the first occurrence of every line gives the line's start (k_line_starts_keep, as for any line tag), and the
// tiles in which such lines start are marked.  k_rx_scan then leaves every other tile at once.  Built once per
// (binding, pattern) by the first synchronous call and used by every later pass, the stream-ordered ones included;
// not built where the factor turns out dense (most tiles would be marked) or the shard is small.
static int d2h_u64(xsg_ctx* c, const uint64_t* d, uint64_t* h);
static int ensure_factor_mask(xsg_shard* s) {
  xsg_ctx* c = s->ctx;
  if (c->pat.kind != kDfa || !c->rx_fac || s->ntiles == 0) return XSG_OK;
  if (s->mask_serial == c->pattern_serial || s->mask_dense_serial == c->pattern_serial) return XSG_OK;
  if (!c->rx_fac_forced && s->total_bytes < (512ull << 20)) return XSG_OK;
  hipStream_t st = c->stream;
  const uint64_t nchunks = s->chunks.size(), ntiles = s->ntiles;
  XSG_TRY(prepare_tiles(s, false, st));
  ScanArgs a = scan_args(s);
  a.pat = c->fac_pat;
  a.pat.hot = 0;
  a.tile_mask = nullptr;
  s->cnt_clean = false;
  HIP_TRY(launch_scan_count(a, false, false, st));
  XSG_TRY(s->d_tile_off.ensure(8 * (ntiles + 1)));
  XSG_TRY(s->d_scan_tmp.ensure(8 * scan_tmp_elems(std::max<uint64_t>(ntiles, nchunks) + 1)));
  HIP_TRY(launch_exclusive_scan_u32(a.tile_cnt, s->d_tile_off.as<uint64_t>(), ntiles, s->d_scan_tmp.as<uint64_t>(), st));
  uint64_t M = 0;
  uint32_t flags = 0;
  HIP_TRY(hipMemcpyAsync(&flags, a.flags, 4, hipMemcpyDeviceToHost, st));
  XSG_TRY(d2h_u64(c, s->d_tile_off.as<uint64_t>() + ntiles, &M));
  if (flags & 1u) {  // non-ASCII data under an ascii_only expression: the search itself will refuse
    HIP_TRY(hipMemsetAsync(a.flags, 0, 4, st));
    return fail(XSG_ENOTSUP, "%s", kNonAsciiMsg);
  }
  if (!c->rx_fac_forced && M * 256 > s->total_bytes) {  // most tiles would be marked
    s->mask_dense_serial = c->pattern_serial;
    return XSG_OK;
  }
  XSG_TRY(s->d_c_pos.ensure(8 * std::max<uint64_t>(M, 1)));
  XSG_TRY(s->d_c_chunk.ensure(4 * std::max<uint64_t>(M, 1)));
  XSG_TRY(s->d_c_keep.ensure(4 * std::max<uint64_t>(M, 1)));
  XSG_TRY(s->d_m_ls.ensure(8 * std::max<uint64_t>(M, 1)));
  XSG_TRY(s->d_tile_mask.ensure(4 * ntiles));
  a.tile_off = s->d_tile_off.as<uint64_t>();
  a.m_pos = s->d_c_pos.as<uint64_t>();
  a.m_chunk = s->d_c_chunk.as<uint32_t>();
  if (M) HIP_TRY(launch_scan_emit(a, st));
  ListArgs l{};
  l.base = s->base;
  l.chunks = a.chunks;
  l.chunk_tile0 = a.chunk_tile0;
  l.nchunks = nchunks;
  l.pat = a.pat;
  l.M = M;
  l.m_pos = a.m_pos;
  l.m_chunk = a.m_chunk;
  l.tile_off = a.tile_off;
  l.m_ls = s->d_m_ls.as<uint64_t>();
  l.keep = s->d_c_keep.as<uint32_t>();
  l.line_mode = 1;
  HIP_TRY(launch_line_starts_keep(l, st));
  HIP_TRY(hipMemsetAsync(s->d_tile_mask.p, 0, 4 * ntiles, st));
  HIP_TRY(launch_rx_mark_tiles(l, s->d_tile_mask.as<uint32_t>(), st));
  s->mask_serial = c->pattern_serial;
  return XSG_OK;
}

static bool use_prefilter(const xsg_shard* s) {
  const xsg_ctx* c = s->ctx;
  return c->pat.kind == kDfa && c->rx_pre && !s->pre_off && (c->rx_pre_forced || s->total_bytes >= (512ull << 20)) &&
         s->chunks.size() < (1u << 24);  // k_rx_heads' keys: chunk number above 40 bits of offset
}
constexpr int kDenseCandidates = 1;  // run_list(outputs = false) on the prefilter route: too many candidates, count by k_rx_scan

// What the next pass of this pattern over this data can know: a needle found at least once per 2 KiB keeps the slow
// path of the 4..8-byte kinds busy in two wave-loads of five, and the kernel waits for its ALUs, not for memory -- the
// wave stagger that pays for a sparse needle (16) costs such a scan 7 % (`that`: 8.25 ms against 7.66 at 4, 50 GiB).
static void note_density(xsg_shard* s, uint64_t results) {
  if (results == UINT64_MAX) return;
  s->density_serial = s->ctx->pattern_serial;
  // 2: the wave stagger that pays for a sparse needle costs such a scan (pick_stagger); 1: already at one result per 8 KiB a
  // 4..8-byte needle sends a quarter of its wave-loads into the slow path and is cheaper decided byte-parallel
  // (dense_bytes_route; natural text, 10 GiB: `return`, one per 4 KiB, 0.72 of peak on the hot filter -- profiles/r04_natural_variants.txt)
  uint64_t per = 8192u;
  if (const char* e = XSG_TOGGLE("XSG_DENSE_PER")) per = std::max<uint64_t>(strtoull(e, nullptr, 10), 1u);  // A/B: scripts/natural_density.py
  s->dense = results > s->total_bytes / 2048u ? 2u : results > s->total_bytes / per ? 1u : 0u;
}

extern "C" int xsg_count(xsg_shard* s, uint32_t mode, uint64_t counters[XSG_NUM_COUNTERS]) {
  XSG_TRY(check_ready(s));
  if (!counters) return fail(XSG_EINVAL, "counters is null");
  const uint32_t m = mode & 0xffu;
  xsg_ctx* c = s->ctx;
  HIP_TRY(hipSetDevice(c->device));
  if (m != XSG_COUNT_MATCHES && m != XSG_COUNT_LINES) return fail(XSG_EINVAL, "mode %u is not a count mode", m);
  XSG_TRY(ensure_factor_mask(s));
  if (use_prefilter(s) && s->pre_dense_serial != c->pattern_serial) {  // (not again where the candidates were found dense)
    // the prefilter route of the automaton family: candidates, verification and the walk produce the list; its
    // length is the count (the newline total, if asked for, comes from the cached per-tile counts)
    if (mode & ~(0xffu | XSG_WITH_NEWLINES)) return fail(XSG_EINVAL, "unknown mode bits 0x%x", mode);
    if (m == XSG_COUNT_LINES && c->pat.has_newline) return fail(XSG_ENOTSUP, "%s", kNewlineExprMsg);
    s->want_nl_total = (mode & XSG_WITH_NEWLINES) != 0;
    const int r = run_list(s, m == XSG_COUNT_MATCHES ? XSG_MATCH_BYTE_OFFSETS : XSG_LINE_BYTE_OFFSETS, false);
    s->want_nl_total = false;
    if (r != kDenseCandidates) {
      XSG_TRY(r);
      memset(counters, 0, 8 * XSG_NUM_COUNTERS);
      counters[m == XSG_COUNT_MATCHES ? XSG_CTR_MATCHES : XSG_CTR_LINES] = s->total;
      counters[XSG_CTR_BYTES] = s->total_bytes;
      if (mode & XSG_WITH_NEWLINES) counters[XSG_CTR_NEWLINES] = s->last_newlines;
      s->last_mode = -1;
      return XSG_OK;
    }
    // too many candidates for the list route to pay: the count passes below walk every line (k_rx_scan)
  }
  if (m == XSG_COUNT_MATCHES && c->bordered) XSG_TRY(ensure_overlap_check(s));
  const bool chain_lines = m == XSG_COUNT_LINES && newline_literal(c);  // the line walk of a literal with '\n' in it
  if ((m == XSG_COUNT_MATCHES && c->bordered && !overlap_free_known(s)) || chain_lines) {
    // greedy non-overlap (and the line walk of a pattern that holds a newline) needs the ordered occurrence list
    if (mode & ~(0xffu | XSG_WITH_NEWLINES)) return fail(XSG_EINVAL, "unknown mode bits 0x%x", mode);
    XSG_TRY(run_list(s, chain_lines ? XSG_LINE_BYTE_OFFSETS : XSG_MATCH_BYTE_OFFSETS, false));
    note_density(s, s->total);
    memset(counters, 0, 8 * XSG_NUM_COUNTERS);
    counters[chain_lines ? XSG_CTR_LINES : XSG_CTR_MATCHES] = s->total;
    counters[XSG_CTR_BYTES] = s->total_bytes;
    if (mode & XSG_WITH_NEWLINES) {
      XSG_TRY(enqueue_count(s, false, false, true, c->stream, s->d_counters.as<uint64_t>(), s->h_counters));
      HIP_TRY(hipStreamSynchronize(c->stream));
      s->table_pending = false;
      counters[XSG_CTR_NEWLINES] = s->h_counters[XSG_CTR_NEWLINES];
    }
    s->last_mode = -1;
    return XSG_OK;
  }
  const bool want_nl = (mode & XSG_WITH_NEWLINES) != 0;
  if (mode & ~(0xffu | XSG_WITH_NEWLINES)) return fail(XSG_EINVAL, "unknown mode bits 0x%x", mode);
  if (m == XSG_COUNT_LINES && c->pat.has_newline) return fail(XSG_ENOTSUP, "%s", kNewlineExprMsg);
  // the finish kernel writes the four values straight into pinned host memory: no copy, one sync
  XSG_TRY(enqueue_count(s, m == XSG_COUNT_MATCHES, m == XSG_COUNT_LINES, want_nl, c->stream,
                        s->d_counters.as<uint64_t>(), s->h_counters));
  HIP_TRY(hipStreamSynchronize(c->stream));
  s->table_pending = false;
  memcpy(counters, s->h_counters, 8 * XSG_NUM_COUNTERS);
  note_density(s, counters[m == XSG_COUNT_MATCHES ? XSG_CTR_MATCHES : XSG_CTR_LINES]);
  return refuse_if_poisoned(counters);
}

// Split-phase xsg_count for host pipelines: begin enqueues the pass on the ctx stream (results go to the shard's
// pinned mirror), end waits for it.  Between the two the caller may enqueue work for other shards/contexts.
extern "C" int xsg_count_begin(xsg_shard* s, uint32_t mode) {
  XSG_TRY(check_ready(s));
  const uint32_t m = mode & 0xffu;
  xsg_ctx* c = s->ctx;
  HIP_TRY(hipSetDevice(c->device));
  if (m != XSG_COUNT_MATCHES && m != XSG_COUNT_LINES) return fail(XSG_EINVAL, "mode %u is not a count mode", m);
  if (mode & ~(0xffu | XSG_WITH_NEWLINES)) return fail(XSG_EINVAL, "unknown mode bits 0x%x", mode);
  s->begin_sync_result = false;
  XSG_TRY(ensure_factor_mask(s));
  if (m == XSG_COUNT_MATCHES && c->bordered) XSG_TRY(ensure_overlap_check(s));
  if ((m == XSG_COUNT_MATCHES && c->bordered && !overlap_free_known(s)) || (m == XSG_COUNT_LINES && newline_literal(c)) ||
      (use_prefilter(s) && s->pre_dense_serial != c->pattern_serial)) {  // needs the ordered list: done synchronously, handed out by _end
    XSG_TRY(xsg_count(s, mode, s->begin_counters));
    s->begin_sync_result = true;
    return XSG_OK;
  }
  if (m == XSG_COUNT_LINES && c->pat.has_newline) return fail(XSG_ENOTSUP, "%s", kNewlineExprMsg);
  XSG_TRY(enqueue_count(s, m == XSG_COUNT_MATCHES, m == XSG_COUNT_LINES, (mode & XSG_WITH_NEWLINES) != 0, c->stream,
                        s->d_counters.as<uint64_t>(), s->h_counters));
  HIP_TRY(hipEventRecord(s->table_ev, c->stream));  // doubles as "pass done": it covers the table upload too
  s->table_pending = true;
  return XSG_OK;
}

extern "C" int xsg_count_end(xsg_shard* s, uint64_t counters[XSG_NUM_COUNTERS]) {
  if (!s || !counters) return fail(XSG_EINVAL, "null argument");
  if (s->begin_sync_result) {
    memcpy(counters, s->begin_counters, 8 * XSG_NUM_COUNTERS);
    s->begin_sync_result = false;
    return XSG_OK;
  }
  HIP_TRY(hipSetDevice(s->ctx->device));
  HIP_TRY(hipEventSynchronize(s->table_ev));
  s->table_pending = false;
  memcpy(counters, s->h_counters, 8 * XSG_NUM_COUNTERS);
  return refuse_if_poisoned(counters);
}

extern "C" int xsg_time_scan_kernel(xsg_shard* s, uint32_t mode, int iters, float* avg_ms) {
  XSG_TRY(check_ready(s));
  if (!avg_ms || iters <= 0) return fail(XSG_EINVAL, "bad iters/avg_ms");
  xsg_ctx* c = s->ctx;
  HIP_TRY(hipSetDevice(c->device));
  const uint32_t m = mode & 0xffu;
  const bool want_nl = (mode & XSG_WITH_NEWLINES) != 0;
  const bool want_lines = m == XSG_COUNT_LINES;
  if (want_nl) XSG_TRY(ensure_tile_nl(s));
  XSG_TRY(choose_hot_filter(s, c->stream, want_nl, want_lines));  // time what a real pass of this mode would launch
  XSG_TRY(prepare_tiles(s, want_lines, c->stream));
  ScanArgs a = scan_args(s, scan_variant(want_nl, want_lines));
  a.lines_only = want_lines;  // XSG_COUNT_LINES: what xsg_count launches for it (enqueue_count)
  s->cnt_clean = s->sum_clean = false;  // no finish kernel runs behind these launches
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(launch_scan_count(a, want_nl, want_lines, c->stream));  // warm-up
  HIP_TRY(hipEventRecord(e0, c->stream));
  for (int i = 0; i < iters; ++i) HIP_TRY(launch_scan_count(a, want_nl, want_lines, c->stream));
  HIP_TRY(hipEventRecord(e1, c->stream));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = ms / (float)iters;
  HIP_TRY(hipMemsetAsync(a.flags, 0, 4, c->stream));  // no finish kernel consumed what the scans may have raised
  return XSG_OK;
}

extern "C" int xsg_scan_kernel_name(xsg_shard* s, uint32_t mode, char* out, size_t cap) {
  XSG_TRY(check_ready(s));
  if (!out || !cap) return fail(XSG_EINVAL, "null output");
  const uint32_t m = mode & 0xffu;
  const bool list = m >= XSG_MATCH_BYTE_OFFSETS;
  // what the FIRST pass of this mode launches on this shard right now (newline counts already cached -> plain kernel)
  const bool want_nl = ((mode & XSG_WITH_NEWLINES) != 0 || m == XSG_LINE_INDICES) && !s->nl_cached;
  ScanArgs a = scan_args(s, scan_variant(want_nl, !list && m == XSG_COUNT_LINES));
  if (use_prefilter(s)) {  // what xsg_count / xsg_search launch: the candidate scan, then the automaton at candidates
    a.pat = s->ctx->pre_pat;
    a.pat.hot = 0;
    char inner[160];
    describe_scan(a, want_nl, false, false, inner, sizeof inner);
    snprintf(out, cap, "%s + xsg::k_rx_verify (prefilter route; xsg_count_async: k_rx_scan)", inner);
    return XSG_OK;
  }
  describe_scan(a, want_nl, !list && m == XSG_COUNT_LINES, false, out, cap);
  return XSG_OK;
}

// Picks the wave stagger of the bulk kernel for THIS shard, pattern and mode by measurement instead of the
// per-variant default (the optimum is sharp and depends on how memory-bound the variant is on the actual data:
// a needle that is dense in this text wants none).  A few launches per candidate; shards under 1 GiB keep the default.
extern "C" int xsg_shard_tune(xsg_shard* s, uint32_t mode, uint32_t* chosen) {
  XSG_TRY(check_ready(s));
  xsg_ctx* c = s->ctx;
  HIP_TRY(hipSetDevice(c->device));
  s->tune = kTuneAuto;
  s->tune_serial = 0;  // (0: the candidate values set inside the loop below apply whatever the serial)
  s->tune_probe = false;
  if (chosen) *chosen = kTuneAuto;
  if (c->tune != kTuneAuto || s->total_bytes < (1ull << 30)) return XSG_OK;  // XSG_TUNE wins; too small to measure
  if (c->pat.kind == kDfa) return XSG_OK;  // k_rx_scan has no stagger
  static const uint32_t cand[] = {0, 4, 8, 10, 12, 14, 16, 20};
  float best_ms = 0;
  uint32_t best = kTuneAuto, best_hot = 0;
  const uint32_t nhot = (is_window_kind(c->pat.kind) && c->hot_env < 0) ? 2u : 1u;
  // the probe first (it also settles a long pattern's filter window, which the loop below keeps), then both hot
  // filters against every stagger at full size
  const bool tune_nl = (mode & XSG_WITH_NEWLINES) != 0, tune_lines = (mode & 0xffu) == XSG_COUNT_LINES;
  const uint32_t v = scan_variant(tune_nl, tune_lines);
  s->hot_serial = 0;
  XSG_TRY(choose_hot_filter(s, c->stream, tune_nl, tune_lines));
  if (nhot == 2) best_hot = s->hot_v[v];
  auto give_up = [&](int r) {
    s->tune = kTuneAuto;
    s->hot_serial = 0;
    s->koff_chosen = false;
    return r;
  };
  // The clocks first: an idle card ramps for several ms and the candidates measured first (the window filter, the small
  // staggers) would be read 3-7 % low -- one bench run in three came back with the slower filter.  ~150 ms of untimed
  // launches, then TWO sweeps and every candidate's better time (profiles/r04_dense_variants.txt).
  {
    float ms = 0;
    s->tune = kDefaultStagger;
    int r = xsg_time_scan_kernel(s, mode, 3, &ms);
    if (r != XSG_OK) return give_up(r);
    const int more = (int)std::min(40.0f, std::max(0.0f, 150.0f / std::max(ms, 0.05f) - 3.0f));
    if (more > 0 && (r = xsg_time_scan_kernel(s, mode, more, &ms)) != XSG_OK) return give_up(r);
  }
  constexpr int kCand = (int)(sizeof cand / sizeof cand[0]);
  float t_ms[2][kCand];
  for (auto& row : t_ms)
    for (float& x : row) x = 1e30f;
  for (int round = 0; round < 2; ++round) {
    for (uint32_t hot = 0; hot < nhot; ++hot) {
      if (nhot == 2) {
        s->hot_v[v] = (uint8_t)hot;
        s->hot_known |= (uint8_t)(1u << v);
      }
      for (int k = 0; k < kCand; ++k) {
        s->tune = cand[k];
        float ms = 0;
        const int r = xsg_time_scan_kernel(s, mode, 3, &ms);
        if (r != XSG_OK) return give_up(r);
        t_ms[hot][k] = std::min(t_ms[hot][k], ms);
      }
    }
  }
  for (uint32_t hot = 0; hot < nhot; ++hot)
    for (int k = 0; k < kCand; ++k)
      if (best == kTuneAuto || t_ms[hot][k] < best_ms) best_ms = t_ms[hot][k], best = cand[k], best_hot = hot;
  s->tune = best;
  s->tune_serial = c->pattern_serial;
  s->tune_probe = false;
  if (nhot == 2) s->hot_v[v] = (uint8_t)best_hot;
  if (chosen) *chosen = best;
  return XSG_OK;
}

// ---------------------------------------------------------------------------
// list searches
// ---------------------------------------------------------------------------
static int d2h_u64(xsg_ctx* c, const uint64_t* d, uint64_t* h) {
  HIP_TRY(hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XSG_OK;
}

// ---------------------------------------------------------------------------
// The one-sync list route.  A list search on the exact route below fetches three to four sizes from the device (raw
// occurrences, kept + tail matches, line bytes), each a stream sync, and launches 17-25 small kernels -- ~0.45 ms on
// top of a 1.5 ms scan of 10 GiB for a few thousand matches (profiles/r03_list_before_kernel_trace.txt).  Here the
// sizes stay on the device: arrays get capacities (a sparse result fits them by a wide margin), every kernel reads
// the counts it needs from a block of device words (FastTot) and bounds itself, the tile ranks and the keep prefix
// take two launches each (ticketed last workgroup), the emit pass visits only the tiles that hold a match, the
// end-of-chunk walk is one wave per chunk on bit masks, and totals and results are ALSO stored into pinned host
// memory by the kernels that produce them.  The host syncs once and reads them there.  A result that does not fit
// (kTotOverflow) is redone on the exact route, which reuses the tile counts of this pass; the binding remembers it.
// ---------------------------------------------------------------------------
constexpr int kFastOverflow = 2;  // run_list_fast: capacity exceeded, tile counts in place -> the exact route from step 2

static uint64_t fast_capacity(const xsg_shard* s) {
  if (const char* e = XSG_TOGGLE("XSG_LIST_CAP")) {  // tests: tiny capacities force the fallback
    const long long v = atoll(e);
    if (v > 0) return (uint64_t)v;
  }
  // one entry per 256 bytes of text, 16 Ki .. 1 Mi entries (a 16 MiB chunk of the file pipeline: 64 Ki)
  return std::min<uint64_t>(std::max<uint64_t>(s->total_bytes / 256, 1u << 14), 1u << 20);
}

static bool fast_route_serves(const xsg_shard* s, uint32_t mode, bool outputs) {
  const xsg_ctx* c = s->ctx;
  const char* e = XSG_TOGGLE("XSG_LIST_FAST");  // 0: every list search takes the exact route (tests, A/B)
  if ((e && *e == '0') || !outputs || s->ntiles == 0 || s->want_nl_total) return false;
  if (c->pat.kind == kDfa) return false;                                // k_rx_scan / the prefilter route: exact route
  if (mode != XSG_MATCH_BYTE_OFFSETS && c->pat.has_newline) return false;  // the line walk of a literal with '\n': a chain, exact route
  if (mode == XSG_MATCH_BYTE_OFFSETS && c->bordered && !overlap_free_known(s)) return false;  // greedy keep: exact route
  if (s->chunks.size() > (1u << 20)) return false;                      // the tail prefix is one workgroup's work
  if (s->ntiles >= (1ull << 32)) return false;                          // hit list: uint32 tile numbers
  return s->fast_dense_serial != c->pattern_serial && c->fast_dense_serial != c->pattern_serial;
}

// The pinned mirrors of list results only grow while results grow: one needle in most lines of a large shard leaves
// gigabytes page-locked (offsets, line lengths, line bytes).  A search that needs less than a sixteenth of what is
// retained gives the large buffers back before it runs (they come again on demand; a caller that repeats the dense
// search keeps them: its results keep needing them).  Sizes in bytes.
static void trim_pinned(xsg_shard* s, size_t need_u64, size_t need_len, size_t need_bytes) {
  constexpr size_t kKeep = 64u << 20;
  if (s->h_result && s->h_result_cap > kKeep && need_u64 < s->h_result_cap / 16) {
    (void)hipHostFree(s->h_result);
    s->h_result = nullptr;
    s->h_result_cap = 0;
  }
  if (s->hp_line_len && 8 * s->hp_line_len_cap > kKeep && need_len < 8 * s->hp_line_len_cap / 16) {
    (void)hipHostFree(s->hp_line_len);
    s->hp_line_len = nullptr;
    s->hp_line_len_cap = 0;
  }
  if (s->hp_line_bytes && s->hp_line_bytes_cap > kKeep && need_bytes < s->hp_line_bytes_cap / 16) {
    (void)hipHostFree(s->hp_line_bytes);
    s->hp_line_bytes = nullptr;
    s->hp_line_bytes_cap = 0;
  }
}

template <typename T>
static int ensure_pinned(T** p, size_t* cap_elems, size_t want_elems) {
  if (*p && *cap_elems >= want_elems) return XSG_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *cap_elems = 0;
  hipError_t e = hipHostMalloc((void**)p, std::max<size_t>(want_elems, 1) * sizeof(T), hipHostMallocDefault);
  if (e != hipSuccess) return fail(XSG_ENOMEM, "hipHostMalloc(%zu) failed: %s", want_elems * sizeof(T), hipGetErrorString(e));
  *cap_elems = want_elems;
  return XSG_OK;
}

static int run_list_fast(xsg_shard* s, uint32_t mode) {
  xsg_ctx* c = s->ctx;
  hipStream_t st = c->stream;
  const bool line_mode = mode != XSG_MATCH_BYTE_OFFSETS;
  const bool want_f = mode == XSG_LINE_INDICES || mode == XSG_LINES;
  const uint64_t nchunks = s->chunks.size(), ntiles = s->ntiles;
  const uint64_t cap = fast_capacity(s);
  const uint32_t tail_cap = std::max<uint32_t>(tail_max_matches(c->pat.plen), 1u);
  const uint64_t fcap = cap + nchunks * tail_cap;  // a list whose raw part fits always fits
  const uint64_t bytes_cap = std::min<uint64_t>(fcap * 128, 32ull << 20);

  // ---- buffers (grow-only; the file pipeline re-binds the same shard for every chunk)
  bool grew = false;
  XSG_TRY(s->d_tot.ensure(8 * kTotWords + 64, &grew));
  if (grew) HIP_TRY(hipMemsetAsync(s->d_tot.p, 0, 8 * kTotWords + 64, st));  // tickets = 0
  if (!s->h_tot) HIP_TRY(hipHostMalloc((void**)&s->h_tot, 8 * (kTotWords + 1), hipHostMallocDefault));
  XSG_TRY(s->d_tile_off.ensure(8 * (ntiles + 1)));
  XSG_TRY(s->d_scan2.ensure(8 * scan2_tmp_elems(std::max<uint64_t>(ntiles, fcap) + 1)));
  XSG_TRY(s->d_hit.ensure(4 * cap));
  grew = false;
  XSG_TRY(s->d_wmask.ensure(4 * (ntiles / 4 + 1), &grew));
  if (grew) HIP_TRY(hipMemsetAsync(s->d_wmask.p, 0, s->d_wmask.cap, st));
  XSG_TRY(s->d_m_pos.ensure(8 * cap));
  XSG_TRY(s->d_m_chunk.ensure(4 * cap));
  if (line_mode) {
    XSG_TRY(s->d_m_ls.ensure(8 * cap));
    XSG_TRY(s->d_keep.ensure(4 * cap));
    XSG_TRY(s->d_keep_pre.ensure(8 * (cap + 1)));
  }
  XSG_TRY(s->d_chunk_shift0.ensure(8 * std::max<uint64_t>(nchunks, 1)));
  XSG_TRY(s->d_tail_cnt.ensure(4 * std::max<uint64_t>(nchunks, 1)));
  XSG_TRY(s->d_tail_pos.ensure(8 * std::max<uint64_t>(nchunks, 1) * tail_cap));
  XSG_TRY(s->d_tail_pre.ensure(8 * (nchunks + 1)));
  XSG_TRY(s->d_out_u64.ensure(8 * fcap));
  if (want_f) {
    XSG_TRY(s->d_f_pos.ensure(8 * fcap));
    XSG_TRY(s->d_f_match.ensure(8 * fcap));
    XSG_TRY(s->d_f_chunk.ensure(4 * fcap));
  }
  trim_pinned(s, 8 * (size_t)fcap, 8 * (size_t)fcap, (size_t)bytes_cap);  // what an earlier dense result left page-locked
  {
    size_t have = s->h_result_cap / 8;
    uint64_t* hp = static_cast<uint64_t*>(s->h_result);
    const int pr = ensure_pinned(&hp, &have, (size_t)fcap);
    s->h_result = hp;  // (written back on failure too: the old buffer is gone then)
    s->h_result_cap = have * 8;
    XSG_TRY(pr);
  }
  if (mode == XSG_LINES) {
    XSG_TRY(s->d_line_len.ensure(8 * fcap));
    XSG_TRY(s->d_line_off.ensure(8 * (fcap + 1)));
    XSG_TRY(s->d_line_bytes.ensure(bytes_cap));
    XSG_TRY(ensure_pinned(&s->hp_line_len, &s->hp_line_len_cap, (size_t)fcap));
    XSG_TRY(ensure_pinned(&s->hp_line_bytes, &s->hp_line_bytes_cap, (size_t)bytes_cap));
  }
  const bool want_nl = mode == XSG_LINE_INDICES;
  if (want_nl) {
    XSG_TRY(ensure_tile_nl(s));
    bool g2 = false;
    XSG_TRY(s->d_tile_nl_off.ensure(8 * (ntiles + 1), &g2));
    if (g2) s->nl_off_cached = false;
  }
  uint64_t* tot = s->d_tot.as<uint64_t>();
  uint32_t* tickets = reinterpret_cast<uint32_t*>(tot + kTotWords);
  memset(s->h_tot, 0, 8 * (kTotWords + 1));  // nothing of this shard is in flight: every search ends in a sync

  // ---- 1. bulk count per tile (+ newlines per tile, once per binding)
  const bool scan_nl = want_nl && !s->nl_cached;
  XSG_TRY(choose_hot_filter(s, st, scan_nl, false));
  XSG_TRY(prepare_tiles(s, false, st));
  ScanArgs a = scan_args(s, scan_variant(scan_nl, false));
  a.tile_wmask = s->d_wmask.as<uint32_t>();  // the count pass marks the waves that found something, the emit pass reads only those
  s->cnt_clean = false;  // the tile counts stay in place for the emit pass: the next pass re-zeroes them
  HIP_TRY(launch_scan_count(a, scan_nl, false, st));
  if (scan_nl) s->nl_cached = true;

  // ---- 2. ranks of the tiles + the ordered list of the tiles that hold a match (two launches)
  Scan2Args r{};
  r.in = a.tile_cnt;
  r.out = s->d_tile_off.as<uint64_t>();
  r.n_cap = ntiles;
  r.blk = s->d_scan2.as<uint64_t>();
  r.ticket = tickets;
  r.tot_dev = tot + kTotRaw;
  r.tot_host = s->h_tot + kTotRaw;
  r.hit_idx = s->d_hit.as<uint32_t>();
  r.hit_cap = cap;
  r.hits_dev = tot + kTotHits;
  r.hits_host = s->h_tot + kTotHits;
  r.ovf_dev = tot + kTotOverflow;
  r.ovf_host = s->h_tot + kTotOverflow;
  r.total_cap = cap;
  r.ovf_bit = 1;
  r.ovf_init = 1;
  HIP_TRY(launch_scan2_u32(r, true, st));
  if (want_nl && !s->nl_off_cached) {
    Scan2Args n{};
    n.in = a.tile_nl;
    n.out = s->d_tile_nl_off.as<uint64_t>();
    n.n_cap = ntiles;
    n.blk = r.blk;
    n.ticket = tickets;
    n.tot_dev = tot + kTotNewlines;
    n.tot_host = s->h_tot + kTotNewlines;
    HIP_TRY(launch_scan2_u32(n, false, st));
  }

  // ---- 3. ordered emission, only from the tiles on the list
  a.tile_off = r.out;
  a.m_pos = s->d_m_pos.as<uint64_t>();
  a.m_chunk = s->d_m_chunk.as<uint32_t>();
  a.m_cap = cap;
  a.hit_tiles = r.hit_idx;
  a.n_hits_dev = tot + kTotHits;
  a.hit_cap = cap;
  HIP_TRY(launch_scan_emit(a, st));

  // ---- 4. which occurrences the walk reports; 5. the end of every chunk; 6. the list
  ListArgs l{};
  l.base = s->base;
  l.chunks = a.chunks;
  l.chunk_tile0 = a.chunk_tile0;
  l.nchunks = nchunks;
  l.pat = a.pat;
  l.M = cap;
  l.M_dev = tot + kTotRaw;
  l.m_pos = a.m_pos;
  l.m_chunk = a.m_chunk;
  l.tile_off = a.tile_off;
  l.m_ls = s->d_m_ls.as<uint64_t>();
  l.keep = s->d_keep.as<uint32_t>();
  l.keep_pre = s->d_keep_pre.as<uint64_t>();
  l.chunk_shift0 = s->d_chunk_shift0.as<uint64_t>();
  l.tail_cnt = s->d_tail_cnt.as<uint32_t>();
  l.tail_pos = s->d_tail_pos.as<uint64_t>();
  l.tail_cap = tail_cap;
  l.tail_pre = s->d_tail_pre.as<uint64_t>();
  l.line_mode = line_mode ? 1u : 0u;
  l.keep_all = line_mode ? 0u : 1u;
  l.tot_dev = tot;
  l.tot_host = s->h_tot;
  l.ticket = tickets;
  l.f_cap = fcap;
  l.f_pos = s->d_f_pos.as<uint64_t>();
  l.f_match = s->d_f_match.as<uint64_t>();
  l.f_chunk = s->d_f_chunk.as<uint32_t>();
  l.want_f = want_f ? 1u : 0u;
  // the global offsets of the final entries leave with k_list_out for every tag but xs::line_indices (whose values
  // are indices); xs::lines also gets its line lengths there
  l.out_u64 = mode == XSG_LINE_INDICES ? nullptr : s->d_out_u64.as<uint64_t>();
  l.out_host = mode == XSG_LINE_INDICES ? nullptr : static_cast<uint64_t*>(s->h_result);
  if (mode == XSG_LINES) {
    l.line_len = s->d_line_len.as<uint64_t>();
    l.line_len_host = s->hp_line_len;
  }
  if (line_mode) {
    HIP_TRY(launch_line_starts_keep(l, st));
    Scan2Args k{};
    k.in = l.keep;
    k.out = s->d_keep_pre.as<uint64_t>();
    k.n_cap = cap;
    k.n_dev = tot + kTotRaw;
    k.blk = r.blk;
    k.ticket = tickets;
    HIP_TRY(launch_scan2_u32(k, false, st));
  }
  HIP_TRY(launch_chunk_tail(l, st));
  HIP_TRY(launch_list_out(l, st));

  if (want_f) {
    LineOutArgs o{};
    o.base = s->base;
    o.chunks = a.chunks;
    o.chunk_tile0 = a.chunk_tile0;
    o.nchunks = nchunks;
    o.pat = c->pat;
    o.total = fcap;
    o.tot_dev = tot;
    o.f_pos = l.f_pos;
    o.f_match = l.f_match;
    o.f_chunk = l.f_chunk;
    o.out_u64 = s->d_out_u64.as<uint64_t>();
    o.out_host = static_cast<uint64_t*>(s->h_result);
    o.shard_line_base = s->shard_line_base;
    o.tile_bytes = s->tile_bytes;
    if (mode == XSG_LINE_INDICES) {
      o.tile_nl_off = s->d_tile_nl_off.as<uint64_t>();
      HIP_TRY(launch_line_index_waves(o, st));
    } else {
      o.line_len = s->d_line_len.as<uint64_t>();
      o.line_len_host = s->hp_line_len;
      o.line_out_off = s->d_line_off.as<uint64_t>();
      o.line_bytes = s->d_line_bytes.as<uint8_t>();
      o.line_bytes_host = s->hp_line_bytes;
      o.line_bytes_cap = bytes_cap;
      Scan2Args b{};  // (the lengths came with k_list_out)
      b.in = o.line_len;
      b.out = s->d_line_off.as<uint64_t>();
      b.n_cap = fcap;
      b.n_dev = tot + kTotFinal;
      b.blk = r.blk;
      b.ticket = tickets;
      b.tot_dev = tot + kTotLineBytes;
      b.tot_host = s->h_tot + kTotLineBytes;
      b.ovf_dev = tot + kTotOverflow;
      b.ovf_host = s->h_tot + kTotOverflow;
      b.total_cap = bytes_cap;
      b.ovf_bit = 4;
      HIP_TRY(launch_scan2_u64(b, st));
      HIP_TRY(launch_line_gather(o, st));
    }
  }
  const bool ascii_only = c->pat.kind == kClass && c->pat.ascii_only;
  if (ascii_only) HIP_TRY(hipMemcpyAsync(s->h_tot + kTotWords, a.flags, 4, hipMemcpyDeviceToHost, st));

  // ---- the one sync
  HIP_TRY(hipStreamSynchronize(st));
  s->table_pending = false;
  if (ascii_only && (s->h_tot[kTotWords] & 1u)) {  // non-ASCII data under an ascii_only expression
    HIP_TRY(hipMemsetAsync(a.flags, 0, 4, st));
    return fail(XSG_ENOTSUP, "%s", kNonAsciiMsg);
  }
  s->last_raw_matches = s->h_tot[kTotRaw];
  if (want_nl && !s->nl_off_cached) {
    s->nl_total = s->h_tot[kTotNewlines];
    s->nl_off_cached = true;
  }
  if (s->h_tot[kTotOverflow]) {
    s->fast_dense_serial = c->pattern_serial;  // later searches of this pattern on this binding: the exact route at once
    c->fast_dense_serial = c->pattern_serial;  // ... and on later bindings of this context (the next chunks of a file)
    return kFastOverflow;
  }
  const uint64_t total = s->h_tot[kTotFinal];
  s->total = total;
  s->fast_result = true;
  if (want_nl) s->last_newlines = s->nl_total;
  if (mode == XSG_LINES) {
    // lines without a terminating '\n' are not reported (search_wrappers.h:199-202)
    uint64_t n = 0;
    for (uint64_t i = 0; i < total; ++i) n += s->hp_line_len[i] != UINT64_MAX;
    s->fast_raw_lines = total;
    s->total = n;
    s->line_bytes = s->h_tot[kTotLineBytes];
  }
  s->last_mode = (int)mode;
  return XSG_OK;
}

static int run_list(xsg_shard* s, uint32_t mode, bool outputs) {
  xsg_ctx* c = s->ctx;
  hipStream_t st = c->stream;
  const bool line_mode = mode != XSG_MATCH_BYTE_OFFSETS;
  const bool want_nl = mode == XSG_LINE_INDICES || s->want_nl_total;
  const uint64_t nchunks = s->chunks.size();
  const uint64_t ntiles = s->ntiles;
  if (line_mode && c->pat.has_newline && !newline_literal(c)) return fail(XSG_ENOTSUP, "%s", kNewlineExprMsg);
  const bool chain_lines = line_mode && newline_literal(c);

  s->last_mode = -1;
  s->total = 0;
  s->line_bytes = 0;
  s->fast_result = false;
  s->line_len_on_device = false;
  XSG_TRY(ensure_factor_mask(s));

  if (mode == XSG_MATCH_BYTE_OFFSETS && c->bordered) XSG_TRY(ensure_overlap_check(s));
  // 0. a result that fits the one-sync route's capacities is done there (one stream sync, a third of the launches)
  bool counts_ready = false;
  if (fast_route_serves(s, mode, outputs)) {
    const int fr = run_list_fast(s, mode);
    if (fr != kFastOverflow) return fr;
    counts_ready = true;  // the per-tile counts (and newline counts) of that pass stand: continue at the ranks
    s->last_mode = -1;
    s->total = 0;
  }

  // 1. bulk count per tile
  if (want_nl) XSG_TRY(ensure_tile_nl(s));
  const bool scan_nl = want_nl && !s->nl_cached;  // newline counts per tile: once per binding, whatever the pattern
  if (!counts_ready) {
    XSG_TRY(choose_hot_filter(s, st, scan_nl, false));
    XSG_TRY(prepare_tiles(s, false, st));
  }
  ScanArgs a = scan_args(s, scan_variant(scan_nl, false));
  const bool pre = use_prefilter(s);  // candidates by the class-sequence matcher, then the automaton
  if (pre) {
    a.pat = c->pre_pat;
    a.pat.hot = 0;
  }
  s->cnt_clean = false;  // the tile counts stay in place for the emit pass: the next pass re-zeroes them
  if (!counts_ready) HIP_TRY(launch_scan_count(a, scan_nl, false, st));
  if (scan_nl) s->nl_cached = true;

  // 2. ranks
  XSG_TRY(s->d_tile_off.ensure(8 * (ntiles + 1)));
  XSG_TRY(s->d_scan_tmp.ensure(8 * scan_tmp_elems(std::max<uint64_t>(ntiles, nchunks) + 1)));
  HIP_TRY(launch_exclusive_scan_u32(a.tile_cnt, s->d_tile_off.as<uint64_t>(), ntiles, s->d_scan_tmp.as<uint64_t>(), st));
  uint64_t M = 0;
  uint32_t scan_flags = 0;
  if ((c->pat.kind == kClass || c->pat.kind == kDfa) && c->pat.ascii_only)
    HIP_TRY(hipMemcpyAsync(&scan_flags, a.flags, 4, hipMemcpyDeviceToHost, st));

  XSG_TRY(d2h_u64(c, s->d_tile_off.as<uint64_t>() + ntiles, &M));
  if (scan_flags & 1u) {  // non-ASCII data under an ascii_only expression
    HIP_TRY(hipMemsetAsync(a.flags, 0, 4, st));
    return fail(XSG_ENOTSUP, "%s", kNonAsciiMsg);
  }

  s->last_raw_matches = M;  // sizes the arrays of the next device-only pass (enqueue_count_bordered)
  // 3. ordered emission of every bulk occurrence
  a.tile_off = s->d_tile_off.as<uint64_t>();
  if (!pre) {
    XSG_TRY(s->d_m_pos.ensure(8 * std::max<uint64_t>(M, 1)));
    XSG_TRY(s->d_m_chunk.ensure(4 * std::max<uint64_t>(M, 1)));
    a.m_pos = s->d_m_pos.as<uint64_t>();
    a.m_chunk = s->d_m_chunk.as<uint32_t>();
    if (M) HIP_TRY(launch_scan_emit(a, st));
  } else {
    // Candidates every few hundred bytes (a start that is a word of the text): a count is cheaper by walking all
    // lines once (k_rx_scan + finish, no list at all) than by listing, verifying and packing tens of millions of
    // entries -- measured on the bench corpus, where `Sher` is a lexicon word: count_lines of `lock(ed|s)?` 18 ms
    // by candidates against 10 ms by k_rx_scan (8 GiB).  The caller takes the other route.
    if (!c->pat.rx_multiline && M * 128 > s->total_bytes) {
      s->cnt_clean = false;
      s->pre_dense_serial = c->pattern_serial;  // later counts of this pattern on this binding go straight to k_rx_scan
      if (!outputs) return kDenseCandidates;
      // a list: the same verdict (tens of millions of anchored scans, each up to 4 KiB, and chains between them, against
      // one walk of the text) -- redone on the line-walking route
      s->pre_off = true;
      const int rr = run_list(s, mode, outputs);
      s->pre_off = false;
      return rr;
    }
    // M candidates so far: emit them, run the anchored automaton at each, walk every chunk's occurrences as the
    // reference does, pack what it reports -- then M is the number of matches and the list is what the emit pass
    // of k_rx_scan would have written
    const uint64_t Mc = M;
    XSG_TRY(s->d_c_pos.ensure(8 * std::max<uint64_t>(Mc, 1)));
    XSG_TRY(s->d_c_chunk.ensure(4 * std::max<uint64_t>(Mc, 1)));
    XSG_TRY(s->d_c_len.ensure(4 * std::max<uint64_t>(Mc, 1)));
    XSG_TRY(s->d_c_keep.ensure(4 * std::max<uint64_t>(Mc, 1)));
    XSG_TRY(s->d_c_pre.ensure(8 * (Mc + 1)));
    XSG_TRY(s->d_scan_tmp.ensure(8 * scan_tmp_elems(std::max<uint64_t>(Mc, std::max<uint64_t>(ntiles, nchunks)) + 1)));
    a.m_pos = s->d_c_pos.as<uint64_t>();
    a.m_chunk = s->d_c_chunk.as<uint32_t>();
    if (Mc) HIP_TRY(launch_scan_emit(a, st));
    RxPreArgs r{};
    r.base = s->base;
    r.chunks = a.chunks;
    r.chunk_tile0 = a.chunk_tile0;
    r.nchunks = nchunks;
    r.pat = c->pat;
    r.n = Mc;
    r.tile_off = a.tile_off;
    r.c_pos = a.m_pos;
    r.c_chunk = a.m_chunk;
    r.c_len = s->d_c_len.as<uint32_t>();
    r.c_keep = s->d_c_keep.as<uint32_t>();
    r.c_pre = s->d_c_pre.as<uint64_t>();
    r.scan_tmp = s->d_scan_tmp.as<uint64_t>();
    r.flags = a.flags;
    HIP_TRY(launch_rx_verify_keep(r, st));
    HIP_TRY(launch_exclusive_scan_u32(r.c_keep, s->d_c_pre.as<uint64_t>(), Mc, s->d_scan_tmp.as<uint64_t>(), st));
    uint32_t vflags = 0;
    HIP_TRY(hipMemcpyAsync(&vflags, a.flags, 4, hipMemcpyDeviceToHost, st));
    XSG_TRY(d2h_u64(c, s->d_c_pre.as<uint64_t>() + Mc, &M));
    if (vflags & 2u) {  // a candidate outran the verification budget: walk the text once instead (the other route)
      HIP_TRY(hipMemsetAsync(a.flags, 0, 4, st));
      s->cnt_clean = false;
      s->pre_off = true;
      const int rr = run_list(s, mode, outputs);
      s->pre_off = false;
      return rr;
    }
    XSG_TRY(s->d_m_pos.ensure(8 * std::max<uint64_t>(M, 1)));
    XSG_TRY(s->d_m_chunk.ensure(4 * std::max<uint64_t>(M, 1)));
    r.m_pos = s->d_m_pos.as<uint64_t>();
    r.m_chunk = s->d_m_chunk.as<uint32_t>();
    HIP_TRY(launch_rx_compact(r, st));
    a.m_pos = r.m_pos;
    a.m_chunk = r.m_chunk;
  }

  // 4. which occurrences the reference walk reports
  XSG_TRY(s->d_keep.ensure(4 * std::max<uint64_t>(M, 1)));
  XSG_TRY(s->d_keep_pre.ensure(8 * (M + 1)));
  XSG_TRY(s->d_scan_tmp.ensure(8 * scan_tmp_elems(std::max<uint64_t>(M, std::max<uint64_t>(ntiles, nchunks)) + 1)));
  if (line_mode) XSG_TRY(s->d_m_ls.ensure(8 * std::max<uint64_t>(M, 1)));
  const uint32_t tail_cap = std::max<uint32_t>(tail_max_matches(c->pat.plen), 1u);
  XSG_TRY(s->d_chunk_shift0.ensure(8 * std::max<uint64_t>(nchunks, 1)));
  XSG_TRY(s->d_tail_cnt.ensure(4 * std::max<uint64_t>(nchunks, 1)));
  XSG_TRY(s->d_tail_pos.ensure(8 * std::max<uint64_t>(nchunks, 1) * tail_cap));
  XSG_TRY(s->d_tail_pre.ensure(8 * (nchunks + 1)));

  ListArgs l{};
  l.base = s->base;
  l.chunks = a.chunks;
  l.chunk_tile0 = a.chunk_tile0;
  l.nchunks = nchunks;
  l.pat = c->pat;
  l.M = M;
  l.m_pos = a.m_pos;
  l.m_chunk = a.m_chunk;
  l.tile_off = a.tile_off;
  l.m_ls = s->d_m_ls.as<uint64_t>();
  l.keep = s->d_keep.as<uint32_t>();
  l.keep_pre = s->d_keep_pre.as<uint64_t>();
  l.chunk_shift0 = s->d_chunk_shift0.as<uint64_t>();
  l.tail_cnt = s->d_tail_cnt.as<uint32_t>();
  l.tail_pos = s->d_tail_pos.as<uint64_t>();
  l.tail_cap = tail_cap;
  l.tail_pre = s->d_tail_pre.as<uint64_t>();
  l.line_mode = line_mode ? 1u : 0u;

  // words behind the finish kernel's ticket and the scan flags: [0] "a chain outran its walker", [1] "a round marked something"
  uint32_t* words = reinterpret_cast<uint32_t*>(s->d_finish.as<uint64_t>() + 3 * (size_t)kFinishBlocks) + 2;
  // closure of the marked entries under the links J (J2: scratch): pointer jumping, log2(longest chain) rounds of
  // "mark J(marked), square J" until a round marks nothing new (xsg_list_kernels.hip: k_greedy_jump)
  auto close_chains = [&](uint32_t* J, uint32_t* J2) -> int {
    for (int round = 0; round < 40; ++round) {  // 2^40 entries would not fit the index type anyway
      uint32_t changed = 0;
      HIP_TRY(hipMemsetAsync(words + 1, 0, 4, st));
      HIP_TRY(launch_greedy_jump(l, J, J2, words + 1, st));
      HIP_TRY(hipMemcpyAsync(&changed, words + 1, 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (!changed) break;
      std::swap(J, J2);
    }
    return XSG_OK;
  };
  if (chain_lines) {
    // a literal that contains '\n': line starts, chunk heads and the walk's links, then the closure (see k_nlpat_links)
    if (M >= 0xffffffffull) return fail(XSG_ENOTSUP, "more than 2^32 occurrences of a pattern that contains '\\n': its line walk is not served");
    if (M) {
      XSG_TRY(s->d_c_pos.ensure(4 * M));  // the link arrays borrow the prefilter route's candidate buffers (unused by literals)
      XSG_TRY(s->d_c_pre.ensure(4 * M));
      HIP_TRY(launch_nlpat_links(l, s->d_c_pos.as<uint32_t>(), st));
      XSG_TRY(close_chains(s->d_c_pos.as<uint32_t>(), s->d_c_pre.as<uint32_t>()));
    }
  } else if (line_mode) {
    HIP_TRY(launch_line_starts_keep(l, st));
  } else if (c->bordered && !overlap_free_known(s)) {
    // chain heads walk their chains (a few entries at text densities); a chain over the budget -- a long run of one
    // byte searched for `aa` is ONE chain per chunk -- raises a flag and is finished by pointer jumping, log2(length)
    // parallel rounds (xsg_list_kernels.hip: k_greedy_links / k_greedy_jump)
    const bool can_jump = M < 0xffffffffull;
    if (M) HIP_TRY(hipMemsetAsync(l.keep, 0, 4 * M, st));
    HIP_TRY(hipMemsetAsync(words, 0, 8, st));
    l.long_flag = can_jump ? words : nullptr;
    HIP_TRY(launch_greedy_keep(l, st));
    uint32_t is_long = 0;
    if (can_jump) {
      HIP_TRY(hipMemcpyAsync(&is_long, words, 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    if (is_long) {
      XSG_TRY(s->d_c_pos.ensure(4 * M));  // the link arrays borrow the prefilter route's candidate buffers (unused by literals)
      XSG_TRY(s->d_c_pre.ensure(4 * M));
      HIP_TRY(launch_greedy_links(l, s->d_c_pos.as<uint32_t>(), st));
      XSG_TRY(close_chains(s->d_c_pos.as<uint32_t>(), s->d_c_pre.as<uint32_t>()));
    }
  } else {
    HIP_TRY(launch_keep_all(l, st));
  }
  HIP_TRY(launch_exclusive_scan_u32(l.keep, s->d_keep_pre.as<uint64_t>(), M, s->d_scan_tmp.as<uint64_t>(), st));

  // 5. the tail zone of every chunk, replayed as the reference walks it
  HIP_TRY(launch_chunk_shift0(l, st));
  HIP_TRY(launch_tail_list(l, st));
  HIP_TRY(launch_exclusive_scan_u32(l.tail_cnt, s->d_tail_pre.as<uint64_t>(), nchunks, s->d_scan_tmp.as<uint64_t>(), st));
  uint64_t kept = 0, tails = 0;
  HIP_TRY(hipMemcpyAsync(&kept, s->d_keep_pre.as<uint64_t>() + M, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(&tails, s->d_tail_pre.as<uint64_t>() + nchunks, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const uint64_t total = kept + tails;
  s->total = total;
  if (s->want_nl_total) {  // xsg_count(... | XSG_WITH_NEWLINES) on the prefilter route: the sum of the per-tile counts
    bool grew = false;
    XSG_TRY(s->d_tile_nl_off.ensure(8 * (ntiles + 1), &grew));
    if (grew) s->nl_off_cached = false;
    XSG_TRY(s->d_scan_tmp.ensure(8 * scan_tmp_elems(ntiles + 1)));
    if (!s->nl_off_cached) {
      HIP_TRY(launch_exclusive_scan_u32(a.tile_nl, s->d_tile_nl_off.as<uint64_t>(), ntiles, s->d_scan_tmp.as<uint64_t>(), st));
      s->nl_off_cached = true;
    }
    XSG_TRY(d2h_u64(c, s->d_tile_nl_off.as<uint64_t>() + ntiles, &s->last_newlines));
    s->nl_total = s->last_newlines;
  }
  if (!outputs) return XSG_OK;

  // 6. final list, in file order
  XSG_TRY(s->d_f_pos.ensure(8 * std::max<uint64_t>(total, 1)));
  XSG_TRY(s->d_f_match.ensure(8 * std::max<uint64_t>(total, 1)));
  XSG_TRY(s->d_f_chunk.ensure(4 * std::max<uint64_t>(total, 1)));
  XSG_TRY(s->d_out_u64.ensure(8 * std::max<uint64_t>(total, 1)));
  l.f_pos = s->d_f_pos.as<uint64_t>();
  l.f_match = s->d_f_match.as<uint64_t>();
  l.f_chunk = s->d_f_chunk.as<uint32_t>();
  l.total = total;
  HIP_TRY(launch_assemble(l, st));

  LineOutArgs o{};
  o.base = s->base;
  o.chunks = a.chunks;
  o.chunk_tile0 = a.chunk_tile0;
  o.nchunks = nchunks;
  o.pat = c->pat;
  o.total = total;
  o.f_pos = l.f_pos;
  o.f_match = l.f_match;
  o.f_chunk = l.f_chunk;
  o.out_u64 = s->d_out_u64.as<uint64_t>();
  o.shard_line_base = s->shard_line_base;
  o.tile_bytes = s->tile_bytes;

  if (mode == XSG_MATCH_BYTE_OFFSETS || mode == XSG_LINE_BYTE_OFFSETS) {
    HIP_TRY(launch_globalize(o, st));
  } else if (mode == XSG_LINE_INDICES) {
    {
      bool grew = false;
      XSG_TRY(s->d_tile_nl_off.ensure(8 * (ntiles + 1), &grew));
      if (grew) s->nl_off_cached = false;
    }
    XSG_TRY(s->d_line_len.ensure(8 * std::max<uint64_t>(total, 1)));
    XSG_TRY(s->d_line_off.ensure(8 * (total + 1)));
    XSG_TRY(s->d_scan_tmp.ensure(8 * scan_tmp_elems(std::max<uint64_t>(total + 1, ntiles + 1))));
    if (!s->nl_off_cached) {
      HIP_TRY(launch_exclusive_scan_u32(a.tile_nl, s->d_tile_nl_off.as<uint64_t>(), ntiles, s->d_scan_tmp.as<uint64_t>(),
                                        st));
      s->nl_off_cached = true;
    }
    o.tile_nl_off = s->d_tile_nl_off.as<uint64_t>();
    // per-entry newline differences -> prefix sums -> indices (k_line_nl_delta / k_line_indices)
    o.line_len = s->d_line_len.as<uint64_t>();
    o.line_out_off = s->d_line_off.as<uint64_t>();
    HIP_TRY(launch_line_nl_delta(o, st));
    HIP_TRY(launch_exclusive_scan_u64(o.line_len, s->d_line_off.as<uint64_t>(), total, s->d_scan_tmp.as<uint64_t>(), st));
    HIP_TRY(launch_line_indices(o, st));
    HIP_TRY(hipMemcpyAsync(&s->last_newlines, s->d_tile_nl_off.as<uint64_t>() + ntiles, 8, hipMemcpyDeviceToHost, st));
  } else {  // XSG_LINES
    XSG_TRY(s->d_line_len.ensure(8 * std::max<uint64_t>(total, 1)));
    XSG_TRY(s->d_line_off.ensure(8 * (total + 1)));
    XSG_TRY(s->d_scan_tmp.ensure(8 * scan_tmp_elems(total + 1)));
    o.line_len = s->d_line_len.as<uint64_t>();
    XSG_TRY(s->d_dropped.ensure(16));
    o.dropped = s->d_dropped.as<uint32_t>();
    HIP_TRY(hipMemsetAsync(o.dropped, 0, 4, st));
    // The result leaves for the shard's pinned mirrors AS IT IS PRODUCED (what xsg_result_lines_view hands out;
    // xsg_result_lines copies from there): k_line_lengths stores lengths and global offsets there as well as on the device,
    // and the gather writes the packed bytes straight into pinned memory -- the kernels are the copies.  A needle in most
    // lines of 10 GiB returns 3 GB over a link that moves 57 GB/s: round 3 moved them in three copies one after the other
    // behind the whole gather (94 ms a search); copies on side streams behind each slice of the gather came to 85 ms, because
    // a kernel that runs beside a device-to-host copy crawls (the copy is a blit kernel whose waves wait for the link and
    // hold the compute units: each 250 MB slice of the gather took 7 ms beside one, the first one 19 ms:
    // profiles/r04_dense_timeline.txt).  Now the link is busy from the moment the list is assembled until the last byte.
    bool eager = 16 * total < (16ull << 30);  // (beyond 16 GiB of pinned memory: the accessors copy on demand)
    if (const char* e = XSG_TOGGLE("XSG_LINES_EAGER")) eager = *e != '0';  // tests: the on-demand path on small results
    if (eager) {
      trim_pinned(s, 8 * (size_t)total, 8 * (size_t)total, SIZE_MAX);  // (what a far larger earlier result left page-locked)
      XSG_TRY(ensure_pinned(&s->hp_line_len, &s->hp_line_len_cap, (size_t)total));
      {
        size_t have = s->h_result_cap / 8;
        uint64_t* hp = static_cast<uint64_t*>(s->h_result);
        const int pr = ensure_pinned(&hp, &have, (size_t)total);
        s->h_result = hp;
        s->h_result_cap = have * 8;
        XSG_TRY(pr);
      }
      o.line_len_host = s->hp_line_len;
      o.out_host = static_cast<uint64_t*>(s->h_result);
    }
    HIP_TRY(launch_line_lengths(o, st));
    HIP_TRY(launch_exclusive_scan_u64(o.line_len, s->d_line_off.as<uint64_t>(), total, s->d_scan_tmp.as<uint64_t>(), st));
    uint64_t nbytes = 0;
    HIP_TRY(hipMemcpyAsync(&s->h_dropped, o.dropped, 4, hipMemcpyDeviceToHost, st));  // (rides on the sync below)
    XSG_TRY(d2h_u64(c, s->d_line_off.as<uint64_t>() + total, &nbytes));
    o.line_out_off = s->d_line_off.as<uint64_t>();
    if (!eager) {
      XSG_TRY(s->d_line_bytes.ensure(std::max<uint64_t>(nbytes, 1)));
      o.line_bytes = s->d_line_bytes.as<uint8_t>();
      HIP_TRY(launch_line_gather(o, st));
      // the lengths stay on the device until a result accessor asks (fetch_line_lengths): how many lines lack their
      // newline -- all the search itself needs to know -- was counted by the kernel
      s->line_len_on_device = true;
    } else {
      trim_pinned(s, SIZE_MAX, SIZE_MAX, (size_t)nbytes + 16);
      XSG_TRY(ensure_pinned(&s->hp_line_bytes, &s->hp_line_bytes_cap, (size_t)nbytes + 16));  // (+16: k_line_gather_edges writes whole units)
      {
        const uint64_t nbnd = total / kBlock + 2;
        XSG_TRY(s->d_scan_tmp.ensure(16 * nbnd));  // (the scan is done with it)
        HIP_TRY(hipMemsetAsync(s->d_scan_tmp.p, 0, 16 * nbnd, st));
        o.edge_units = s->d_scan_tmp.as<uint32_t>();
      }
      o.line_bytes = nullptr;  // no device copy of the packed bytes: nothing reads one when the mirrors hold the result
      o.line_bytes_host = s->hp_line_bytes;
      HIP_TRY(launch_line_gather(o, st));
      s->fast_result = true;  // the result lives in the pinned mirrors: the accessors read it there
    }
    s->line_bytes = nbytes;
    s->fast_raw_lines = total;
  }
  HIP_TRY(hipStreamSynchronize(st));
  if (mode == XSG_LINE_INDICES) s->nl_total = s->last_newlines;
  if (mode == XSG_LINES) {
    // lines without a terminating '\n' are not reported (search_wrappers.h:199-202)
    s->total = total - s->h_dropped;
  }
  s->last_mode = (int)mode;
  return XSG_OK;
}

extern "C" int xsg_search(xsg_shard* s, uint32_t mode, uint64_t* n_results) {
  XSG_TRY(check_ready(s));
  if (mode != XSG_MATCH_BYTE_OFFSETS && mode != XSG_LINE_BYTE_OFFSETS && mode != XSG_LINE_INDICES &&
      mode != XSG_LINES)
    return fail(XSG_EINVAL, "xsg_search: mode %u is not a list mode", mode);
  HIP_TRY(hipSetDevice(s->ctx->device));
  XSG_TRY(run_list(s, mode, true));
  if (!s->fast_result) trim_pinned(s, 8 * (size_t)s->total, 8 * (size_t)s->total, (size_t)s->line_bytes);  // (an exact-route result is still on the device)
  if (n_results) *n_results = s->total;
  return XSG_OK;
}

extern "C" int xsg_result_u64(xsg_shard* s, uint64_t* out, uint64_t cap) {
  if (!s) return fail(XSG_EINVAL, "shard is null");
  if (s->last_mode != XSG_MATCH_BYTE_OFFSETS && s->last_mode != XSG_LINE_BYTE_OFFSETS &&
      s->last_mode != XSG_LINE_INDICES)
    return fail(XSG_ESTATE, "no uint64 list result is pending on this shard");
  if (cap < s->total) return fail(XSG_EINVAL, "output capacity %llu < %llu results", (unsigned long long)cap,
                                  (unsigned long long)s->total);
  if (s->total == 0) return XSG_OK;
  if (!out) return fail(XSG_EINVAL, "out is null");
  if (s->fast_result) {  // the one-sync route: the kernels stored the result into the shard's pinned buffer as well
    memcpy(out, s->h_result, 8 * s->total);
    return XSG_OK;
  }
  xsg_ctx* c = s->ctx;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(out, s->d_out_u64.p, 8 * s->total, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XSG_OK;
}

// The same result without a copy into caller memory: moved once into a pinned buffer the shard owns (grow-only) and
// handed out as a pointer, valid until the next search on the shard.  A D2H copy into pageable memory runs at
// ~8 GB/s on this platform, into pinned memory at ~50: what matters when a dense needle returns hundreds of MB.
extern "C" int xsg_result_u64_view(xsg_shard* s, const uint64_t** out, uint64_t* n) {
  if (!s || !out || !n) return fail(XSG_EINVAL, "null argument");
  if (s->last_mode != XSG_MATCH_BYTE_OFFSETS && s->last_mode != XSG_LINE_BYTE_OFFSETS &&
      s->last_mode != XSG_LINE_INDICES)
    return fail(XSG_ESTATE, "no uint64 list result is pending on this shard");
  *out = nullptr;
  *n = s->total;
  if (s->total == 0) return XSG_OK;
  if (s->fast_result) {  // already there
    *out = static_cast<const uint64_t*>(s->h_result);
    return XSG_OK;
  }
  xsg_ctx* c = s->ctx;
  HIP_TRY(hipSetDevice(c->device));
  const size_t need = 8 * (size_t)s->total;
  if (need > s->h_result_cap) {
    if (s->h_result) (void)hipHostFree(s->h_result);
    s->h_result = nullptr;
    s->h_result_cap = 0;
    const size_t want = std::max(need, s->h_result_cap + s->h_result_cap / 2);
    hipError_t e = hipHostMalloc(&s->h_result, want, hipHostMallocDefault);
    if (e != hipSuccess) return fail(XSG_ENOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    s->h_result_cap = want;
  }
  HIP_TRY(hipMemcpyAsync(s->h_result, s->d_out_u64.p, need, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *out = static_cast<const uint64_t*>(s->h_result);
  return XSG_OK;
}

extern "C" int xsg_result_newlines(xsg_shard* s, uint64_t* newlines) {
  if (!s || !newlines) return fail(XSG_EINVAL, "null argument");
  if (s->last_mode != XSG_LINE_INDICES) return fail(XSG_ESTATE, "no XSG_LINE_INDICES result is pending on this shard");
  *newlines = s->last_newlines;
  return XSG_OK;
}

// xs::lines, exact route: the line lengths into the shard's pinned buffer (once per result)
static int fetch_line_lengths(xsg_shard* s) {
  if (!s->line_len_on_device) return XSG_OK;
  xsg_ctx* c = s->ctx;
  const uint64_t raw = s->fast_raw_lines;
  HIP_TRY(hipSetDevice(c->device));
  XSG_TRY(ensure_pinned(&s->hp_line_len, &s->hp_line_len_cap, (size_t)raw));
  if (raw) HIP_TRY(hipMemcpyAsync(s->hp_line_len, s->d_line_len.p, 8 * raw, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  s->line_len_on_device = false;
  return XSG_OK;
}

extern "C" int xsg_result_lines_size(xsg_shard* s, uint64_t* n_lines, uint64_t* total_bytes) {
  if (!s) return fail(XSG_EINVAL, "shard is null");
  if (s->last_mode != XSG_LINES) return fail(XSG_ESTATE, "no XSG_LINES result is pending on this shard");
  if (n_lines) *n_lines = s->total;
  if (total_bytes) *total_bytes = s->line_bytes;
  return XSG_OK;
}

extern "C" int xsg_result_lines(xsg_shard* s, uint64_t* lengths, char* bytes, uint64_t bytes_cap, uint64_t* offsets) {
  if (!s) return fail(XSG_EINVAL, "shard is null");
  if (s->last_mode != XSG_LINES) return fail(XSG_ESTATE, "no XSG_LINES result is pending on this shard");
  if (bytes_cap < s->line_bytes) return fail(XSG_EINVAL, "bytes_cap too small");
  if (s->fast_result) {  // the one-sync route: lengths, offsets and bytes are in the pinned mirrors
    if (s->line_bytes) {
      if (!bytes) return fail(XSG_EINVAL, "bytes is null");
      memcpy(bytes, s->hp_line_bytes, s->line_bytes);
    }
    const uint64_t* goff = static_cast<const uint64_t*>(s->h_result);
    uint64_t k = 0;
    for (uint64_t i = 0; i < s->fast_raw_lines; ++i) {
      if (s->hp_line_len[i] == UINT64_MAX) continue;
      if (lengths) lengths[k] = s->hp_line_len[i];
      if (offsets) offsets[k] = goff[i];
      ++k;
    }
    return XSG_OK;
  }
  xsg_ctx* c = s->ctx;
  HIP_TRY(hipSetDevice(c->device));
  XSG_TRY(fetch_line_lengths(s));
  const uint64_t raw = s->fast_raw_lines;
  std::vector<uint64_t> goff;
  if (offsets && raw) {
    goff.resize(raw);
    HIP_TRY(hipMemcpyAsync(goff.data(), s->d_out_u64.p, 8 * raw, hipMemcpyDeviceToHost, c->stream));
  }
  if (s->line_bytes) {
    if (!bytes) return fail(XSG_EINVAL, "bytes is null");
    HIP_TRY(hipMemcpyAsync(bytes, s->d_line_bytes.p, s->line_bytes, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  // dropped (unterminated) lines own no bytes, so the packed bytes are already contiguous
  uint64_t k = 0;
  for (uint64_t i = 0; i < raw; ++i) {
    if (s->hp_line_len[i] == UINT64_MAX) continue;
    if (lengths) lengths[k] = s->hp_line_len[i];
    if (offsets) offsets[k] = goff[i];
    ++k;
  }
  return XSG_OK;
}

// xs::lines without a copy into caller memory: lengths, offsets and packed bytes in the shard's pinned buffers.
extern "C" int xsg_result_lines_view(xsg_shard* s, const uint64_t** lengths, const char** bytes, const uint64_t** offsets,
                                     uint64_t* n_lines, uint64_t* total_bytes) {
  if (!s) return fail(XSG_EINVAL, "shard is null");
  if (s->last_mode != XSG_LINES) return fail(XSG_ESTATE, "no XSG_LINES result is pending on this shard");
  xsg_ctx* c = s->ctx;
  uint64_t raw = s->fast_raw_lines;
  XSG_TRY(fetch_line_lengths(s));
  if (!s->fast_result) {  // the exact route left bytes and offsets on the device as well: two more pinned copies
    if (8 * raw * 2 + s->line_bytes > (16ull << 30)) return fail(XSG_ENOMEM, "the result needs more than 16 GiB of pinned memory");
    HIP_TRY(hipSetDevice(c->device));
    XSG_TRY(ensure_pinned(&s->hp_line_bytes, &s->hp_line_bytes_cap, (size_t)s->line_bytes));
    {
      size_t have = s->h_result_cap / 8;
      uint64_t* hp = static_cast<uint64_t*>(s->h_result);
      const int pr = ensure_pinned(&hp, &have, (size_t)raw);
      s->h_result = hp;
      s->h_result_cap = have * 8;
      XSG_TRY(pr);
    }
    if (raw) HIP_TRY(hipMemcpyAsync(s->h_result, s->d_out_u64.p, 8 * raw, hipMemcpyDeviceToHost, c->stream));
    if (s->line_bytes) HIP_TRY(hipMemcpyAsync(s->hp_line_bytes, s->d_line_bytes.p, s->line_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    s->fast_result = true;  // from here on the result lives in the pinned buffers (xsg_result_lines reads them too)
  }
  if (s->total != raw) {  // lines without a terminating '\n' are not reported: squeeze them out (they own no bytes)
    uint64_t* len = s->hp_line_len;
    uint64_t* off = static_cast<uint64_t*>(s->h_result);
    uint64_t k = 0;
    for (uint64_t i = 0; i < raw; ++i) {
      if (len[i] == UINT64_MAX) continue;
      len[k] = len[i];
      off[k] = off[i];
      ++k;
    }
    s->fast_raw_lines = k;
  }
  if (lengths) *lengths = s->hp_line_len;
  if (offsets) *offsets = static_cast<const uint64_t*>(s->h_result);
  if (bytes) *bytes = reinterpret_cast<const char*>(s->hp_line_bytes);
  if (n_lines) *n_lines = s->total;
  if (total_bytes) *total_bytes = s->line_bytes;
  return XSG_OK;
}
