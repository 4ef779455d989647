// device_double.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A stand-in for the GPU below x-search_amd/csrc/xsg_file.cpp so that the HOST pipeline -- reader threads, the pinned
// buffer queue, device workers, ordered publication, the blocking result cursor (what replaces the reference's
// include/xsearch/Searcher.h:100-120 and ResultTypes.h:48-60) -- runs under ThreadSanitizer and AddressSanitizer on a
// box without a GPU (the reference keeps such builds: Makefile:30-38).  It provides
//   * the handful of HIP runtime entry points xsg_file.cpp calls, over plain host memory ("device" = malloc,
//     hipMemcpyAsync = memcpy, streams are synchronous), and
//   * the chunk-level C ABI of xsg_api.cpp (xsg_ctx_*, xsg_shard_*, xsg_count*, xsg_search, xsg_result_*), answered by
//     the CPU oracle (oracle/xs_oracle.c, linked from tests/ only).
// Nothing here is a CPU search path of the product: the product library has none, and this file is compiled only
// into tests/cpp/build/pipeline_{tsan,asan}.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>
#include <vector>

#include "../../x-search_amd/csrc/xsg_objects.h"

extern "C" {
// oracle/xs_oracle.c
uint64_t xso_count(const char* data, size_t len, const char* pat, size_t plen, int skip_to_nl);
uint64_t xso_byte_offsets_match(const char* data, size_t len, const char* pat, size_t plen, int skip_to_nl, uint64_t* out,
                                uint64_t cap);
uint64_t xso_byte_offsets_line(const char* data, size_t len, const char* pat, size_t plen, uint64_t* out, uint64_t cap);
uint64_t xso_lines(const char* data, size_t len, const char* pat, size_t plen, uint64_t* begin, uint64_t* length, uint64_t cap);
uint64_t xso_count_newlines(const char* data, size_t len);
uint64_t xso_line_indices(const char* data, size_t len, const char* pat, size_t plen, uint64_t line_base, uint64_t* out,
                          uint64_t cap);
}

// ---- errors (xsg_api.cpp's) -------------------------------------------------------------------------------------
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
bool trace_on() { return false; }
bool test_hooks() { return true; }
void trace(const char*, ...) {}
}  // namespace xsg
using xsg::fail;
extern "C" const char* xsg_last_error(void) { return g_err; }
extern "C" const char* xsg_strerror(int code) { return code == XSG_OK ? "ok" : "error"; }
extern "C" int xsg_abi_version(void) { return XSG_ABI_VERSION; }

// ---- HIP runtime, over host memory --------------------------------------------------------------------------------
static std::atomic<long> g_live_allocs{0};
extern "C" long xsg_double_live_allocations(void) { return g_live_allocs.load(); }
extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) {
  *p = malloc(n ? n : 1);
  if (*p) ++g_live_allocs;
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
  if (p) --g_live_allocs;
  free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) {
  *p = malloc(n ? n : 1);
  if (*p) ++g_live_allocs;
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) {
  if (p) --g_live_allocs;
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) {
  memcpy(dst, src, n);  // synchronous: the strictest reading of "the copy may read the pinned buffer until the stream says so"
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char* s, int n, int) {
  snprintf(s, (size_t)n, "0000:00:00.0");
  return hipSuccess;
}
const char* hipGetErrorString(hipError_t) { return "double"; }
}

// ---- chunk-level C ABI, answered by the oracle ----------------------------------------------------------------------
extern "C" int xsg_device_count(int* n) {
  if (!n) return fail(XSG_EINVAL, "null");
  *n = 2;  // two "devices": the XS_DEVICES fan-out has something to fan out over
  return XSG_OK;
}
extern "C" int xsg_ctx_create(int device, xsg_ctx** out) {
  if (!out) return fail(XSG_EINVAL, "null");
  if (device < 0 || device >= 2) return fail(XSG_ENODEV, "device %d out of range", device);
  xsg_ctx* c = new xsg_ctx();
  c->device = device;
  *out = c;
  return XSG_OK;
}
extern "C" void xsg_ctx_destroy(xsg_ctx* c) { delete c; }
extern "C" int xsg_set_pattern(xsg_ctx* c, const void* p, size_t n, uint32_t flags) {
  if (!c || !p || !n) return fail(XSG_EINVAL, "bad pattern");
  if (flags & (XSG_FLAG_REGEX | XSG_FLAG_IGNORE_CASE | XSG_FLAG_EXACT_TAIL)) return fail(XSG_ENOTSUP, "the double serves plain literals");
  c->pattern.assign(static_cast<const uint8_t*>(p), static_cast<const uint8_t*>(p) + n);
  c->flags = flags;
  c->pat.has_newline = memchr(p, '\n', n) != nullptr;
  return XSG_OK;
}
extern "C" int xsg_regex_check(const void*, size_t, uint32_t, uint32_t*, uint32_t*) { return fail(XSG_ENOTSUP, "no regex in the double"); }
extern "C" int xsg_regex_dfa_info(const void*, size_t, uint32_t, xsg_regex_dfa*, uint16_t*, uint16_t*, size_t) {
  return fail(XSG_ENOTSUP, "no regex in the double");
}

// results of the last list search live here (the product keeps them in the shard's pinned buffers, xsg_objects.h)
struct DoubleResult {
  std::vector<uint64_t> u64;
  std::vector<uint64_t> line_begin, line_len;
  std::vector<char> line_bytes;
};
static DoubleResult* res_of(xsg_shard* s) { return static_cast<DoubleResult*>(s->h_result); }

static int bind(xsg_shard* s, const void* base, uint64_t cap, const xsg_chunk* chunks, uint64_t n) {
  s->base = static_cast<const uint8_t*>(base);
  s->capacity = cap;
  s->chunks.assign(chunks, chunks + n);
  s->total_bytes = 0;
  for (uint64_t i = 0; i < n; ++i) {
    if (chunks[i].offset + chunks[i].length > cap) return fail(XSG_EINVAL, "chunk beyond capacity");
    s->total_bytes += chunks[i].length;
  }
  s->last_mode = -1;
  return XSG_OK;
}
extern "C" int xsg_shard_create(xsg_ctx* c, const void* base, uint64_t cap, const xsg_chunk* chunks, uint64_t n, xsg_shard** out) {
  if (!c || !out) return fail(XSG_EINVAL, "null");
  xsg_shard* s = new xsg_shard();
  s->ctx = c;
  s->h_result = new DoubleResult();
  const int r = bind(s, base, cap, chunks, n);
  if (r != XSG_OK) {
    delete res_of(s);
    delete s;
    return r;
  }
  *out = s;
  return XSG_OK;
}
extern "C" int xsg_shard_rebind(xsg_shard* s, const void* base, uint64_t cap, const xsg_chunk* chunks, uint64_t n) {
  if (!s) return fail(XSG_EINVAL, "null");
  return bind(s, base, cap, chunks, n);
}
extern "C" void xsg_shard_destroy(xsg_shard* s) {
  if (!s) return;
  delete res_of(s);
  s->h_result = nullptr;
  delete s;
}

static const char* pat(xsg_shard* s) { return reinterpret_cast<const char*>(s->ctx->pattern.data()); }
static size_t plen(xsg_shard* s) { return s->ctx->pattern.size(); }

extern "C" int xsg_count(xsg_shard* s, uint32_t mode, uint64_t ctr[XSG_NUM_COUNTERS]) {
  if (!s || !ctr) return fail(XSG_EINVAL, "null");
  if (s->ctx->pattern.empty()) return fail(XSG_ESTATE, "no pattern");
  const uint32_t m = mode & 0xffu;
  if (m != XSG_COUNT_MATCHES && m != XSG_COUNT_LINES) return fail(XSG_EINVAL, "not a count mode");
  memset(ctr, 0, 8 * XSG_NUM_COUNTERS);
  for (const xsg_chunk& ch : s->chunks) {
    const char* d = reinterpret_cast<const char*>(s->base + ch.offset);
    ctr[m == XSG_COUNT_MATCHES ? XSG_CTR_MATCHES : XSG_CTR_LINES] += xso_count(d, ch.length, pat(s), plen(s), m == XSG_COUNT_LINES);
    if (mode & XSG_WITH_NEWLINES) ctr[XSG_CTR_NEWLINES] += xso_count_newlines(d, ch.length);
    ctr[XSG_CTR_BYTES] += ch.length;
  }
  return XSG_OK;
}
// split-phase: the work is done in _begin (the double has no queue), _end hands it out
extern "C" int xsg_count_begin(xsg_shard* s, uint32_t mode) { return xsg_count(s, mode, s->begin_counters); }
extern "C" int xsg_count_end(xsg_shard* s, uint64_t ctr[XSG_NUM_COUNTERS]) {
  if (!s || !ctr) return fail(XSG_EINVAL, "null");
  memcpy(ctr, s->begin_counters, 8 * XSG_NUM_COUNTERS);
  return XSG_OK;
}

extern "C" int xsg_search(xsg_shard* s, uint32_t mode, uint64_t* n_results) {
  if (!s) return fail(XSG_EINVAL, "null");
  if (s->ctx->pattern.empty()) return fail(XSG_ESTATE, "no pattern");
  DoubleResult& r = *res_of(s);
  r.u64.clear();
  r.line_begin.clear();
  r.line_len.clear();
  r.line_bytes.clear();
  uint64_t nl_before = 0;
  for (const xsg_chunk& ch : s->chunks) {
    const char* d = reinterpret_cast<const char*>(s->base + ch.offset);
    const uint64_t cap = ch.length / (plen(s) ? plen(s) : 1) + 2;
    std::vector<uint64_t> a(cap), b(cap);
    uint64_t n = 0;
    switch (mode) {
      case XSG_MATCH_BYTE_OFFSETS:
        n = xso_byte_offsets_match(d, ch.length, pat(s), plen(s), 0, a.data(), cap);
        for (uint64_t i = 0; i < n; ++i) r.u64.push_back(a[i] + ch.global_offset);
        break;
      case XSG_LINE_BYTE_OFFSETS:
        n = xso_byte_offsets_line(d, ch.length, pat(s), plen(s), a.data(), cap);
        for (uint64_t i = 0; i < n; ++i) r.u64.push_back(a[i] + ch.global_offset);
        break;
      case XSG_LINE_INDICES: {
        const uint64_t base = ch.line_base == XSG_LINE_BASE_AUTO ? s->shard_line_base + nl_before : ch.line_base;
        n = xso_line_indices(d, ch.length, pat(s), plen(s), base, a.data(), cap);
        for (uint64_t i = 0; i < n; ++i) r.u64.push_back(a[i]);
        break;
      }
      case XSG_LINES:
        n = xso_lines(d, ch.length, pat(s), plen(s), a.data(), b.data(), cap);
        for (uint64_t i = 0; i < n; ++i) {
          r.line_begin.push_back(a[i] + ch.global_offset);
          r.line_len.push_back(b[i]);
          r.line_bytes.insert(r.line_bytes.end(), d + a[i], d + a[i] + b[i]);
        }
        break;
      default:
        return fail(XSG_EINVAL, "not a list mode");
    }
    nl_before += xso_count_newlines(d, ch.length);
  }
  s->last_newlines = nl_before;
  s->last_mode = (int)mode;
  s->total = mode == XSG_LINES ? r.line_len.size() : r.u64.size();
  if (n_results) *n_results = s->total;
  return XSG_OK;
}
extern "C" int xsg_result_u64(xsg_shard* s, uint64_t* out, uint64_t cap) {
  if (!s || s->last_mode < XSG_MATCH_BYTE_OFFSETS || s->last_mode > XSG_LINE_INDICES) return fail(XSG_ESTATE, "no list result");
  if (cap < s->total) return fail(XSG_EINVAL, "capacity");
  if (s->total) memcpy(out, res_of(s)->u64.data(), 8 * s->total);
  return XSG_OK;
}
extern "C" int xsg_result_u64_view(xsg_shard* s, const uint64_t** out, uint64_t* n) {
  if (!s || !out || !n) return fail(XSG_EINVAL, "null");
  if (s->last_mode < XSG_MATCH_BYTE_OFFSETS || s->last_mode > XSG_LINE_INDICES) return fail(XSG_ESTATE, "no list result");
  *out = res_of(s)->u64.data();
  *n = s->total;
  return XSG_OK;
}
extern "C" int xsg_result_newlines(xsg_shard* s, uint64_t* nl) {
  if (!s || !nl || s->last_mode != XSG_LINE_INDICES) return fail(XSG_ESTATE, "no line-index result");
  *nl = s->last_newlines;
  return XSG_OK;
}
extern "C" int xsg_result_lines_size(xsg_shard* s, uint64_t* n, uint64_t* bytes) {
  if (!s || s->last_mode != XSG_LINES) return fail(XSG_ESTATE, "no lines result");
  if (n) *n = res_of(s)->line_len.size();
  if (bytes) *bytes = res_of(s)->line_bytes.size();
  return XSG_OK;
}
extern "C" int xsg_result_lines(xsg_shard* s, uint64_t* lengths, char* bytes, uint64_t bytes_cap, uint64_t* offsets) {
  if (!s || s->last_mode != XSG_LINES) return fail(XSG_ESTATE, "no lines result");
  DoubleResult& r = *res_of(s);
  if (bytes_cap < r.line_bytes.size()) return fail(XSG_EINVAL, "capacity");
  if (!r.line_bytes.empty()) memcpy(bytes, r.line_bytes.data(), r.line_bytes.size());
  for (size_t i = 0; i < r.line_len.size(); ++i) {
    if (lengths) lengths[i] = r.line_len[i];
    if (offsets) offsets[i] = r.line_begin[i];
  }
  return XSG_OK;
}
extern "C" int xsg_result_lines_view(xsg_shard* s, const uint64_t** lengths, const char** bytes, const uint64_t** offsets,
                                     uint64_t* n, uint64_t* total_bytes) {
  if (!s || s->last_mode != XSG_LINES) return fail(XSG_ESTATE, "no lines result");
  DoubleResult& r = *res_of(s);
  if (lengths) *lengths = r.line_len.data();
  if (bytes) *bytes = reinterpret_cast<const char*>(r.line_bytes.data());
  if (offsets) *offsets = r.line_begin.data();
  if (n) *n = r.line_len.size();
  if (total_bytes) *total_bytes = r.line_bytes.size();
  return XSG_OK;
}
// no RCCL in the double: the fan-out adds on the host (and says so)
extern "C" int xsg_comm_create_local(xsg_ctx* const*, int, xsg_comm**) { return fail(XSG_ENOTSUP, "no librccl in the double"); }
extern "C" int xsg_reduce_counts(xsg_comm*, uint64_t* const*, int, uint64_t*) { return fail(XSG_ENOTSUP, "no librccl in the double"); }
extern "C" void xsg_comm_destroy(xsg_comm*) {}
