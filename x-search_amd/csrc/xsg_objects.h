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
}  // namespace xsg

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
  int ensure(size_t bytes) {
    if (bytes <= cap && p) return XSG_OK;
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
  xsg::PatternDev pat{};
  DevBuf d_pat;
  uint32_t tile_bytes = xsg::kDefaultTileBytes;  // geometry new shards get (XSG_TILE_KIB)
  uint32_t tune = xsg::kTuneAuto;                  // wave stagger: per kernel variant (XSG_TUNE overrides)
  char arch[128] = "";
  int cus = 0;
  uint64_t hbm = 0;
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

  DevBuf d_chunks, d_tile_chunk, d_chunk_tile0;
  DevBuf d_tile_cnt, d_tile_nl, d_tile_sum, d_tile_last, d_counters;
  DevBuf d_tile_off, d_tile_nl_off, d_scan_tmp;
  DevBuf d_m_pos, d_m_chunk, d_m_ls, d_keep, d_keep_pre;
  DevBuf d_chunk_shift0, d_tail_cnt, d_tail_pos, d_tail_pre;
  DevBuf d_f_pos, d_f_match, d_f_chunk, d_out_u64, d_line_len, d_line_off, d_line_bytes;

  int last_mode = -1;
  uint64_t total = 0;       // elements of the last list search
  uint64_t line_bytes = 0;  // XSG_LINES: packed bytes
  uint64_t last_newlines = 0;  // XSG_LINE_INDICES: '\n' in the shard (for chaining line bases)
  std::vector<uint64_t> h_line_len, h_line_off;

  void release_all() {
    DevBuf* all[] = {&d_chunks, &d_tile_chunk, &d_chunk_tile0, &d_tile_cnt, &d_tile_nl, &d_tile_sum, &d_tile_last,
                     &d_counters, &d_tile_off, &d_tile_nl_off, &d_scan_tmp, &d_m_pos, &d_m_chunk, &d_m_ls, &d_keep,
                     &d_keep_pre, &d_chunk_shift0, &d_tail_cnt, &d_tail_pos, &d_tail_pre, &d_f_pos, &d_f_match,
                     &d_f_chunk, &d_out_u64, &d_line_len, &d_line_off, &d_line_bytes};
    for (DevBuf* b : all) b->release();
  }
};

