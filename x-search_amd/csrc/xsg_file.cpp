// xsg_file.cpp -- the host pipeline behind xs::extern_search: chunk plans,
// metafiles, reader/feeder threads, ordered result store.  C++ on the host (the
// reference's pipeline is C++: include/xsearch/Searcher.h, tasks/readers.h,
// ResultTypes.h); the scan itself always runs on the GPU through the shard API
// of xsg_api.cpp -- there is no CPU search path in here.
#include <ctype.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "xsg_lz4.h"
#include "xsg_objects.h"

using namespace xsg;
using Clock = std::chrono::steady_clock;

static double seconds_since(Clock::time_point t0) {
  return std::chrono::duration<double>(Clock::now() - t0).count();
}

extern "C" void xsg_free(void* p) { free(p); }

// No C++ exception may cross the C ABI: host-side containers sized from file contents can throw.
template <typename F>
static int guarded(const char* what, F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(XSG_ENOMEM, "%s: host allocation failed", what);
  } catch (const std::exception& e) {
    return fail(XSG_EIO, "%s: %s", what, e.what());
  }
}

// ---------------------------------------------------------------------------
// file helpers
// ---------------------------------------------------------------------------
static int pread_full(int fd, void* buf, uint64_t n, uint64_t off) {
  uint8_t* p = static_cast<uint8_t*>(buf);
  while (n) {
    const ssize_t r = pread(fd, p, n > (1u << 30) ? (1u << 30) : n, (off_t)off);
    if (r < 0) {
      if (errno == EINTR) continue;
      return fail(XSG_EIO, "pread failed: %s", strerror(errno));
    }
    if (r == 0) return fail(XSG_EIO, "unexpected end of file at offset %llu", (unsigned long long)off);
    p += r;
    off += (uint64_t)r;
    n -= (uint64_t)r;
  }
  return XSG_OK;
}

static int open_ro(const char* path, int* fd, uint64_t* size) {
  if (!path || !*path) return fail(XSG_EINVAL, "empty file path");
  const int f = open(path, O_RDONLY | O_CLOEXEC);
  if (f < 0) return fail(XSG_EIO, "cannot open '%s': %s", path, strerror(errno));
  struct stat st;
  if (fstat(f, &st) != 0) {
    close(f);
    return fail(XSG_EIO, "cannot stat '%s': %s", path, strerror(errno));
  }
  if (!S_ISREG(st.st_mode)) {
    close(f);
    return fail(XSG_EIO, "'%s' is not a regular file", path);
  }
  *fd = f;
  *size = (uint64_t)st.st_size;
  return XSG_OK;
}

// ---------------------------------------------------------------------------
// chunk plan of a plain file: >= target bytes, extended to just past the next '\n'
// (the layout of the reference's fixtures: SURVEY 5.1)
// ---------------------------------------------------------------------------
static int plan_plain(int fd, uint64_t size, uint64_t target, std::vector<xsg_file_chunk>& out) {
  if (target < 1) target = 1;
  uint64_t pos = 0;
  std::vector<uint8_t> probe(1 << 16);
  while (pos < size) {
    uint64_t end = pos + target;
    if (end >= size) {
      end = size;
    } else {
      // first '\n' at or after end-1
      uint64_t q = end - 1;
      bool found = false;
      while (q < size) {
        const uint64_t n = std::min<uint64_t>(probe.size(), size - q);
        XSG_TRY(pread_full(fd, probe.data(), n, q));
        const void* hit = memchr(probe.data(), '\n', n);
        if (hit) {
          end = q + (uint64_t)((const uint8_t*)hit - probe.data()) + 1;
          found = true;
          break;
        }
        q += n;
      }
      if (!found) end = size;
    }
    xsg_file_chunk c{};
    c.original_offset = pos;
    c.actual_offset = pos;
    c.original_size = end - pos;
    c.actual_size = end - pos;
    c.first_line = XSG_LINE_BASE_AUTO;
    c.n_mappings = 0;
    out.push_back(c);
    pos = end;
  }
  return XSG_OK;
}

static int to_malloc(const std::vector<xsg_file_chunk>& v, xsg_file_chunk** chunks, uint64_t* n) {
  *n = v.size();
  *chunks = static_cast<xsg_file_chunk*>(malloc(sizeof(xsg_file_chunk) * std::max<size_t>(v.size(), 1)));
  if (!*chunks) return fail(XSG_ENOMEM, "host allocation failed");
  if (!v.empty()) memcpy(*chunks, v.data(), sizeof(xsg_file_chunk) * v.size());
  return XSG_OK;
}

extern "C" int xsg_plan_chunks(const char* file_path, uint64_t target_bytes, xsg_file_chunk** chunks, uint64_t* n) {
  if (!chunks || !n) return fail(XSG_EINVAL, "null output");
  int fd;
  uint64_t size;
  XSG_TRY(open_ro(file_path, &fd, &size));
  return guarded("xsg_plan_chunks", [&]() -> int {
    std::vector<xsg_file_chunk> v;
    int r;
    try {
      r = plan_plain(fd, size, target_bytes ? target_bytes : (16u << 20), v);
    } catch (...) {
      close(fd);
      throw;
    }
    close(fd);
    if (r != XSG_OK) return r;
    return to_malloc(v, chunks, n);
  });
}

// ---------------------------------------------------------------------------
// metafile (little-endian, packed):
//   int32 compression_type, then per chunk
//   u64 original_offset, actual_offset, original_size, actual_size, n, n x {u64 byte offset, u64 line index}
// ---------------------------------------------------------------------------
static int read_meta(const char* path, int32_t* compression, std::vector<xsg_file_chunk>& chunks,
                     std::vector<uint64_t>* mappings) {
  // The mapping tables are ~3 % of the corpus size (one entry per >= 500 bytes): a
  // search only needs the 40-byte record headers and each chunk's first entry, so
  // the file is walked with small preads and the tables are read only on request.
  int fd;
  uint64_t size;
  XSG_TRY(open_ro(path, &fd, &size));
  struct Closer {
    int fd;
    ~Closer() { close(fd); }
  } closer{fd};
  if (size < 4) return fail(XSG_EIO, "metafile '%s' is too short", path);
  int32_t ct;
  XSG_TRY(pread_full(fd, &ct, 4, 0));
  if (ct != XSG_COMPRESSION_NONE && ct != XSG_COMPRESSION_ZSTD && ct != XSG_COMPRESSION_LZ4)
    return fail(XSG_EIO, "metafile '%s': unknown compression type %d", path, ct);
  *compression = ct;
  uint64_t pos = 4;
  uint64_t expect_orig = 0;
  while (pos < size) {
    if (size - pos < 40) return fail(XSG_EIO, "metafile '%s': truncated chunk record at byte %llu", path,
                                     (unsigned long long)pos);
    uint64_t f[7] = {0, 0, 0, 0, 0, 0, 0};  // 5 header words + the first mapping entry
    const uint64_t want = std::min<uint64_t>(56, size - pos);
    XSG_TRY(pread_full(fd, f, want, pos));
    pos += 40;
    const uint64_t n = f[4];
    if (n > (size - pos) / 16) return fail(XSG_EIO, "metafile '%s': mapping table of chunk %zu overruns the file", path,
                                           chunks.size());
    xsg_file_chunk c{};
    c.original_offset = f[0];
    c.actual_offset = f[1];
    c.original_size = f[2];
    c.actual_size = f[3];
    c.n_mappings = n;
    c.first_line = XSG_LINE_BASE_AUTO;
    if (c.original_offset != expect_orig)
      return fail(XSG_EIO, "metafile '%s': chunk %zu does not start where its predecessor ends", path, chunks.size());
    if (c.original_size >= (1ull << 40) || c.actual_size >= (1ull << 40))
      return fail(XSG_EIO, "metafile '%s': chunk %zu has an implausible size", path, chunks.size());
    expect_orig += c.original_size;  // < 2^40 each and at most size/40 records: no wrap
    if (n) {
      // the first mapping entry of a chunk is the chunk start
      if (f[5] == c.original_offset) c.first_line = f[6];
      if (mappings) {
        const size_t at = mappings->size();
        mappings->resize(at + 2 * n);
        XSG_TRY(pread_full(fd, mappings->data() + at, 16 * n, pos));
      }
    }
    pos += 16 * n;
    chunks.push_back(c);
  }
  return XSG_OK;
}

extern "C" int xsg_meta_read(const char* meta_path, int32_t* compression, xsg_file_chunk** chunks, uint64_t* n,
                             uint64_t** mappings, uint64_t* n_mapping_pairs) {
  if (!compression || !chunks || !n) return fail(XSG_EINVAL, "null output");
  return guarded("xsg_meta_read", [&]() -> int {
    std::vector<xsg_file_chunk> v;
    std::vector<uint64_t> maps;
    XSG_TRY(read_meta(meta_path, compression, v, mappings ? &maps : nullptr));
    XSG_TRY(to_malloc(v, chunks, n));
    if (mappings) {
      *mappings = static_cast<uint64_t*>(malloc(8 * std::max<size_t>(maps.size(), 1)));
      if (!*mappings) return fail(XSG_ENOMEM, "host allocation failed");
      if (!maps.empty()) memcpy(*mappings, maps.data(), 8 * maps.size());
      if (n_mapping_pairs) *n_mapping_pairs = maps.size() / 2;
    }
    return XSG_OK;
  });
}

// ---------------------------------------------------------------------------
// LZ4 / ZSTD through the system libraries (the reference links liblz4 / libzstd,
// Dockerfile:8).  Loaded on demand so that plain-text searches need neither.
// ---------------------------------------------------------------------------
struct Codecs {
  void* lz4 = nullptr;
  void* zstd = nullptr;
  int (*LZ4_decompress_safe)(const char*, char*, int, int) = nullptr;
  int (*LZ4_compress_default)(const char*, char*, int, int) = nullptr;
  int (*LZ4_compress_HC)(const char*, char*, int, int, int) = nullptr;
  int (*LZ4_compressBound)(int) = nullptr;
  size_t (*ZSTD_decompress)(void*, size_t, const void*, size_t) = nullptr;
  size_t (*ZSTD_compress)(void*, size_t, const void*, size_t, int) = nullptr;
  size_t (*ZSTD_compressBound)(size_t) = nullptr;
  unsigned (*ZSTD_isError)(size_t) = nullptr;
};

static Codecs& codecs() {
  static Codecs c;
  return c;
}
static std::mutex g_codec_mu;

static void* open_any(const char* const* names) {
  for (; *names; ++names) {
    void* h = dlopen(*names, RTLD_NOW | RTLD_LOCAL);
    if (h) return h;
  }
  return nullptr;
}

// the built-in block codec (xsg_lz4.h) behind liblz4's signatures
static int builtin_lz4_decompress_safe(const char* src, char* dst, int src_n, int dst_cap) {
  if (src_n < 0 || dst_cap < 0) return -1;
  const int64_t r = xsg::lz4_block_decode((const uint8_t*)src, (size_t)src_n, (uint8_t*)dst, (size_t)dst_cap);
  return r < 0 ? -1 : (int)r;
}
static int builtin_lz4_compress_default(const char* src, char* dst, int src_n, int dst_cap) {
  return xsg::lz4_block_encode((const uint8_t*)src, src_n, (uint8_t*)dst, dst_cap);
}
static int builtin_lz4_compress_bound(int n) { return xsg::lz4_compress_bound(n); }
static char g_builtin_lz4_token;  // non-null marker for Codecs::lz4 when the built-in codec is in use

// liblz4 if the host has one, else the built-in codec (XSG_NO_LIBLZ4=1 forces the built-in one: tests)
static int need_lz4() {
  std::lock_guard<std::mutex> g(g_codec_mu);
  Codecs& c = codecs();
  if (c.lz4) return XSG_OK;
  static const char* names[] = {"liblz4.so.1", "liblz4.so", "/opt/conda/lib/liblz4.so.1", nullptr};
  const char* no = getenv("XSG_NO_LIBLZ4");
  void* h = (no && *no && *no != '0') ? nullptr : open_any(names);
  if (h) {
    c.LZ4_decompress_safe = (int (*)(const char*, char*, int, int))dlsym(h, "LZ4_decompress_safe");
    c.LZ4_compress_default = (int (*)(const char*, char*, int, int))dlsym(h, "LZ4_compress_default");
    c.LZ4_compress_HC = (int (*)(const char*, char*, int, int, int))dlsym(h, "LZ4_compress_HC");
    c.LZ4_compressBound = (int (*)(int))dlsym(h, "LZ4_compressBound");
    if (c.LZ4_decompress_safe && c.LZ4_compress_default && c.LZ4_compressBound) {
      c.lz4 = h;
      return XSG_OK;
    }
  }
  c.LZ4_decompress_safe = builtin_lz4_decompress_safe;
  c.LZ4_compress_default = builtin_lz4_compress_default;
  c.LZ4_compress_HC = nullptr;
  c.LZ4_compressBound = builtin_lz4_compress_bound;
  c.lz4 = &g_builtin_lz4_token;
  return XSG_OK;
}

static int need_zstd() {
  std::lock_guard<std::mutex> g(g_codec_mu);
  Codecs& c = codecs();
  if (c.zstd) return XSG_OK;
  static const char* names[] = {"libzstd.so.1", "libzstd.so", "/opt/conda/lib/libzstd.so.1", nullptr};
  void* h = open_any(names);
  if (!h) return fail(XSG_ENOTSUP, "libzstd not found on this host (needed for ZSTD metafiles)");
  c.ZSTD_decompress = (size_t(*)(void*, size_t, const void*, size_t))dlsym(h, "ZSTD_decompress");
  c.ZSTD_compress = (size_t(*)(void*, size_t, const void*, size_t, int))dlsym(h, "ZSTD_compress");
  c.ZSTD_compressBound = (size_t(*)(size_t))dlsym(h, "ZSTD_compressBound");
  c.ZSTD_isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
  if (!c.ZSTD_decompress || !c.ZSTD_compress || !c.ZSTD_compressBound || !c.ZSTD_isError)
    return fail(XSG_ENOTSUP, "libzstd lacks the expected symbols");
  c.zstd = h;
  return XSG_OK;
}

extern "C" const char* xsg_codec_name(int32_t compression) {
  if (compression == XSG_COMPRESSION_NONE) return "none";
  if (compression == XSG_COMPRESSION_LZ4) {
    if (need_lz4() != XSG_OK) return "";
    return codecs().lz4 == &g_builtin_lz4_token ? "built-in LZ4 block codec" : "liblz4";
  }
  if (compression == XSG_COMPRESSION_ZSTD) return need_zstd() == XSG_OK ? "libzstd" : "";
  return "";
}

static int decompress_chunk(int32_t type, const uint8_t* src, uint64_t src_n, uint8_t* dst, uint64_t dst_n) {
  if (type == XSG_COMPRESSION_LZ4) {
    // raw LZ4 block per chunk, decoded to exactly original_size bytes
    if (src_n > INT32_MAX || dst_n > INT32_MAX) return fail(XSG_EIO, "LZ4 chunk too large");
    const int r = codecs().LZ4_decompress_safe((const char*)src, (char*)dst, (int)src_n, (int)dst_n);
    if (r < 0 || (uint64_t)r != dst_n) return fail(XSG_EIO, "LZ4 chunk does not decode to its recorded size");
    return XSG_OK;
  }
  if (type == XSG_COMPRESSION_ZSTD) {
    const size_t r = codecs().ZSTD_decompress(dst, dst_n, src, src_n);
    if (codecs().ZSTD_isError(r) || r != dst_n) return fail(XSG_EIO, "ZSTD chunk does not decode to its recorded size");
    return XSG_OK;
  }
  return fail(XSG_EINVAL, "bad compression type %d", type);
}

// ---------------------------------------------------------------------------
// metafile writer / preprocessor
// ---------------------------------------------------------------------------
static int meta_write_impl(const char* file_path, const char* meta_out_path, const char* data_out_path,
                           int32_t compression, uint64_t chunk_bytes, uint64_t mapping_gap, int hc);
extern "C" int xsg_meta_write(const char* file_path, const char* meta_out_path, const char* data_out_path,
                              int32_t compression, uint64_t chunk_bytes, uint64_t mapping_gap, int hc) {
  return guarded("xsg_meta_write", [&]() -> int {
    return meta_write_impl(file_path, meta_out_path, data_out_path, compression, chunk_bytes, mapping_gap, hc);
  });
}
static int meta_write_impl(const char* file_path, const char* meta_out_path, const char* data_out_path,
                           int32_t compression, uint64_t chunk_bytes, uint64_t mapping_gap, int hc) {
  if (!meta_out_path) return fail(XSG_EINVAL, "meta_out_path is null");
  if (compression != XSG_COMPRESSION_NONE && compression != XSG_COMPRESSION_ZSTD && compression != XSG_COMPRESSION_LZ4)
    return fail(XSG_EINVAL, "bad compression type %d", compression);
  if (compression != XSG_COMPRESSION_NONE && !data_out_path) return fail(XSG_EINVAL, "data_out_path is null");
  if (compression == XSG_COMPRESSION_LZ4) XSG_TRY(need_lz4());
  if (compression == XSG_COMPRESSION_ZSTD) XSG_TRY(need_zstd());
  if (!chunk_bytes) chunk_bytes = 16u << 20;
  if (!mapping_gap) mapping_gap = 500;
  int fd;
  uint64_t size;
  XSG_TRY(open_ro(file_path, &fd, &size));
  std::vector<xsg_file_chunk> plan;
  int r = plan_plain(fd, size, chunk_bytes, plan);
  if (r != XSG_OK) {
    close(fd);
    return r;
  }
  FILE* mf = fopen(meta_out_path, "wb");
  FILE* df = compression != XSG_COMPRESSION_NONE ? fopen(data_out_path, "wb") : nullptr;
  if (!mf || (compression != XSG_COMPRESSION_NONE && !df)) {
    if (mf) fclose(mf);
    if (df) fclose(df);
    close(fd);
    return fail(XSG_EIO, "cannot create output files: %s", strerror(errno));
  }
  fwrite(&compression, 4, 1, mf);
  std::vector<uint8_t> raw, packed;
  std::vector<uint64_t> maps;
  uint64_t line = 0, actual_off = 0;
  for (const xsg_file_chunk& pc : plan) {
    raw.resize(pc.original_size);
    r = pread_full(fd, raw.data(), pc.original_size, pc.original_offset);
    if (r != XSG_OK) break;
    // mapping: the chunk start, then the first line start >= previous entry + gap
    maps.clear();
    maps.push_back(pc.original_offset);
    maps.push_back(line);
    uint64_t last_entry = 0;
    for (uint64_t i = 0; i < pc.original_size; ++i) {
      if (raw[i] == '\n') {
        ++line;
        const uint64_t ls = i + 1;
        if (ls < pc.original_size && ls >= last_entry + mapping_gap) {
          maps.push_back(pc.original_offset + ls);
          maps.push_back(line);
          last_entry = ls;
        }
      }
    }
    uint64_t actual_size = pc.original_size;
    if (compression == XSG_COMPRESSION_LZ4) {
      if (pc.original_size > INT32_MAX) {
        r = fail(XSG_EINVAL, "chunk too large for an LZ4 block");
        break;
      }
      packed.resize((size_t)codecs().LZ4_compressBound((int)pc.original_size));
      const int n = hc && codecs().LZ4_compress_HC
                        ? codecs().LZ4_compress_HC((const char*)raw.data(), (char*)packed.data(), (int)raw.size(),
                                                   (int)packed.size(), 9)
                        : codecs().LZ4_compress_default((const char*)raw.data(), (char*)packed.data(), (int)raw.size(),
                                                        (int)packed.size());
      if (n <= 0 && pc.original_size) {
        r = fail(XSG_EIO, "LZ4 compression failed");
        break;
      }
      actual_size = (uint64_t)n;
    } else if (compression == XSG_COMPRESSION_ZSTD) {
      packed.resize(codecs().ZSTD_compressBound(raw.size()));
      const size_t n = codecs().ZSTD_compress(packed.data(), packed.size(), raw.data(), raw.size(), 3);
      if (codecs().ZSTD_isError(n)) {
        r = fail(XSG_EIO, "ZSTD compression failed");
        break;
      }
      actual_size = n;
    }
    if (df && actual_size && fwrite(packed.data(), 1, actual_size, df) != actual_size) {
      r = fail(XSG_EIO, "short write to '%s'", data_out_path);
      break;
    }
    const uint64_t rec[5] = {pc.original_offset, compression == XSG_COMPRESSION_NONE ? pc.original_offset : actual_off,
                             pc.original_size, actual_size, maps.size() / 2};
    fwrite(rec, 8, 5, mf);
    fwrite(maps.data(), 8, maps.size(), mf);
    actual_off += actual_size;
  }
  if (fclose(mf) != 0 && r == XSG_OK) r = fail(XSG_EIO, "cannot write '%s'", meta_out_path);
  if (df && fclose(df) != 0 && r == XSG_OK) r = fail(XSG_EIO, "cannot write '%s'", data_out_path);
  close(fd);
  return r;
}

// ---------------------------------------------------------------------------
// the job
// ---------------------------------------------------------------------------
struct Partial {
  uint64_t count = 0;     // count tags
  uint64_t newlines = 0;  // line indices without metafile bases
  bool indices_local = false;
  std::vector<uint64_t> u64;
  std::vector<std::string> lines;
};

static std::atomic<uint64_t> g_job_serial{0};

struct xsg_job {
  const uint64_t serial = ++g_job_serial;
  xsg_job_opts opts{};
  std::vector<uint8_t> pattern;
  int fd = -1;
  int32_t compression = XSG_COMPRESSION_NONE;
  std::vector<xsg_file_chunk> plan;
  uint64_t max_orig = 0, max_actual = 0;

  // Two stages.  num_max_readers reader threads (Searcher.h:39,106 num_concurrent_reads)
  // pread -- and decompress -- chunks into pinned buffers; num_threads device workers take
  // the filled buffers, hipMemcpyAsync them on their own stream and run the scan.  The
  // pinned buffers circulate: free -> (reader) -> ready -> (worker) -> free.
  std::vector<std::thread> threads;
  std::atomic<bool> stop{false};
  std::atomic<int> active{0};   // device workers still running
  int readers_running = 0;      // guarded by q_mu
  std::mutex q_mu;
  std::condition_variable q_cv_free, q_cv_ready;
  std::vector<struct HostBuf*> all_bufs, free_bufs;
  std::deque<struct HostBuf*> ready_bufs;
  // Reading is decoupled from the chunk plan: the readers fill ONE chunk at a time together, piece by piece (a plain
  // file's chunk of 16 MiB is read as pieces of 4 MiB by whichever readers are free), so that a file of six chunks keeps
  // eight readers busy and its first chunk is on its way to the device after a quarter of a chunk's read time.  A
  // compressed chunk is one piece (its decoder needs all of it).  All guarded by q_mu.
  uint64_t next_chunk = 0;            // next chunk of the plan to open
  struct HostBuf* open_buf = nullptr; // the chunk being filled
  uint64_t open_next_piece = 0, open_npieces = 0;
  uint64_t piece_bytes = 4u << 20;
  int bufs_max = 0, bufs_created = 0; // the pinned ring grows on demand (readers allocate their buffers in parallel)

  // ordered result store
  std::mutex mu;
  std::condition_variable cv;
  std::map<uint64_t, Partial> pending;
  uint64_t next_publish = 0;
  std::vector<uint64_t> values;   // count tags: running totals; u64 tags: elements
  std::deque<std::string> lines;  // XSG_LINES
  uint64_t total = 0;             // count so far / elements so far
  uint64_t nl_running = 0;        // '\n' in all published chunks
  bool finished = false;
  int error = XSG_OK;
  std::string errmsg;

  Clock::time_point t_start;
  xsg_job_stats stats{};
  bool joined = false;
};

static void job_fail(xsg_job* j, int code) {
  std::lock_guard<std::mutex> g(j->mu);
  if (j->error == XSG_OK) {
    j->error = code;
    j->errmsg = last_error_message();
  }
  j->stop.store(true);
  { std::lock_guard<std::mutex> lk(j->q_mu); }  // a waiter is either before its predicate check or already blocked
  j->q_cv_free.notify_all();
  j->q_cv_ready.notify_all();
}

static void publish(xsg_job* j, uint64_t index, Partial&& p) {
  std::lock_guard<std::mutex> g(j->mu);
  j->pending.emplace(index, std::move(p));
  const uint32_t mode = j->opts.mode;
  for (auto it = j->pending.find(j->next_publish); it != j->pending.end(); it = j->pending.find(j->next_publish)) {
    Partial& q = it->second;
    if (mode == XSG_COUNT_MATCHES || mode == XSG_COUNT_LINES) {
      j->total += q.count;
      j->values.push_back(j->total);
    } else if (mode == XSG_LINES) {
      for (std::string& s : q.lines) j->lines.push_back(std::move(s));
      j->total = j->lines.size();
    } else {
      if (q.indices_local)
        for (uint64_t& v : q.u64) v += j->nl_running;
      j->values.insert(j->values.end(), q.u64.begin(), q.u64.end());
      j->total = j->values.size();
    }
    j->nl_running += q.newlines;
    j->pending.erase(it);
    ++j->next_publish;
  }
  j->cv.notify_all();
}

// A pinned host buffer travelling between the two stages.
struct HostBuf {
  void* pinned = nullptr;
  uint64_t cap = 0;
  std::vector<uint8_t> staging;  // compressed bytes before decode
  uint64_t index = 0;            // chunk it currently holds
  std::atomic<uint64_t> pieces_left{0};  // pieces of that chunk still being read
  ~HostBuf() {
    if (pinned) (void)hipHostFree(pinned);
  }
};

// Device side of a worker: two lanes of ctx (= stream) + shard scratch + device buffer.  Every tag alternates
// between them so that the copy of chunk i+1 is already in the queue (and moving) while chunk i is scanned and
// its result waited for: count tags enqueue copy + scan and collect the counters one chunk later; list tags enqueue
// the copy and run their search (which ends in a stream sync of ITS lane) one chunk later.
struct Lane {
  xsg_ctx* ctx = nullptr;
  xsg_shard* shard = nullptr;
  void* dev = nullptr;
  uint64_t cap = 0;
  uint64_t job_serial = 0;  // the job this lane carries the pattern of (lane_prepare)
};
struct Slot {
  Lane lane[2];
  int device = 0;
  ~Slot() {
    for (Lane& l : lane) {
      if (l.shard) xsg_shard_destroy(l.shard);
      if (l.dev) (void)hipFree(l.dev);
      if (l.ctx) xsg_ctx_destroy(l.ctx);
    }
  }
};

// Slots and pinned buffers are kept in per-process pools between jobs: creating them
// costs ~10 ms of hipHostMalloc / hipMalloc / stream setup, which dominated searches of
// small files.  The pools are never torn down (a static destructor would race the HIP
// runtime's own exit).
static std::mutex g_pool_mu;
static std::vector<Slot*>& slot_pool() {
  static std::vector<Slot*>* pool = new std::vector<Slot*>();
  return *pool;
}
static std::vector<HostBuf*>& buf_pool() {
  static std::vector<HostBuf*>* pool = new std::vector<HostBuf*>();
  return *pool;
}
constexpr size_t kMaxIdleSlots = 64, kMaxIdleBufs = 128;

static Slot* slot_take(int device) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  std::vector<Slot*>& pool = slot_pool();
  for (size_t i = 0; i < pool.size(); ++i) {
    if (pool[i]->device == device) {
      Slot* s = pool[i];
      pool.erase(pool.begin() + (ptrdiff_t)i);
      return s;
    }
  }
  return nullptr;
}

static void slot_give_back(Slot* s) {
  {
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (slot_pool().size() < kMaxIdleSlots) {
      slot_pool().push_back(s);
      return;
    }
  }
  delete s;
}

static bool is_count_mode(uint32_t mode) { return mode == XSG_COUNT_MATCHES || mode == XSG_COUNT_LINES; }

// One lane of a slot, ready for this job: its context and shard (created on first use), the job's pattern, a device
// buffer that holds the job's largest chunk.  `serial` = the job the lane was last prepared for.
static int lane_prepare(xsg_job* j, Slot& s, int k) {
  Lane& l = s.lane[k];
  if (l.job_serial == j->serial) return XSG_OK;
  HIP_TRY(hipSetDevice(j->opts.device));
  const uint64_t need = ((j->max_orig + 15u) & ~(uint64_t)15u) + 256u;
  if (!l.ctx) {
    XSG_TRY(xsg_ctx_create(j->opts.device, &l.ctx));
    XSG_TRY(xsg_shard_create(l.ctx, nullptr, 0, nullptr, 0, &l.shard));
    XSG_TRACE("slot: lane %d ctx + shard created", k);
  }
  XSG_TRY(xsg_set_pattern(l.ctx, j->pattern.data(), j->pattern.size(), j->opts.pattern_flags));
  if (need > l.cap) {
    if (l.dev) (void)hipFree(l.dev);
    l.dev = nullptr;
    l.cap = 0;
    HIP_TRY(hipMalloc(&l.dev, need));
    l.cap = need;
    XSG_TRACE("slot: lane %d device buffer %llu bytes", k, (unsigned long long)need);
  }
  l.job_serial = j->serial;
  return XSG_OK;
}

// A worker's slot: taken from the pool or new; lane 0 is made ready here, lane 1 when a second chunk arrives while the
// first is in flight (a stream costs 10-20 ms to create in a fresh process, a device buffer up to 5: a file of one
// chunk never pays for the second lane).
static int slot_prepare(xsg_job* j, Slot** out) {
  Slot* s = slot_take(j->opts.device);
  std::unique_ptr<Slot> own(s ? s : new (std::nothrow) Slot());
  if (!own) return fail(XSG_ENOMEM, "host allocation failed");
  own->device = j->opts.device;
  XSG_TRY(lane_prepare(j, *own, 0));
  *out = own.release();
  return XSG_OK;
}

static int buf_prepare(xsg_job* j, HostBuf** out) {
  HostBuf* b = nullptr;
  {
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (!buf_pool().empty()) {
      b = buf_pool().back();
      buf_pool().pop_back();
    }
  }
  std::unique_ptr<HostBuf> own(b ? b : new (std::nothrow) HostBuf());
  if (!own) return fail(XSG_ENOMEM, "host allocation failed");
  const uint64_t need = ((j->max_orig + 15u) & ~(uint64_t)15u) + 256u;
  if (need > own->cap) {
    if (own->pinned) (void)hipHostFree(own->pinned);
    own->pinned = nullptr;
    own->cap = 0;
    HIP_TRY(hipSetDevice(j->opts.device));
    HIP_TRY(hipHostMalloc(&own->pinned, need, hipHostMallocDefault));
    own->cap = need;
    XSG_TRACE("pinned buffer %llu bytes", (unsigned long long)need);
  }
  if (j->compression != XSG_COMPRESSION_NONE && own->staging.size() < j->max_actual) {
    try {
      own->staging.resize(j->max_actual);  // <= the data file's size (validated)
    } catch (const std::bad_alloc&) {
      return fail(XSG_ENOMEM, "cannot allocate %llu bytes of staging for compressed chunks", (unsigned long long)j->max_actual);
    }
  }
  *out = own.release();
  return XSG_OK;
}

static void bufs_release(xsg_job* j) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  for (HostBuf* b : j->all_bufs) {
    if (buf_pool().size() < kMaxIdleBufs)
      buf_pool().push_back(b);
    else
      delete b;
  }
  j->all_bufs.clear();
}

// ---- NUMA: the feeder threads of a device run on the CPUs next to it -------------------------------
// one reader/feeder group per GPU with NUMA-local pinned buffers.  The pinned buffers already are
// (hipHostMalloc places them on the node nearest the current device unless hipHostMallocNumaUser is given); the
// threads that fill and drain them are bound here to that node's CPUs, read from the PCI device's
// local_cpulist, intersected with what the process may use.  XSG_NUMA=0 switches the binding off.
static bool parse_cpulist(const char* text, cpu_set_t* out) {
  CPU_ZERO(out);
  const char* p = text;
  bool any = false;
  while (*p) {
    char* end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    p = end;
    if (*p == '-') {
      b = strtol(p + 1, &end, 10);
      if (end == p + 1) break;
      p = end;
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
      if (c >= 0) CPU_SET((int)c, out), any = true;
    if (*p == ',') ++p;
    else break;
  }
  return any;
}

static int device_pci_file(int device, const char* leaf, char* buf, size_t cap) {
  char bdf[64] = "";
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return -1;
  for (char* q = bdf; *q; ++q) *q = (char)tolower((unsigned char)*q);
  char path[160];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/%s", bdf, leaf);
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  const size_t n = fread(buf, 1, cap - 1, f);
  fclose(f);
  buf[n] = 0;
  while (n && (buf[strlen(buf) - 1] == '\n' || buf[strlen(buf) - 1] == ' ')) buf[strlen(buf) - 1] = 0;
  return 0;
}

// CPUs local to the device that this process may run on; false = no usable information (or XSG_NUMA=0)
static bool device_cpuset(int device, cpu_set_t* out) {
  static const char* const env = getenv("XSG_NUMA");
  if (env && *env == '0') return false;
  char text[1024];
  if (device_pci_file(device, "local_cpulist", text, sizeof text) != 0) return false;
  cpu_set_t local, mine;
  if (!parse_cpulist(text, &local)) return false;
  if (sched_getaffinity(0, sizeof mine, &mine) != 0) return false;
  CPU_AND(out, &local, &mine);
  return CPU_COUNT(out) > 0 && CPU_COUNT(out) < CPU_COUNT(&mine);  // nothing to do when it is the whole set anyway
}

static void bind_thread_to_device(int device) {
  cpu_set_t set;
  if (device_cpuset(device, &set)) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}

extern "C" int xsg_device_numa(int device, int* node, char* cpulist, size_t cap) {
  int n = 0;
  XSG_TRY(xsg_device_count(&n));
  if (device < 0 || device >= n) return fail(XSG_ENODEV, "device %d out of range", device);
  char text[1024] = "";
  if (node) {
    *node = -1;
    if (device_pci_file(device, "numa_node", text, sizeof text) == 0) *node = atoi(text);
  }
  if (cpulist && cap) {
    cpulist[0] = 0;
    if (device_pci_file(device, "local_cpulist", text, sizeof text) == 0) snprintf(cpulist, cap, "%s", text);
  }
  return XSG_OK;
}

// CPUs this process may use: its affinity mask, capped by the cgroup's CPU quota (cpu.max)
static int cpu_budget() {
  int n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = "";
    long long period = 0;
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long long quota = atoll(q);
      if (quota > 0) n = (int)std::min<long long>(n, std::max<long long>(1, quota / period));
    }
    fclose(f);
  }
  return n;
}
static uint64_t env_u64(const char* name, uint64_t dflt) {
  const char* v = getenv(name);
  return v && *v ? strtoull(v, nullptr, 0) : dflt;
}
// reader threads the pipeline runs whatever the caller asked for (see job_start_impl)
static int auto_readers() {
  static const int n = [] {
    const uint64_t e = env_u64("XSG_MIN_READERS", 0);
    if (e) return (int)std::min<uint64_t>(e, 64);
    return std::min(8, std::max(2, cpu_budget() / 2));
  }();
  return n;
}

// ---- stage 1: read (+ decompress) into a pinned buffer ------------------------
static uint64_t pieces_of(const xsg_job* j, const xsg_file_chunk& fc) {
  if (j->compression != XSG_COMPRESSION_NONE) return 1;  // a decoder needs the whole chunk
  return std::max<uint64_t>(1, fc.actual_size / j->piece_bytes);  // the last piece takes the remainder
}

// A reader's next piece of work: (buffer, piece) of the chunk that is open, opening the next chunk of the plan -- in a
// free buffer of the ring, or in one this reader allocates (the ring grows on demand, up to bufs_max: a fresh process
// pays 3-5 ms per 16 MiB of hipHostMalloc, and several readers pay it side by side while the first chunk is already
// being read).  false: nothing left to read, or the job stopped.
static bool next_piece(xsg_job* j, HostBuf** buf, uint64_t* piece) {
  std::unique_lock<std::mutex> lk(j->q_mu);
  HostBuf* mine = nullptr;  // a buffer this reader holds but has not placed yet
  for (;;) {
    if (j->stop.load()) break;
    if (j->open_buf && j->open_next_piece < j->open_npieces) {
      *buf = j->open_buf;
      *piece = j->open_next_piece++;
      if (mine) {
        j->free_bufs.push_back(mine);
        j->q_cv_free.notify_one();
      }
      return true;
    }
    if (j->next_chunk >= j->plan.size()) break;
    if (!mine && !j->free_bufs.empty()) {
      mine = j->free_bufs.back();
      j->free_bufs.pop_back();
    }
    if (!mine && j->bufs_created < j->bufs_max) {
      ++j->bufs_created;
      lk.unlock();
      HostBuf* b = nullptr;
      const int r = buf_prepare(j, &b);
      lk.lock();
      if (r != XSG_OK) {
        lk.unlock();
        job_fail(j, r);
        return false;
      }
      j->all_bufs.push_back(b);
      mine = b;
      continue;  // the state may have moved while the lock was released
    }
    if (!mine) {
      j->q_cv_free.wait(lk);
      continue;
    }
    j->open_buf = mine;
    mine = nullptr;
    j->open_buf->index = j->next_chunk++;
    j->open_npieces = pieces_of(j, j->plan[j->open_buf->index]);
    j->open_buf->pieces_left.store(j->open_npieces);
    j->open_next_piece = 0;
    j->q_cv_free.notify_all();  // readers waiting for a buffer: there are pieces to take now
  }
  if (mine) j->free_bufs.push_back(mine);
  return false;
}

static void reader_loop(xsg_job* j, double& t_read, double& t_dec, uint64_t& rbytes) {
  HostBuf* b = nullptr;
  uint64_t piece = 0;
  while (next_piece(j, &b, &piece)) {
    const xsg_file_chunk& fc = j->plan[b->index];
    const uint64_t np = pieces_of(j, fc);
    const uint64_t lo = piece * j->piece_bytes;
    const uint64_t n = piece + 1 == np ? fc.actual_size - lo : j->piece_bytes;
    auto t0 = Clock::now();
    int r = XSG_OK;
    // the plan was validated at start; the buffers were sized from it -- checked again where the bytes land
    if (fc.original_size > b->cap || (j->compression != XSG_COMPRESSION_NONE && fc.actual_size > b->staging.size()) ||
        (j->compression == XSG_COMPRESSION_NONE && fc.actual_size > b->cap))
      r = fail(XSG_EIO, "chunk %llu does not fit its buffers", (unsigned long long)b->index);
    if (r == XSG_OK && n) {
      uint8_t* dst = j->compression == XSG_COMPRESSION_NONE ? static_cast<uint8_t*>(b->pinned) : b->staging.data();
      r = pread_full(j->fd, dst + lo, n, fc.actual_offset + lo);
    }
    t_read += seconds_since(t0);
    if (r == XSG_OK && j->compression != XSG_COMPRESSION_NONE && fc.original_size) {
      t0 = Clock::now();
      r = decompress_chunk(j->compression, b->staging.data(), fc.actual_size, static_cast<uint8_t*>(b->pinned),
                           fc.original_size);
      t_dec += seconds_since(t0);
    }
    if (r != XSG_OK) {
      job_fail(j, r);  // (the buffer stays with the job: other readers may still be writing their pieces of it)
      break;
    }
    rbytes += n;
    if (b->pieces_left.fetch_sub(1) == 1) {  // the chunk is complete
      {
        std::lock_guard<std::mutex> lk(j->q_mu);
        j->ready_bufs.push_back(b);
      }
      j->q_cv_ready.notify_one();
    }
  }
}

static void reader_main(xsg_job* j) {
  double t_read = 0, t_dec = 0;
  uint64_t rbytes = 0;
  bind_thread_to_device(j->opts.device);
  try {
    reader_loop(j, t_read, t_dec, rbytes);
  } catch (const std::bad_alloc&) {
    (void)fail(XSG_ENOMEM, "host allocation failed in a reader");
    job_fail(j, XSG_ENOMEM);
  } catch (const std::exception& e) {
    (void)fail(XSG_EIO, "reader: %s", e.what());
    job_fail(j, XSG_EIO);
  }
  {
    std::lock_guard<std::mutex> lk(j->q_mu);
    --j->readers_running;
  }
  j->q_cv_ready.notify_all();  // the last reader leaving lets idle workers finish
  std::lock_guard<std::mutex> g(j->mu);
  j->stats.bytes_read += rbytes;
  j->stats.seconds_read += t_read;
  j->stats.seconds_decompress += t_dec;
}

// ---- stage 2: H2D on the worker's stream, scan, publish -------------------------
static int bind_chunk(xsg_job* j, Lane& l, const HostBuf& hb) {
  const xsg_file_chunk& fc = j->plan[hb.index];
  if (fc.original_size > hb.cap || fc.original_size > l.cap) return fail(XSG_EINVAL, "chunk larger than its buffers");
  if (fc.original_size)
    HIP_TRY(hipMemcpyAsync(l.dev, hb.pinned, fc.original_size, hipMemcpyHostToDevice, l.ctx->stream));
  xsg_chunk ch{};
  ch.offset = 0;
  ch.length = fc.original_size;
  ch.global_offset = fc.original_offset;
  ch.line_base = fc.first_line;  // from the metafile, or AUTO (then local indices + running base at publish)
  return xsg_shard_rebind(l.shard, l.dev, l.cap, &ch, 1);
}

// list tags, second half: the search on the lane whose copy was queued one chunk earlier, then publish
static int list_collect(xsg_job* j, Lane& l, const HostBuf& hb) {
  const xsg_file_chunk& fc = j->plan[hb.index];
  const uint32_t mode = j->opts.mode;
  Partial p;
  if (mode == XSG_LINES) {
    uint64_t n = 0, nl = 0, nb = 0;
    const uint64_t* lens = nullptr;
    const char* bytes = nullptr;
    XSG_TRY(xsg_search(l.shard, mode, &n));
    XSG_TRY(xsg_result_lines_view(l.shard, &lens, &bytes, nullptr, &nl, &nb));  // the shard's pinned buffers: one copy, into the strings
    p.lines.reserve(nl);
    uint64_t at = 0;
    for (uint64_t i = 0; i < nl; ++i) {
      p.lines.emplace_back(bytes + at, lens[i]);
      at += lens[i];
    }
  } else {
    // the result is read where the search left it: the shard's pinned buffer (stored there by the kernels on the
    // one-sync route, one pinned D2H copy otherwise) -- no second trip through the device queue per chunk
    uint64_t n = 0;
    const uint64_t* view = nullptr;
    XSG_TRY(xsg_search(l.shard, mode, &n));
    XSG_TRY(xsg_result_u64_view(l.shard, &view, &n));
    p.u64.assign(view, view + n);
    if (mode == XSG_LINE_INDICES && fc.first_line == XSG_LINE_BASE_AUTO) {
      p.indices_local = true;
      XSG_TRY(xsg_result_newlines(l.shard, &p.newlines));
    }
  }
  // the search synchronised the lane's stream: its pinned buffer is free again
  publish(j, hb.index, std::move(p));
  return XSG_OK;
}

// count tags, first half: copy + scan of the chunk go into lane k's queue; nothing is waited for
static int count_enqueue(xsg_job* j, Slot& s, int k, const HostBuf& hb) {
  XSG_TRY(bind_chunk(j, s.lane[k], hb));
  return xsg_count_begin(s.lane[k].shard, j->opts.mode);
}
// second half: wait for lane k, publish
static int count_collect(xsg_job* j, Slot& s, int k, uint64_t index) {
  uint64_t ctr[XSG_NUM_COUNTERS];
  XSG_TRY(xsg_count_end(s.lane[k].shard, ctr));
  Partial p;
  p.count = ctr[j->opts.mode == XSG_COUNT_MATCHES ? XSG_CTR_MATCHES : XSG_CTR_LINES];
  publish(j, index, std::move(p));
  return XSG_OK;
}

static void give_back(xsg_job* j, HostBuf* b) {
  {
    std::lock_guard<std::mutex> lk(j->q_mu);
    j->free_bufs.push_back(b);
  }
  j->q_cv_free.notify_one();
}

static void worker_body(xsg_job* j, double& t_dev, uint64_t& bytes, uint64_t& chunks) {
  Slot* sp = nullptr;
  int r = slot_prepare(j, &sp);
  if (r != XSG_OK) {
    job_fail(j, r);
    return;
  }
  const bool counting = is_count_mode(j->opts.mode);
  HostBuf* inflight = nullptr;  // the chunk whose copy (count tags: and pass) is in lane (k ^ 1)'s queue
  int k = 0;
  // The second lane costs a fresh process 10-20 ms (a stream, a device buffer: profiles/r04_cli_start.txt) and buys
  // the ~50 us a chunk that one lane loses between two chunks: made when the job is long enough to pay that back --
  // or used when the slot already has it (a pooled slot of an earlier, larger job).
  const uint64_t per_worker = (j->plan.size() + (uint64_t)j->opts.num_threads - 1) / (uint64_t)std::max(j->opts.num_threads, 1);
  const bool two_lanes = sp->lane[1].ctx != nullptr || per_worker >= env_u64("XSG_LANE2_MIN_CHUNKS", 32);
  for (;;) {
    HostBuf* b = nullptr;
    bool done = false;
    {
      std::unique_lock<std::mutex> lk(j->q_mu);
      // with a chunk in flight, do not sleep on the queue: collect that result first (its pinned buffer goes
      // back to the readers, who may be waiting for exactly that one)
      if (!inflight)
        j->q_cv_ready.wait(lk, [&] { return !j->ready_bufs.empty() || j->readers_running == 0 || j->stop.load(); });
      if (j->stop.load()) {
        done = true;
      } else if (!j->ready_bufs.empty()) {
        b = j->ready_bufs.front();
        j->ready_bufs.pop_front();
      } else if (!inflight) {
        done = true;  // readers are done and nothing is left
      }
    }
    if (done) break;
    const auto t0 = Clock::now();
    if (b) r = lane_prepare(j, *sp, k);
    if (b && r == XSG_OK) r = counting ? count_enqueue(j, *sp, k, *b) : bind_chunk(j, sp->lane[k], *b);
    if (inflight) {
      const int r2 = counting ? count_collect(j, *sp, k ^ 1, inflight->index) : list_collect(j, sp->lane[k ^ 1], *inflight);
      bytes += j->plan[inflight->index].original_size;
      ++chunks;
      if (r2 != XSG_OK) (void)hipDeviceSynchronize();  // the failed lane may still read the pinned buffer
      give_back(j, inflight);
      inflight = nullptr;
      if (r == XSG_OK) r = r2;
    }
    if (b) {
      if (r == XSG_OK) {
        inflight = b;
        if (two_lanes) {
          k ^= 1;
        } else {  // one lane: this chunk's result is collected before the next chunk is looked at
          const int r2 = counting ? count_collect(j, *sp, k, inflight->index) : list_collect(j, sp->lane[k], *inflight);
          bytes += j->plan[inflight->index].original_size;
          ++chunks;
          if (r2 != XSG_OK) (void)hipDeviceSynchronize();
          give_back(j, inflight);
          inflight = nullptr;
          r = r2;
        }
      } else {
        (void)hipDeviceSynchronize();  // the failed lane may still read the pinned buffer
        give_back(j, b);
      }
    }
    t_dev += seconds_since(t0);
    if (r != XSG_OK) {
      job_fail(j, r);
      break;
    }
  }
  if (inflight) {  // stopped with a chunk in flight: let its lane drain before the buffer is handed back
    (void)hipStreamSynchronize(sp->lane[k ^ 1].ctx->stream);
    give_back(j, inflight);
  }
  if (r == XSG_OK)
    slot_give_back(sp);
  else
    delete sp;  // do not recycle a slot whose last operation failed
}

static void worker_main(xsg_job* j) {
  double t_dev = 0;
  uint64_t bytes = 0, chunks = 0;
  bind_thread_to_device(j->opts.device);
  try {
    worker_body(j, t_dev, bytes, chunks);
  } catch (const std::bad_alloc&) {  // nothing may leave a thread body (or the extern "C" boundary) as an exception
    (void)fail(XSG_ENOMEM, "host allocation failed in a device worker");
    job_fail(j, XSG_ENOMEM);
  } catch (const std::exception& e) {
    (void)fail(XSG_EIO, "device worker: %s", e.what());
    job_fail(j, XSG_EIO);
  }
  if (j->stop.load()) {  // wake everybody that may still be waiting on a queue
    { std::lock_guard<std::mutex> lk(j->q_mu); }
    j->q_cv_free.notify_all();
    j->q_cv_ready.notify_all();
  }
  std::lock_guard<std::mutex> g(j->mu);
  j->stats.bytes_scanned += bytes;
  j->stats.chunks += chunks;
  j->stats.seconds_device += t_dev;
  if (--j->active == 0) {  // the last worker closes the result (Searcher.h:116-119)
    XSG_TRACE("job: last worker done (%llu chunks)", (unsigned long long)j->stats.chunks);
    j->finished = true;
    j->stats.seconds_total = seconds_since(j->t_start);
    j->cv.notify_all();
  }
}

extern "C" void xsg_job_opts_init(xsg_job_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof *o);
  o->struct_size = sizeof *o;
  o->mode = XSG_COUNT_MATCHES;
  o->device = 0;
  o->num_threads = 1;
  o->num_max_readers = 1;
  o->chunk_bytes = 16u << 20;
}

static int job_start_impl(const void* pattern, size_t plen, const char* file_path, const char* meta_file_path,
                          const xsg_job_opts* opts, xsg_job** out);

extern "C" int xsg_job_start(const void* pattern, size_t plen, const char* file_path, const char* meta_file_path,
                             const xsg_job_opts* opts, xsg_job** out) {
  try {
    return job_start_impl(pattern, plen, file_path, meta_file_path, opts, out);
  } catch (const std::bad_alloc&) {  // e.g. a plan with millions of records
    return fail(XSG_ENOMEM, "host allocation failed while setting up the job");
  } catch (const std::exception& e) {
    return fail(XSG_EIO, "xsg_job_start: %s", e.what());
  }
}

static int job_start_impl(const void* pattern, size_t plen, const char* file_path, const char* meta_file_path,
                          const xsg_job_opts* opts, xsg_job** out) {
  XSG_TRACE("job: start");
  if (!out) return fail(XSG_EINVAL, "out is null");
  *out = nullptr;
  if (!opts || opts->struct_size != sizeof(xsg_job_opts)) return fail(XSG_EINVAL, "bad xsg_job_opts (struct_size)");
  if (!pattern || plen == 0 || plen > XSG_MAX_PATTERN) return fail(XSG_EINVAL, "pattern must be 1..%u bytes",
                                                                  XSG_MAX_PATTERN);
  if ((opts->pattern_flags & XSG_FLAG_REGEX) && plen > XSG_MAX_REGEX) return fail(XSG_EINVAL, "expression longer than %u bytes", XSG_MAX_REGEX);
  if (opts->mode > XSG_LINES) return fail(XSG_EINVAL, "bad mode %u", opts->mode);
  if (opts->num_threads < 1 || opts->num_max_readers < 1) return fail(XSG_EINVAL, "num_threads/num_max_readers < 1");
  const bool line_mode = opts->mode != XSG_COUNT_MATCHES && opts->mode != XSG_MATCH_BYTE_OFFSETS;
  if (opts->pattern_flags & XSG_FLAG_REGEX) {  // refuse an expression the kernel cannot decide here, not in a worker
    uint32_t npos = 0;
    std::vector<uint32_t> sets(32 * 8, 0);
    XSG_TRY(xsg_regex_check(pattern, plen, opts->pattern_flags & XSG_FLAG_IGNORE_CASE, &npos, sets.data()));
    for (uint32_t k = 0; line_mode && k < npos; ++k)
      if (sets[8 * k] & (1u << '\n')) return fail(XSG_ENOTSUP, "line modes do not accept a pattern that can match '\\n'");
    if (line_mode && npos == 0) {  // the automaton route: does a set of the expression accept '\n'?
      xsg_regex_dfa info;
      XSG_TRY(xsg_regex_dfa_info(pattern, plen, opts->pattern_flags & XSG_FLAG_IGNORE_CASE, &info, nullptr, nullptr, 0));
      if (info.multiline) return fail(XSG_ENOTSUP, "line modes do not accept a pattern that can match '\\n'");
    }
  }  // (a LITERAL that contains '\n' is served by every tag since round 4: xsg.h, xsg_set_pattern)
  std::unique_ptr<xsg_job> j(new (std::nothrow) xsg_job());
  if (!j) return fail(XSG_ENOMEM, "host allocation failed");
  j->opts = *opts;
  if (!j->opts.chunk_bytes) j->opts.chunk_bytes = 16u << 20;
  j->pattern.assign((const uint8_t*)pattern, (const uint8_t*)pattern + plen);
  uint64_t fsize = 0;
  XSG_TRY(open_ro(file_path, &j->fd, &fsize));
  int r = XSG_OK;
  if (meta_file_path && *meta_file_path) {
    r = read_meta(meta_file_path, &j->compression, j->plan, nullptr);
    // A metafile is input like any other: every record is checked BEFORE a buffer is sized from it or a byte is
    // read or decoded through it (sizes < 2^40 were enforced by read_meta).
    for (size_t i = 0; r == XSG_OK && i < j->plan.size(); ++i) {
      const xsg_file_chunk& c = j->plan[i];
      if (c.actual_size > fsize || c.actual_offset > fsize - c.actual_size)
        r = fail(XSG_EIO, "metafile '%s': chunk %zu lies beyond the end of '%s'", meta_file_path, i, file_path);
      else if (j->compression == XSG_COMPRESSION_NONE && c.actual_size != c.original_size)
        r = fail(XSG_EIO, "metafile '%s': chunk %zu of an uncompressed file has actual_size != original_size",
                 meta_file_path, i);
      else if (j->compression == XSG_COMPRESSION_LZ4 && (c.original_size > INT32_MAX || c.actual_size > INT32_MAX))
        r = fail(XSG_EIO, "metafile '%s': chunk %zu is too large for an LZ4 block", meta_file_path, i);
      else if (j->compression != XSG_COMPRESSION_NONE && c.original_size && !c.actual_size)
        r = fail(XSG_EIO, "metafile '%s': chunk %zu has no compressed bytes", meta_file_path, i);
    }
    if (r == XSG_OK && j->compression == XSG_COMPRESSION_LZ4) r = need_lz4();
    if (r == XSG_OK && j->compression == XSG_COMPRESSION_ZSTD) r = need_zstd();
  } else {
    r = plan_plain(j->fd, fsize, j->opts.chunk_bytes, j->plan);
  }
  if (r != XSG_OK) {
    close(j->fd);
    return r;
  }
  // fail early and loudly without a device: there is no CPU search path (after the file checks, so that a bad
  // path or metafile is reported as such on any host)
  int ndev = 0;
  r = xsg_device_count(&ndev);
  if (r == XSG_OK && (opts->device < 0 || opts->device >= ndev)) r = fail(XSG_ENODEV, "device %d out of range", opts->device);
  if (r != XSG_OK) {
    close(j->fd);
    return r;
  }
  j->stats.plan_chunks = j->plan.size();
  XSG_TRACE("job: plan of %zu chunks, device count %d", j->plan.size(), ndev);
  if (opts->chunk_begin || opts->chunk_end) {
    if (opts->chunk_begin > opts->chunk_end || opts->chunk_end > j->plan.size()) {
      close(j->fd);
      return fail(XSG_EINVAL, "chunk range [%llu, %llu) outside the plan of %zu chunks",
                  (unsigned long long)opts->chunk_begin, (unsigned long long)opts->chunk_end, j->plan.size());
    }
    j->plan = std::vector<xsg_file_chunk>(j->plan.begin() + (ptrdiff_t)opts->chunk_begin,
                                          j->plan.begin() + (ptrdiff_t)opts->chunk_end);
  }
  for (const xsg_file_chunk& c : j->plan) {
    j->max_orig = std::max(j->max_orig, c.original_size);
    j->max_actual = std::max(j->max_actual, c.actual_size);
  }
  j->t_start = Clock::now();
  const uint64_t nch = std::max<uint64_t>(j->plan.size(), 1);
  const int nworkers = (int)std::min<uint64_t>((uint64_t)opts->num_threads, nch);
  // How many threads READ is the pipeline's business, not the caller's: num_threads = 1, num_max_readers = 1 is what the
  // reference's one-call API defaults to (README.md:37), and one thread copying page-cache bytes into pinned memory
  // moves 6-10 GiB/s where the link to the device takes 50.  The caller's number is a floor; the pipeline adds readers
  // up to half the CPUs the process may use (2..8; XSG_MIN_READERS overrides), and they share the chunks piece by piece.
  j->piece_bytes = std::max<uint64_t>(env_u64("XSG_READ_PIECE", 4u << 20), 1u << 16);
  uint64_t pieces = 0;
  for (const xsg_file_chunk& c : j->plan) pieces += pieces_of(j.get(), c);
  // (readers of a compressed file DECODE: that is what its rate is made of -- 3.5 GiB/s of text per LZ4 thread -- and the device
  // workers mostly sleep, so they get the whole CPU budget, 16 at most)
  const int want_readers = j->compression != XSG_COMPRESSION_NONE && !env_u64("XSG_MIN_READERS", 0)
                               ? (int)std::min<uint64_t>(std::max<uint64_t>(cpu_budget(), 2), 16)
                               : auto_readers();
  const int nreaders = (int)std::min<uint64_t>((uint64_t)std::max(opts->num_max_readers, want_readers), std::max<uint64_t>(pieces, 1));
  // the ring: two chunks per worker (one being scanned, one whose copy is queued behind it), the one the readers are
  // filling, and two complete ones in between -- never more buffers than chunks.  Allocated on demand by the readers.
  // A COMPRESSED chunk is one piece: every reader decodes a whole chunk into a buffer of its own, so the ring also needs one
  // buffer per reader -- with 2 * workers + 3 twelve LZ4 decoders found three to seven buffers free and ran at 25 GiB/s of
  // text where 12 x 3.5 GiB/s were there (bench.py e2e.lz4, profiles/r04_lz4_threads.txt).
  const uint64_t ring = j->compression != XSG_COMPRESSION_NONE ? (uint64_t)(2 * nworkers + nreaders + 1) : (uint64_t)(2 * nworkers + 3);
  j->bufs_max = (int)std::min<uint64_t>(ring, nch);
  j->active = nworkers;
  j->readers_running = nreaders;
  XSG_TRACE("job: starting %d readers + %d workers, ring of up to %d pinned buffers, %llu pieces", nreaders, nworkers,
            j->bufs_max, (unsigned long long)pieces);
  xsg_job* raw = j.release();
  for (int t = 0; t < nreaders; ++t) raw->threads.emplace_back(reader_main, raw);
  for (int t = 0; t < nworkers; ++t) raw->threads.emplace_back(worker_main, raw);
  *out = raw;
  return XSG_OK;
}

extern "C" int xsg_job_join(xsg_job* j) {
  if (!j) return fail(XSG_EINVAL, "job is null");
  for (std::thread& t : j->threads)
    if (t.joinable()) t.join();
  j->joined = true;
  std::lock_guard<std::mutex> g(j->mu);
  if (j->error != XSG_OK) return fail(j->error, "%s", j->errmsg.c_str());
  return XSG_OK;
}

extern "C" void xsg_job_destroy(xsg_job* j) {
  if (!j) return;
  j->stop.store(true);
  { std::lock_guard<std::mutex> lk(j->q_mu); }
  j->q_cv_free.notify_all();
  j->q_cv_ready.notify_all();
  for (std::thread& t : j->threads)
    if (t.joinable()) t.join();
  bufs_release(j);
  if (j->fd >= 0) close(j->fd);
  delete j;
}

extern "C" int xsg_job_total(xsg_job* j, uint64_t* total) {
  if (!j || !total) return fail(XSG_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(j->mu);
  *total = j->total;
  return XSG_OK;
}

extern "C" int xsg_job_wait(xsg_job* j, uint64_t index, uint64_t* available, int* finished) {
  if (!j) return fail(XSG_EINVAL, "job is null");
  std::unique_lock<std::mutex> lk(j->mu);
  const bool is_lines = j->opts.mode == XSG_LINES;
  auto avail = [&] { return is_lines ? (uint64_t)j->lines.size() : (uint64_t)j->values.size(); };
  j->cv.wait(lk, [&] { return avail() > index || j->finished; });
  if (available) *available = avail();
  if (finished) *finished = j->finished ? 1 : 0;
  if (j->finished && j->error != XSG_OK && avail() <= index) return fail(j->error, "%s", j->errmsg.c_str());
  return XSG_OK;
}

extern "C" int xsg_job_poll(xsg_job* j, uint64_t* available, int* finished) {
  if (!j) return fail(XSG_EINVAL, "job is null");
  std::lock_guard<std::mutex> g(j->mu);
  if (available) *available = j->opts.mode == XSG_LINES ? (uint64_t)j->lines.size() : (uint64_t)j->values.size();
  if (finished) *finished = j->finished ? 1 : 0;
  return XSG_OK;
}

extern "C" int xsg_job_get_u64(xsg_job* j, uint64_t first, uint64_t n, uint64_t* out) {
  if (!j || (!out && n)) return fail(XSG_EINVAL, "null argument");
  if (j->opts.mode == XSG_LINES) return fail(XSG_ESTATE, "XSG_LINES results are strings: use xsg_job_get_line");
  std::lock_guard<std::mutex> g(j->mu);
  if (first + n > j->values.size()) return fail(XSG_EINVAL, "range beyond the available results");
  if (n) memcpy(out, j->values.data() + first, 8 * n);
  return XSG_OK;
}

extern "C" int xsg_job_get_line(xsg_job* j, uint64_t index, const char** data, uint64_t* len) {
  if (!j || !data || !len) return fail(XSG_EINVAL, "null argument");
  if (j->opts.mode != XSG_LINES) return fail(XSG_ESTATE, "not an XSG_LINES job");
  std::lock_guard<std::mutex> g(j->mu);
  if (index >= j->lines.size()) return fail(XSG_EINVAL, "index beyond the available results");
  const std::string& s = j->lines[index];  // deque: references stay valid while the job lives
  *data = s.data();
  *len = s.size();
  return XSG_OK;
}

extern "C" int xsg_job_stats_get(xsg_job* j, xsg_job_stats* stats) {
  if (!j || !stats) return fail(XSG_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(j->mu);
  *stats = j->stats;
  stats->newlines = j->nl_running;
  if (!j->finished) stats->seconds_total = seconds_since(j->t_start);
  return XSG_OK;
}


// ---------------------------------------------------------------------------
// several jobs of one search (one per device): the count is the sum of their totals.  That sum is the search's one
// exchange step and goes over RCCL: every job's total is put into a word on its own GPU, one ncclAllReduce over a
// per-process communicator of those devices (ncclCommInitAll, kept for the life of the process: creating one costs
// far more than the 8-byte exchange), and the result is read back from the first device.  Falls back to adding on
// the host -- and says so through *via_rccl -- when there is no librccl, when a device is listed twice (RCCL
// wants one rank per GPU), or when XS_REDUCE=host.
// ---------------------------------------------------------------------------
namespace {
struct LocalClique {
  std::vector<int> devices;
  std::vector<xsg_ctx*> ctxs;
  std::vector<uint64_t*> d_word;
  xsg_comm* comm = nullptr;
};
std::mutex g_clique_mu;
std::vector<LocalClique*>& cliques() {
  static std::vector<LocalClique*>* v = new std::vector<LocalClique*>();  // never torn down (see slot_pool)
  return *v;
}

int clique_for(const std::vector<int>& devices, LocalClique** out) {
  for (LocalClique* c : cliques())
    if (c->devices == devices) {
      *out = c;
      return XSG_OK;
    }
  std::unique_ptr<LocalClique> c(new LocalClique());
  c->devices = devices;
  int r = XSG_OK;
  for (int d : devices) {
    xsg_ctx* x = nullptr;
    r = xsg_ctx_create(d, &x);
    if (r != XSG_OK) break;
    c->ctxs.push_back(x);
    void* p = nullptr;
    if (hipSetDevice(d) != hipSuccess || hipMalloc(&p, 64) != hipSuccess) {
      r = fail(XSG_ENOMEM, "hipMalloc on device %d failed", d);
      break;
    }
    c->d_word.push_back(static_cast<uint64_t*>(p));
  }
  if (r == XSG_OK) r = xsg_comm_create_local(c->ctxs.data(), (int)c->ctxs.size(), &c->comm);
  if (r != XSG_OK) {
    for (size_t i = 0; i < c->d_word.size(); ++i) {
      (void)hipSetDevice(devices[i]);
      (void)hipFree(c->d_word[i]);
    }
    for (xsg_ctx* x : c->ctxs) xsg_ctx_destroy(x);
    return r;
  }
  cliques().push_back(c.get());
  *out = c.release();
  return XSG_OK;
}
}  // namespace

extern "C" int xsg_jobs_reduce_total(xsg_job* const* jobs, int n, uint64_t* total, int* via_rccl) {
  if (!jobs || n < 1 || !total) return fail(XSG_EINVAL, "bad argument");
  return guarded("xsg_jobs_reduce_total", [&]() -> int {
    if (via_rccl) *via_rccl = 0;
    uint64_t host_sum = 0;
    std::vector<int> devices;
    std::vector<uint64_t> totals;
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
      if (!jobs[i]) return fail(XSG_EINVAL, "job %d is null", i);
      uint64_t t = 0;
      XSG_TRY(xsg_job_total(jobs[i], &t));
      host_sum += t;
      totals.push_back(t);
      for (int d : devices) distinct &= d != jobs[i]->opts.device;
      devices.push_back(jobs[i]->opts.device);
    }
    *total = host_sum;
    const char* how = getenv("XS_REDUCE");
    if (n < 2 || !distinct || (how && strcmp(how, "host") == 0)) return XSG_OK;
    std::lock_guard<std::mutex> g(g_clique_mu);
    LocalClique* c = nullptr;
    if (clique_for(devices, &c) != XSG_OK) return XSG_OK;  // no librccl / init refused: the host sum stands
    for (int i = 0; i < n; ++i) {
      HIP_TRY(hipSetDevice(devices[i]));
      HIP_TRY(hipMemcpyAsync(c->d_word[i], &totals[i], 8, hipMemcpyHostToDevice, c->ctxs[i]->stream));
    }
    uint64_t sum = 0;
    XSG_TRY(xsg_reduce_counts(c->comm, c->d_word.data(), 1, &sum));
    if (sum != host_sum)
      return fail(XSG_EHIP, "RCCL sum %llu differs from the host sum %llu", (unsigned long long)sum,
                  (unsigned long long)host_sum);
    if (via_rccl) *via_rccl = 1;
    return XSG_OK;
  });
}

// ---------------------------------------------------------------------------
// the lower seam: chunks in host memory (reference-style searcher functors)
// ---------------------------------------------------------------------------
struct HostSlot {
  xsg_ctx* ctx = nullptr;
  xsg_shard* shard = nullptr;
  void* pinned = nullptr;
  void* dev = nullptr;
  uint64_t cap = 0;
  ~HostSlot() {
    if (shard) xsg_shard_destroy(shard);
    if (pinned) (void)hipHostFree(pinned);
    if (dev) (void)hipFree(dev);
    if (ctx) xsg_ctx_destroy(ctx);
  }
};

struct xsg_host_searcher {
  int device = 0;
  std::vector<uint8_t> pattern;
  uint32_t flags = 0;
  int max_slots = 1;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::unique_ptr<HostSlot>> all;
  std::vector<HostSlot*> free_slots;
};

static int host_slot_acquire(xsg_host_searcher* hs, HostSlot** out) {
  std::unique_lock<std::mutex> lk(hs->mu);
  for (;;) {
    if (!hs->free_slots.empty()) {
      *out = hs->free_slots.back();
      hs->free_slots.pop_back();
      return XSG_OK;
    }
    if ((int)hs->all.size() < hs->max_slots) {
      hs->all.emplace_back(new HostSlot());
      HostSlot* s = hs->all.back().get();
      lk.unlock();
      int r = xsg_ctx_create(hs->device, &s->ctx);
      if (r == XSG_OK) r = xsg_set_pattern(s->ctx, hs->pattern.data(), hs->pattern.size(), hs->flags);
      if (r == XSG_OK) r = xsg_shard_create(s->ctx, nullptr, 0, nullptr, 0, &s->shard);
      if (r != XSG_OK) {
        // give the place back, or later callers would wait for a slot that never comes
        lk.lock();
        for (size_t i = 0; i < hs->all.size(); ++i)
          if (hs->all[i].get() == s) {
            hs->all.erase(hs->all.begin() + (ptrdiff_t)i);
            break;
          }
        lk.unlock();
        hs->cv.notify_one();
        return r;
      }
      *out = s;
      return XSG_OK;
    }
    hs->cv.wait(lk);
  }
}

static void host_slot_release(xsg_host_searcher* hs, HostSlot* s) {
  {
    std::lock_guard<std::mutex> g(hs->mu);
    hs->free_slots.push_back(s);
  }
  hs->cv.notify_one();
}

// stage the chunk: host -> pinned -> device, bind it as a 1-chunk shard (local offsets, line base 0)
static int host_stage(HostSlot* s, const void* data, uint64_t len) {
  const uint64_t need = ((len + 15u) & ~(uint64_t)15u) + 256u;
  if (need > s->cap) {
    if (s->pinned) (void)hipHostFree(s->pinned);
    if (s->dev) (void)hipFree(s->dev);
    s->pinned = s->dev = nullptr;
    s->cap = 0;
    const uint64_t cap = std::max<uint64_t>(need, 1u << 20);
    HIP_TRY(hipSetDevice(s->ctx->device));
    HIP_TRY(hipHostMalloc(&s->pinned, cap, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&s->dev, cap));
    s->cap = cap;
  }
  if (len) {
    memcpy(s->pinned, data, len);
    HIP_TRY(hipMemcpyAsync(s->dev, s->pinned, len, hipMemcpyHostToDevice, s->ctx->stream));
  }
  xsg_chunk ch{};
  ch.offset = 0;
  ch.length = len;
  ch.global_offset = 0;
  ch.line_base = 0;
  return xsg_shard_rebind(s->shard, s->dev, s->cap, &ch, 1);
}

extern "C" int xsg_host_searcher_create(int device, const void* pattern, size_t plen, uint32_t flags, int max_slots,
                                        xsg_host_searcher** out) {
  if (!out) return fail(XSG_EINVAL, "out is null");
  *out = nullptr;
  if (!pattern || plen == 0 || plen > XSG_MAX_PATTERN) return fail(XSG_EINVAL, "pattern must be 1..%u bytes",
                                                                  XSG_MAX_PATTERN);
  if (flags & XSG_FLAG_REGEX) XSG_TRY(xsg_regex_check(pattern, plen, flags & XSG_FLAG_IGNORE_CASE, nullptr, nullptr));
  int ndev = 0;
  XSG_TRY(xsg_device_count(&ndev));
  if (device < 0 || device >= ndev) return fail(XSG_ENODEV, "device %d out of range", device);
  std::unique_ptr<xsg_host_searcher> hs(new (std::nothrow) xsg_host_searcher());
  if (!hs) return fail(XSG_ENOMEM, "host allocation failed");
  hs->device = device;
  hs->pattern.assign((const uint8_t*)pattern, (const uint8_t*)pattern + plen);
  hs->flags = flags;
  hs->max_slots = max_slots < 1 ? 1 : max_slots;
  // build the first slot now so that a bad pattern / missing GPU fails here, not in the first search
  HostSlot* s = nullptr;
  XSG_TRY(host_slot_acquire(hs.get(), &s));
  host_slot_release(hs.get(), s);
  *out = hs.release();
  return XSG_OK;
}

extern "C" void xsg_host_searcher_destroy(xsg_host_searcher* hs) { delete hs; }

struct SlotLease {
  xsg_host_searcher* hs;
  HostSlot* s = nullptr;
  explicit SlotLease(xsg_host_searcher* h) : hs(h) {}
  ~SlotLease() {
    if (s) host_slot_release(hs, s);
  }
};

extern "C" int xsg_host_count(xsg_host_searcher* hs, const void* data, uint64_t len, int skip_to_nl,
                              uint64_t* count) {
  if (!hs || !count || (!data && len)) return fail(XSG_EINVAL, "null argument");
  SlotLease l(hs);
  XSG_TRY(host_slot_acquire(hs, &l.s));
  XSG_TRY(host_stage(l.s, data, len));
  uint64_t ctr[XSG_NUM_COUNTERS];
  XSG_TRY(xsg_count(l.s->shard, skip_to_nl ? XSG_COUNT_LINES : XSG_COUNT_MATCHES, ctr));
  *count = ctr[skip_to_nl ? XSG_CTR_LINES : XSG_CTR_MATCHES];
  return XSG_OK;
}

extern "C" int xsg_host_offsets(xsg_host_searcher* hs, uint32_t mode, const void* data, uint64_t len, uint64_t** out,
                                uint64_t* n) {
  if (!hs || !out || !n || (!data && len)) return fail(XSG_EINVAL, "null argument");
  if (mode != XSG_MATCH_BYTE_OFFSETS && mode != XSG_LINE_BYTE_OFFSETS && mode != XSG_LINE_INDICES)
    return fail(XSG_EINVAL, "mode %u has no uint64 list result", mode);
  SlotLease l(hs);
  XSG_TRY(host_slot_acquire(hs, &l.s));
  XSG_TRY(host_stage(l.s, data, len));
  uint64_t cnt = 0;
  XSG_TRY(xsg_search(l.s->shard, mode, &cnt));
  uint64_t* buf = static_cast<uint64_t*>(malloc(8 * std::max<uint64_t>(cnt, 1)));
  if (!buf) return fail(XSG_ENOMEM, "host allocation failed");
  const int r = xsg_result_u64(l.s->shard, buf, cnt);
  if (r != XSG_OK) {
    free(buf);
    return r;
  }
  *out = buf;
  *n = cnt;
  return XSG_OK;
}

extern "C" int xsg_host_lines(xsg_host_searcher* hs, const void* data, uint64_t len, uint64_t** lengths, char** bytes,
                              uint64_t* n, uint64_t* nbytes) {
  if (!hs || !lengths || !bytes || !n || !nbytes || (!data && len)) return fail(XSG_EINVAL, "null argument");
  SlotLease l(hs);
  XSG_TRY(host_slot_acquire(hs, &l.s));
  XSG_TRY(host_stage(l.s, data, len));
  uint64_t cnt = 0, nl = 0, nb = 0;
  const uint64_t* vlens = nullptr;
  const char* vbytes = nullptr;
  XSG_TRY(xsg_search(l.s->shard, XSG_LINES, &cnt));
  // out of the shard's pinned buffers (a large result arrives there by pinned copies, not by a pageable D2H)
  XSG_TRY(xsg_result_lines_view(l.s->shard, &vlens, &vbytes, nullptr, &nl, &nb));
  uint64_t* lens = static_cast<uint64_t*>(malloc(8 * std::max<uint64_t>(nl, 1)));
  char* buf = static_cast<char*>(malloc(std::max<uint64_t>(nb, 1)));
  if (!lens || !buf) {
    free(lens);
    free(buf);
    return fail(XSG_ENOMEM, "host allocation failed");
  }
  if (nl) memcpy(lens, vlens, 8 * nl);
  if (nb) memcpy(buf, vbytes, nb);
  *lengths = lens;
  *bytes = buf;
  *n = nl;
  *nbytes = nb;
  return XSG_OK;
}
