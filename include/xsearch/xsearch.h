// xsearch/xsearch.h -- the one-call C++ API of lfreist/x-search, served by the
// MI355X scan engine (libxsg.so).  Header-only on top of the C ABI in xsg.h.
//
// Mirrors the surface the reference's users and tests call (the reference
// snapshot holds only call sites, no definition -- SURVEY 0 / 8b):
//
//   auto res = xs::extern_search<xs::count>(pattern, file, meta, threads, readers);   README.md:72
//   auto s   = xs::extern_search<xs::lines>(pattern, file, ignore_case, threads);     README.md:37, example/grep.cpp:69-76
//   res->join();  res->getResult()->size();                                           test/src/xsearchTest.cpp:344-346
//   auto v = res->getResult()->copyResultSafe();                                      test/src/xsearchTest.cpp:450,657
//   for (auto r : *res->getResult()) { ... }     // live, blocks per element          test/src/xsearchTest.cpp:735-739,888-890
//
// Tags: xs::count (== xs::count_matches), xs::count_lines, xs::match_byte_offsets,
// xs::line_byte_offsets, xs::line_indices, xs::lines (README.md:77-81).
//
// Semantics (include/xsearch/string_search/search_wrappers.h, bit-exact):
//   count tags   getResult()->size() is the count; live iteration yields the
//                running total after every chunk (last value == total)
//   vector tags  flat elements: global byte offsets / 0-based line indices /
//                lines without the trailing '\n'; published in file order
//   the searcher's threads are joined on destruction (README.md:91; Searcher.h:41)
//
// Errors (the reference defines none; a bad path spins there, readers.h:40-47):
// extern_search throws std::runtime_error when the file or metafile cannot be
// opened/parsed or no MI355X is usable; a failure inside a worker ends the
// search and is thrown by join() / by the iterator that runs into it.
//
// Regular expressions: a pattern the reference would hand to RE2 (utils/utils.h:17-25) is searched as a
// regex here too IF it is a fixed-length sequence of byte classes (`She[r ]lock`, `[0-9]{4}-\d\d`: XSG_FLAG_REGEX
// in xsg.h -- every regex the reference's tests use); any other regex makes extern_search throw
// std::invalid_argument instead of being searched as text with different results (XS_FORCE_LITERAL=1 searches
// it as plain text).  ignore_case folds ASCII letters only (what the reference's simd::toLower does).
//
// Environment: XS_DEVICE (HIP device index, default 0), XS_DEVICES ("0,1,2,3" or "all": one search fans its chunk
// ranges out over several devices of the node, one job each, results in file order), XS_CHUNK_BYTES (target
// chunk size without a metafile, default 16 MiB).
#pragma once

#include <xsg.h>

#include <cstdint>
#include <cstdlib>
#include <iterator>
#include <memory>
#include <mutex>
#include <regex>
#include <stdexcept>
#include <string>
#include <vector>

namespace xs {

// ---- tags -------------------------------------------------------------------
struct count {};
using count_matches = count;
struct count_lines {};
struct match_byte_offsets {};
struct line_byte_offsets {};
struct line_indices {};
struct lines {};

namespace detail {

template <class Tag>
struct traits;
template <>
struct traits<count> {
  using value_type = uint64_t;
  static constexpr uint32_t mode = XSG_COUNT_MATCHES;
  static constexpr bool is_count = true;
};
template <>
struct traits<count_lines> {
  using value_type = uint64_t;
  static constexpr uint32_t mode = XSG_COUNT_LINES;
  static constexpr bool is_count = true;
};
template <>
struct traits<match_byte_offsets> {
  using value_type = uint64_t;
  static constexpr uint32_t mode = XSG_MATCH_BYTE_OFFSETS;
  static constexpr bool is_count = false;
};
template <>
struct traits<line_byte_offsets> {
  using value_type = uint64_t;
  static constexpr uint32_t mode = XSG_LINE_BYTE_OFFSETS;
  static constexpr bool is_count = false;
};
template <>
struct traits<line_indices> {
  using value_type = uint64_t;
  static constexpr uint32_t mode = XSG_LINE_INDICES;
  static constexpr bool is_count = false;
};
template <>
struct traits<lines> {
  using value_type = std::string;
  static constexpr uint32_t mode = XSG_LINES;
  static constexpr bool is_count = false;
};

[[noreturn]] inline void throw_last(const char* what, int code) {
  throw std::runtime_error(std::string(what) + ": " + xsg_strerror(code) + " (" + xsg_last_error() + ")");
}

inline uint64_t env_u64(const char* name, uint64_t dflt) {
  const char* v = std::getenv(name);
  if (!v || !*v) return dflt;
  return std::strtoull(v, nullptr, 10);
}

// Devices of one search.  XS_DEVICES="0,1,2,3" or "all": the chunks fan out over these devices, one job each
// (one process, no collective: only counts and newline totals cross devices, on the host).  Unset: the single
// device XS_DEVICE (default 0).  One process per GPU under torch.distributed is x-search_amd/dist_search.py.
inline std::vector<int> device_list() {
  std::vector<int> out;
  const char* v = std::getenv("XS_DEVICES");
  if (!v || !*v) {
    out.push_back(static_cast<int>(env_u64("XS_DEVICE", 0)));
    return out;
  }
  if (std::string(v) == "all") {
    int n = 0;
    const int r = xsg_device_count(&n);
    if (r != XSG_OK) throw_last("xs::extern_search", r);
    for (int i = 0; i < n; ++i) out.push_back(i);
  } else {
    const char* p = v;
    while (*p) {
      char* e = nullptr;
      const long d = std::strtol(p, &e, 10);
      if (e == p || d < 0) throw std::invalid_argument(std::string("XS_DEVICES: cannot parse '") + v + "'");
      out.push_back(static_cast<int>(d));
      p = *e == ',' ? e + 1 : e;
      if (*e && *e != ',') throw std::invalid_argument(std::string("XS_DEVICES: cannot parse '") + v + "'");
    }
  }
  if (out.empty()) throw std::invalid_argument("XS_DEVICES names no device");
  return out;
}

// number of chunks of the search plan (metafile, or the newline-aligned plan of a plain file)
inline uint64_t count_chunks(const std::string& file_path, const char* meta, uint64_t chunk_bytes) {
  xsg_file_chunk* chunks = nullptr;
  uint64_t n = 0;
  int32_t comp = 0;
  const int r = meta ? xsg_meta_read(meta, &comp, &chunks, &n, nullptr, nullptr)
                     : xsg_plan_chunks(file_path.c_str(), chunk_bytes, &chunks, &n);
  if (r != XSG_OK) throw_last("xs::extern_search", r);
  xsg_free(chunks);
  return n;
}

// The reference decides per pattern whether it is a regular expression: it is one
// iff the pattern, read as a regex, does not match itself (utils/utils.h:17-25; an
// invalid regex counts as plain text).  Such a pattern goes to the GPU matchers -- the scan
// kernel's class-sequence matcher, or the automaton route for expressions of variable length --
// when one of them can decide it (xsg_regex_check) and is refused loudly otherwise, never
// searched as a literal with different results.  XS_FORCE_LITERAL=1 overrides.
inline bool reference_routes_to_regex(const std::string& pattern) {
  try {
    return !std::regex_match(pattern, std::regex("^" + pattern + "$"));
  } catch (const std::regex_error&) {
    return false;
  }
}

// XSG_FLAG_* for a pattern, routed as the reference routes it; throws std::invalid_argument for a
// regular expression the GPU matcher does not serve.
inline uint32_t pattern_flags(const std::string& pattern, bool ignore_case) {
  uint32_t flags = ignore_case ? XSG_FLAG_IGNORE_CASE : 0u;  // ASCII, as simd::toLower (string_utils.cpp:11-33)
  if (reference_routes_to_regex(pattern) && env_u64("XS_FORCE_LITERAL", 0) == 0) {
    if (xsg_regex_check(pattern.data(), pattern.size(), flags, nullptr, nullptr) != XSG_OK)
      throw std::invalid_argument("xs::extern_search: '" + pattern +
                                  "' is a regular expression for the reference (utils/utils.h:17-25) that the GPU "
                                  "matcher does not serve: " + xsg_last_error() +
                                  " (set XS_FORCE_LITERAL=1 to search it as plain text)");
    flags |= XSG_FLAG_REGEX;
  }
  return flags;
}

template <class T>
struct fetch;
template <>
struct fetch<uint64_t> {
  static uint64_t get(xsg_job* j, uint64_t i) {
    uint64_t v = 0;
    const int r = xsg_job_get_u64(j, i, 1, &v);
    if (r != XSG_OK) throw_last("xs::Result", r);
    return v;
  }
};
template <>
struct fetch<std::string> {
  static std::string get(xsg_job* j, uint64_t i) {
    const char* p = nullptr;
    uint64_t n = 0;
    const int r = xsg_job_get_line(j, i, &p, &n);
    if (r != XSG_OK) throw_last("xs::Result", r);
    return std::string(p, n);
  }
};

}  // namespace detail

// ---- the jobs of one search ---------------------------------------------------
// One xsg_job per device.  With several devices (XS_DEVICES) the chunks of the file fan out in contiguous
// ranges, device g taking chunks [g*C/G, (g+1)*C/G), so the concatenation of the jobs' results in
// device order is the file order.  Only two things cross jobs, both a single integer: a count tag's value is the
// sum of the jobs' counts, and without a metafile a line index needs the number of newlines in all earlier ranges.
namespace detail {

struct JobSet {
  std::vector<xsg_job*> jobs;
  bool is_count = false;
  bool add_newline_base = false;  // xs::line_indices without a metafile
  bool reduced = false;           // count tags, several devices, after join(): reduced_total is the exchanged sum
  bool reduced_via_rccl = false;
  uint64_t reduced_total = 0;
  mutable std::mutex _mu;
  mutable std::vector<uint64_t> _carry;

  // Finds the job that holds element i of the concatenated sequence; blocks until the element exists or every
  // job is closed.  Every job before the returned one is finished.
  bool locate(uint64_t i, size_t* jx, uint64_t* local) const {
    uint64_t base = 0;
    for (size_t k = 0; k < jobs.size(); ++k) {
      uint64_t avail = 0;
      int fin = 0;
      const int r = xsg_job_wait(jobs[k], i - base, &avail, &fin);
      if (r != XSG_OK) throw_last("xs::Result", r);
      if (avail > i - base) {
        *jx = k;
        *local = i - base;
        return true;
      }
      base += avail;  // job k is closed and holds `avail` elements in all
    }
    return false;
  }
  // what the finished jobs before job k add to a value of job k (constant once asked for: those jobs are closed)
  uint64_t carry(size_t k) const {
    if (k == 0 || !(is_count || add_newline_base)) return 0;
    std::lock_guard<std::mutex> g(_mu);
    if (_carry.size() < jobs.size()) _carry.assign(jobs.size(), UINT64_MAX);
    if (_carry[k] != UINT64_MAX) return _carry[k];
    uint64_t c = 0;
    for (size_t h = 0; h < k; ++h) {
      if (is_count) {
        uint64_t t = 0;
        const int r = xsg_job_total(jobs[h], &t);
        if (r != XSG_OK) throw_last("xs::Result", r);
        c += t;
      } else {
        xsg_job_stats st{};
        const int r = xsg_job_stats_get(jobs[h], &st);
        if (r != XSG_OK) throw_last("xs::Result", r);
        c += st.newlines;
      }
    }
    return _carry[k] = c;
  }
  template <class V>
  V get(size_t k, uint64_t local) const {
    return adjust(fetch<V>::get(jobs[k], local), k);
  }
  uint64_t adjust(uint64_t v, size_t k) const { return k == 0 ? v : v + carry(k); }
  std::string adjust(std::string v, size_t) const { return v; }
  // elements that can be read now without waiting, in order: all of every finished job, then what the first
  // unfinished one has so far
  uint64_t available(bool* all_finished) const {
    uint64_t n = 0;
    bool fin_all = true;
    for (xsg_job* j : jobs) {
      uint64_t a = 0;
      int fin = 0;
      const int r = xsg_job_poll(j, &a, &fin);
      if (r != XSG_OK) throw_last("xs::Result", r);
      n += a;
      if (!fin) {
        fin_all = false;
        break;
      }
    }
    if (all_finished) *all_finished = fin_all;
    return n;
  }
};

}  // namespace detail

// ---- result container (semantics of include/xsearch/ResultTypes.h:31-130) -----
template <class Tag>
class Result {
 public:
  using value_type = typename detail::traits<Tag>::value_type;

  // Blocking input iterator: `it != end()` waits until the element exists or
  // the search is closed (ResultTypes.h:48-60).
  class iterator {
   public:
    using iterator_category = std::input_iterator_tag;
    using value_type = typename Result::value_type;
    using difference_type = std::ptrdiff_t;
    using pointer = const value_type*;
    using reference = value_type;

    iterator(const detail::JobSet* set, uint64_t index, bool is_end) : _set(set), _index(index), _end(is_end) {}
    value_type operator*() const {
      size_t k = 0;
      uint64_t local = 0;
      if (!_set->locate(_index, &k, &local)) throw std::out_of_range("xs::Result::iterator");
      return _set->template get<value_type>(k, local);
    }
    iterator& operator++() {
      ++_index;
      return *this;
    }
    bool operator!=(const iterator& other) const { return !(*this == other); }
    bool operator==(const iterator& other) const {
      if (_end && other._end) return true;
      if (!_end && !other._end) return _index == other._index;
      const iterator& it = _end ? other : *this;
      size_t k = 0;
      uint64_t local = 0;
      return !it._set->locate(it._index, &k, &local);
    }

   private:
    const detail::JobSet* _set;
    uint64_t _index;
    bool _end;
  };

  explicit Result(const detail::JobSet* set) : _set(set) {}
  Result(const Result&) = delete;
  Result& operator=(const Result&) = delete;

  iterator begin() { return iterator(_set, 0, false); }
  iterator end() { return iterator(_set, 0, true); }

  // count tags: the count (so far; final after join()).  Vector tags: number of
  // elements available so far.  (test/src/xsearchTest.cpp:346)
  size_t size() const {
    if (_set->reduced) return static_cast<size_t>(_set->reduced_total);
    uint64_t sum = 0;
    for (xsg_job* j : _set->jobs) {
      uint64_t t = 0;
      const int r = xsg_job_total(j, &t);
      if (r != XSG_OK) detail::throw_last("xs::Result::size", r);
      sum += t;
    }
    return static_cast<size_t>(sum);
  }
  bool empty() const { return size() == 0; }

  // Thread-safe copy of everything available now (test/src/xsearchTest.cpp:450).
  // Count tags: the running totals published so far (one per chunk).
  std::vector<value_type> copyResultSafe() const {
    const uint64_t n = _set->available(nullptr);
    std::vector<value_type> out;
    out.reserve(n);
    for (uint64_t i = 0; i < n; ++i) out.push_back((*this)[i]);
    return out;
  }
  // the read side of ResultTypes.h:91-115 (get / operator[] / at / is_closed); elements are flat
  // values here (the reference's containers hold one partial result per chunk, its API layer flattens)
  std::vector<value_type> get() const { return copyResultSafe(); }
  value_type operator[](size_t index) const {
    size_t k = 0;
    uint64_t local = 0;
    if (!_set->locate(index, &k, &local)) throw std::out_of_range("xs::Result::operator[]");
    return _set->template get<value_type>(k, local);
  }
  value_type at(size_t index) const {
    if (index >= _set->available(nullptr)) throw std::out_of_range("xs::Result::at");
    return (*this)[index];
  }
  bool is_closed() const {
    bool fin = false;
    _set->available(&fin);
    return fin;
  }

 private:
  const detail::JobSet* _set;
};

// ---- the handle xs::extern_search returns --------------------------------------
template <class Tag>
class ExternSearcher {
 public:
  using ResultT = Result<Tag>;

  ExternSearcher(const std::string& pattern, const std::string& file_path, const std::string& meta_file_path,
                 bool ignore_case, int num_threads, int num_max_readers) {
    xsg_job_opts o;
    xsg_job_opts_init(&o);
    o.pattern_flags = detail::pattern_flags(pattern, ignore_case);
    o.mode = detail::traits<Tag>::mode;
    o.num_threads = num_threads < 1 ? 1 : num_threads;
    o.num_max_readers = num_max_readers < 1 ? 1 : num_max_readers;
    o.chunk_bytes = detail::env_u64("XS_CHUNK_BYTES", 16u << 20);
    const char* meta = meta_file_path.empty() ? nullptr : meta_file_path.c_str();
    _set.is_count = detail::traits<Tag>::is_count;
    _set.add_newline_base = o.mode == XSG_LINE_INDICES && meta == nullptr;
    const std::vector<int> devices = detail::device_list();
    try {
      if (devices.size() == 1) {
        o.device = devices[0];
        start(pattern, file_path, meta, o);
      } else {
        // contiguous chunk ranges, one job per device; devices left without a chunk start no job
        const uint64_t nchunks = detail::count_chunks(file_path, meta, o.chunk_bytes);
        const uint64_t G = devices.size();
        for (uint64_t g = 0; g < G; ++g) {
          o.device = devices[g];
          o.chunk_begin = g * nchunks / G;
          o.chunk_end = (g + 1) * nchunks / G;
          if (o.chunk_end > o.chunk_begin || (nchunks == 0 && g == 0)) start(pattern, file_path, meta, o);
        }
      }
    } catch (...) {
      for (xsg_job* j : _set.jobs) xsg_job_destroy(j);
      throw;
    }
    _result.reset(new ResultT(&_set));
  }
  ~ExternSearcher() {
    for (xsg_job* j : _set.jobs) xsg_job_destroy(j);  // joins the workers (README.md:91)
  }
  ExternSearcher(const ExternSearcher&) = delete;
  ExternSearcher& operator=(const ExternSearcher&) = delete;
  ExternSearcher(ExternSearcher&&) = delete;
  ExternSearcher& operator=(ExternSearcher&&) = delete;

  // Blocks until all workers are done (README.md:84); throws the first worker error.
  void join() {
    int first = XSG_OK;
    std::string msg;
    for (xsg_job* j : _set.jobs) {
      const int r = xsg_job_join(j);
      if (r != XSG_OK && first == XSG_OK) {
        first = r;
        msg = xsg_last_error();
      }
    }
    if (first != XSG_OK)
      throw std::runtime_error(std::string("xs::ExternSearcher::join: ") + xsg_strerror(first) + " (" + msg + ")");
    // several devices: the count is the one thing they exchange -- an RCCL all-reduce of the per-device totals
    // (xsg_jobs_reduce_total; host addition when there is no librccl or a device is listed twice)
    if (_set.is_count && _set.jobs.size() > 1 && !_set.reduced) {
      uint64_t total = 0;
      int via = 0;
      const int r = xsg_jobs_reduce_total(_set.jobs.data(), (int)_set.jobs.size(), &total, &via);
      if (r != XSG_OK) detail::throw_last("xs::ExternSearcher::join", r);
      _set.reduced_total = total;
      _set.reduced_via_rccl = via != 0;
      _set.reduced = true;
    }
  }
  // after join(): did the devices' counts meet over RCCL (true) or were they added on the host (false)?
  bool reduced_over_rccl() const { return _set.reduced && _set.reduced_via_rccl; }
  ResultT* getResult() { return _result.get(); }
  bool running() const { return !_result->is_closed(); }
  size_t num_devices() const { return _set.jobs.size(); }
  // summed over the devices; the seconds are those of the slowest device
  xsg_job_stats stats() const {
    xsg_job_stats sum{};
    for (xsg_job* j : _set.jobs) {
      xsg_job_stats st{};
      xsg_job_stats_get(j, &st);
      sum.chunks += st.chunks;
      sum.bytes_read += st.bytes_read;
      sum.bytes_scanned += st.bytes_scanned;
      sum.newlines += st.newlines;
      sum.plan_chunks = st.plan_chunks;
      sum.seconds_total = st.seconds_total > sum.seconds_total ? st.seconds_total : sum.seconds_total;
      sum.seconds_read += st.seconds_read;
      sum.seconds_decompress += st.seconds_decompress;
      sum.seconds_device += st.seconds_device;
    }
    return sum;
  }

 private:
  void start(const std::string& pattern, const std::string& file_path, const char* meta, const xsg_job_opts& o) {
    xsg_job* j = nullptr;
    const int r = xsg_job_start(pattern.data(), pattern.size(), file_path.c_str(), meta, &o, &j);
    if (r != XSG_OK) detail::throw_last("xs::extern_search", r);
    _set.jobs.push_back(j);
  }
  detail::JobSet _set;
  std::unique_ptr<ResultT> _result;
};

// ---- one-call API (overloads are told apart by arity) -----------------
template <class Tag>
std::shared_ptr<ExternSearcher<Tag>> extern_search(const std::string& pattern, const std::string& file_path,
                                                   bool ignore_case = false, int num_threads = 1) {
  // no reader count in this overload: one reader thread per device worker keeps the workers fed
  return std::make_shared<ExternSearcher<Tag>>(pattern, file_path, std::string(), ignore_case, num_threads, num_threads);
}
// a C string in third place is a metafile path, never a bool
template <class Tag>
std::shared_ptr<ExternSearcher<Tag>> extern_search(const std::string& pattern, const std::string& file_path,
                                                   const char* meta_file_path) {
  return std::make_shared<ExternSearcher<Tag>>(pattern, file_path, std::string(meta_file_path ? meta_file_path : ""),
                                               false, 1, 1);
}
template <class Tag>
std::shared_ptr<ExternSearcher<Tag>> extern_search(const std::string& pattern, const std::string& file_path,
                                                   const std::string& meta_file_path, bool ignore_case,
                                                   int num_threads, int num_max_readers) {
  return std::make_shared<ExternSearcher<Tag>>(pattern, file_path, meta_file_path, ignore_case, num_threads,
                                               num_max_readers);
}
template <class Tag>
std::shared_ptr<ExternSearcher<Tag>> extern_search(const std::string& pattern, const std::string& file_path,
                                                   const std::string& meta_file_path, int num_threads,
                                                   int num_max_readers) {
  return std::make_shared<ExternSearcher<Tag>>(pattern, file_path, meta_file_path, false, num_threads,
                                               num_max_readers);
}

}  // namespace xs
