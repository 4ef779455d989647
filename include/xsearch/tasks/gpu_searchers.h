// xsearch/tasks/gpu_searchers.h -- MI355X-backed searcher functors with the call
// signature of the reference's task adapters
// (include/xsearch/tasks/searchers.h:38-93):
//
//   std::optional<PartRes> operator()(const DataT& data) const
//
// so they satisfy the reference's SearcherC concept (include/xsearch/concepts.h:36-39)
// and plug into its worker loop (include/xsearch/Searcher.h:100-120) in place of
// IndexSearcher / LineIndexSearcher / LineSearcher.  Like those, they return
// CHUNK-LOCAL results and std::nullopt when the chunk has no hit
// (searchers.h:51-53), and one shared const instance may be called from many
// threads at once (Searcher.h:110): each concurrent call uses its own pinned
// staging buffer, device buffer and HIP stream inside libxsg.
//
// DataT: anything with data() -> convertible to const char* and size()
// (the reference's DefaultDataC, concepts.h:17-22; xs::strtype = std::vector<char>).
#pragma once

#include <xsg.h>

#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace xs {

namespace gpu_detail {

struct Handle {
  xsg_host_searcher* hs = nullptr;
  Handle(const std::string& pattern, int device, int max_slots, uint32_t flags) {
    const int r = xsg_host_searcher_create(device, pattern.data(), pattern.size(), flags, max_slots, &hs);
    if (r != XSG_OK)
      throw std::runtime_error(std::string("xs::Gpu*Searcher: ") + xsg_strerror(r) + " (" + xsg_last_error() + ")");
  }
  ~Handle() { xsg_host_searcher_destroy(hs); }
  Handle(const Handle&) = delete;
  Handle& operator=(const Handle&) = delete;
};

[[noreturn]] inline void raise(int r) {
  throw std::runtime_error(std::string("xs::Gpu*Searcher: ") + xsg_strerror(r) + " (" + xsg_last_error() + ")");
}

template <class T>
std::optional<std::vector<uint64_t>> offsets(const Handle& h, uint32_t mode, const T& data) {
  uint64_t* out = nullptr;
  uint64_t n = 0;
  const int r = xsg_host_offsets(h.hs, mode, data.data(), data.size(), &out, &n);
  if (r != XSG_OK) raise(r);
  if (n == 0) {
    xsg_free(out);
    return {};
  }
  std::vector<uint64_t> v(out, out + n);
  xsg_free(out);
  return std::make_optional(std::move(v));
}

}  // namespace gpu_detail

// replaces IndexSearcher (searchers.h:38-59): search::byte_offsets_match(data, pattern, false)
template <class T = std::vector<char>>
class GpuIndexSearcher {
 public:
  explicit GpuIndexSearcher(std::string pattern, int device = 0, int max_concurrent = 8, uint32_t flags = 0)
      : _h(std::make_shared<gpu_detail::Handle>(pattern, device, max_concurrent, flags)) {}
  std::optional<std::vector<uint64_t>> operator()(const T& data) const {
    return gpu_detail::offsets(*_h, XSG_MATCH_BYTE_OFFSETS, data);
  }

 private:
  std::shared_ptr<gpu_detail::Handle> _h;
};

// replaces LineIndexSearcher (searchers.h:61-76): search::byte_offsets_line(data, pattern)
template <class T = std::vector<char>>
class GpuLineIndexSearcher {
 public:
  explicit GpuLineIndexSearcher(std::string pattern, int device = 0, int max_concurrent = 8, uint32_t flags = 0)
      : _h(std::make_shared<gpu_detail::Handle>(pattern, device, max_concurrent, flags)) {}
  std::optional<std::vector<uint64_t>> operator()(const T& data) const {
    return gpu_detail::offsets(*_h, XSG_LINE_BYTE_OFFSETS, data);
  }

 private:
  std::shared_ptr<gpu_detail::Handle> _h;
};

// replaces LineSearcher (searchers.h:78-93): search::line(data, pattern)
template <class T = std::vector<char>>
class GpuLineSearcher {
 public:
  explicit GpuLineSearcher(std::string pattern, int device = 0, int max_concurrent = 8, uint32_t flags = 0)
      : _h(std::make_shared<gpu_detail::Handle>(pattern, device, max_concurrent, flags)) {}
  std::optional<std::vector<std::string>> operator()(const T& data) const {
    uint64_t* lens = nullptr;
    char* bytes = nullptr;
    uint64_t n = 0, nb = 0;
    const int r = xsg_host_lines(_h->hs, data.data(), data.size(), &lens, &bytes, &n, &nb);
    if (r != XSG_OK) gpu_detail::raise(r);
    std::vector<std::string> v;
    v.reserve(n);
    uint64_t at = 0;
    for (uint64_t i = 0; i < n; ++i) {
      v.emplace_back(bytes + at, lens[i]);
      at += lens[i];
    }
    xsg_free(lens);
    xsg_free(bytes);
    if (v.empty()) return {};
    return v;
  }

 private:
  std::shared_ptr<gpu_detail::Handle> _h;
};

// search::count(data, pattern, skip_to_nl) as a functor (no reference adapter exists for it)
template <class T = std::vector<char>>
class GpuCountSearcher {
 public:
  explicit GpuCountSearcher(std::string pattern, bool skip_to_nl = true, int device = 0, int max_concurrent = 8,
                            uint32_t flags = 0)
      : _h(std::make_shared<gpu_detail::Handle>(pattern, device, max_concurrent, flags)), _skip(skip_to_nl) {}
  std::optional<uint64_t> operator()(const T& data) const {
    uint64_t c = 0;
    const int r = xsg_host_count(_h->hs, data.data(), data.size(), _skip ? 1 : 0, &c);
    if (r != XSG_OK) gpu_detail::raise(r);
    if (c == 0) return {};
    return c;
  }

 private:
  std::shared_ptr<gpu_detail::Handle> _h;
  bool _skip;
};

}  // namespace xs
