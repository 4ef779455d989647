// xsearch/tasks/aligned_reader.h -- a chunk reader with the call signature of the
// reference's reader task (include/xsearch/tasks/readers.h:29-54):
//
//   std::optional<DataT> operator()()        -- the next chunk, std::nullopt at the end
//
// so it satisfies the reference's ReaderC concept (include/xsearch/concepts.h:24-27)
// and plugs into its worker loop (include/xsearch/Searcher.h:100-120, read under the
// semaphore at :106) in place of FileReader.  The snapshot's FileReader cuts at a
// fixed 512 KiB (readers.h:32,44-46) and would lose every match or line that
// straddles a cut; this one hands out what the reference's own fixtures describe
// (test/files/*.meta): chunks of at least `chunk_bytes` that end just
// after a '\n' (the last one at EOF), i.e. independent units for the searcher
// functors of gpu_searchers.h.  With a metafile of an UNCOMPRESSED file the chunk
// table comes from it (compressed corpora go through xs::extern_search, which
// decodes in its reader threads).
//
// Thread-safe: several workers may call one shared instance concurrently (chunks
// are claimed with an atomic counter and read with pread).  Copies share the
// position, like copies of a stream handle.
#pragma once

#include <fcntl.h>
#include <unistd.h>
#include <xsg.h>

#include <atomic>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace xs {

template <class T = std::vector<char>>
class AlignedFileReader {
 public:
  struct Chunk {
    T data;
    uint64_t index = 0;       // position in the plan
    uint64_t offset = 0;      // byte offset of data[0] in the file
    uint64_t first_line = 0;  // 0-based index of the chunk's first line (metafile only, else UINT64_MAX)
  };

  explicit AlignedFileReader(const std::string& file_path, uint64_t chunk_bytes = 16u << 20,
                             const std::string& meta_file_path = std::string())
      : _s(std::make_shared<State>()) {
    xsg_file_chunk* chunks = nullptr;
    uint64_t n = 0;
    int r;
    if (meta_file_path.empty()) {
      r = xsg_plan_chunks(file_path.c_str(), chunk_bytes, &chunks, &n);
    } else {
      int32_t comp = 0;
      r = xsg_meta_read(meta_file_path.c_str(), &comp, &chunks, &n, nullptr, nullptr);
      if (r == XSG_OK && comp != XSG_COMPRESSION_NONE) {
        xsg_free(chunks);
        throw std::invalid_argument("xs::AlignedFileReader: '" + meta_file_path +
                                    "' describes a compressed file; use xs::extern_search for those");
      }
    }
    if (r != XSG_OK)
      throw std::runtime_error(std::string("xs::AlignedFileReader: ") + xsg_strerror(r) + " (" + xsg_last_error() + ")");
    _s->plan.assign(chunks, chunks + n);
    xsg_free(chunks);
    _s->fd = ::open(file_path.c_str(), O_RDONLY | O_CLOEXEC);
    if (_s->fd < 0) throw std::runtime_error("xs::AlignedFileReader: cannot open '" + file_path + "'");
  }

  // the reference's reader signature
  std::optional<T> operator()() {
    auto c = next();
    if (!c) return std::nullopt;
    return std::move(c->data);
  }

  // the same with the chunk's place in the file (what a result needs to become global)
  std::optional<Chunk> next() {
    const uint64_t i = _s->next.fetch_add(1);
    if (i >= _s->plan.size()) return std::nullopt;
    const xsg_file_chunk& fc = _s->plan[i];
    Chunk c;
    c.index = i;
    c.offset = fc.original_offset;
    c.first_line = fc.first_line;
    c.data.resize(fc.original_size);
    uint64_t got = 0;
    while (got < fc.original_size) {
      const ssize_t k = ::pread(_s->fd, c.data.data() + got, fc.original_size - got, (off_t)(fc.actual_offset + got));
      if (k <= 0) throw std::runtime_error("xs::AlignedFileReader: short read (file changed underneath?)");
      got += (uint64_t)k;
    }
    return c;
  }

  size_t num_chunks() const { return _s->plan.size(); }
  void rewind() { _s->next.store(0); }

 private:
  struct State {
    int fd = -1;
    std::vector<xsg_file_chunk> plan;
    std::atomic<uint64_t> next{0};
    ~State() {
      if (fd >= 0) ::close(fd);
    }
  };
  std::shared_ptr<State> _s;
};

}  // namespace xs
