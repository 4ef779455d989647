// CPU self-test of include/xsearch/tasks/aligned_reader.h (no GPU: the reader only plans and reads).
// usage: reader_selftest <file> <chunk_bytes> [metafile]
// Checks: the concatenation of the chunks is the file; every chunk but the last ends with '\n' and is at least
// chunk_bytes long; operator()() has the reference's reader signature (concepts.h:24-27) and several threads can
// share one instance; a copy shares the position.
#include <xsearch/tasks/aligned_reader.h>

#include <cstdio>
#include <fstream>
#include <iterator>
#include <map>
#include <mutex>
#include <thread>
#include <type_traits>

using strtype = std::vector<char>;
static_assert(std::is_move_constructible<xs::AlignedFileReader<strtype>>::value, "ReaderC: move constructible");
static_assert(std::is_same<decltype(std::declval<xs::AlignedFileReader<strtype>&>()()), std::optional<strtype>>::value,
              "ReaderC: std::optional<DataT> operator()()");

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string path = argv[1];
  const uint64_t target = std::stoull(argv[2]);
  const std::string meta = argc > 3 ? argv[3] : "";
  std::ifstream f(path, std::ios::binary);
  const std::vector<char> all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  try {
    xs::AlignedFileReader<strtype> reader(path, target, meta);
    std::map<uint64_t, xs::AlignedFileReader<strtype>::Chunk> got;
    std::mutex mu;
    std::vector<std::thread> ts;
    for (int t = 0; t < 4; ++t)
      ts.emplace_back([&] {
        xs::AlignedFileReader<strtype> mine = reader;  // a copy shares the position
        while (auto c = mine.next()) {
          std::lock_guard<std::mutex> g(mu);
          got.emplace(c->index, std::move(*c));
        }
      });
    for (auto& t : ts) t.join();
    if (got.size() != reader.num_chunks()) return std::printf("chunk count %zu != %zu\n", got.size(), reader.num_chunks()), 1;
    uint64_t at = 0;
    for (auto& kv : got) {
      const auto& c = kv.second;
      if (c.offset != at) return std::printf("chunk %llu starts at %llu, expected %llu\n", (unsigned long long)kv.first,
                                             (unsigned long long)c.offset, (unsigned long long)at), 1;
      if (at + c.data.size() > all.size() || !std::equal(c.data.begin(), c.data.end(), all.begin() + (long)at))
        return std::printf("chunk %llu: bytes differ from the file\n", (unsigned long long)kv.first), 1;
      at += c.data.size();
      const bool last = kv.first + 1 == got.size();
      if (!last && (c.data.empty() || c.data.back() != '\n')) return std::printf("chunk %llu does not end with a newline\n", (unsigned long long)kv.first), 1;
      if (!last && meta.empty() && c.data.size() < target) return std::printf("chunk %llu shorter than the target\n", (unsigned long long)kv.first), 1;
    }
    if (at != all.size()) return std::printf("chunks cover %llu of %zu bytes\n", (unsigned long long)at, all.size()), 1;
    if (reader()) return std::printf("reader not exhausted\n"), 1;
    reader.rewind();
    auto first = reader();
    if (!all.empty() && (!first || first->size() != got.begin()->second.data.size())) return std::printf("rewind failed\n"), 1;
    std::printf("reader selftest ok: %zu chunks, %zu bytes\n", got.size(), all.size());
  } catch (const std::exception& e) {
    std::printf("error: %s\n", e.what());
    return 1;
  }
  return 0;
}
