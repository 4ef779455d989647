// pipeline_sanitize.cpp -- the host pipeline (x-search_amd/csrc/xsg_file.cpp: readers -> pinned queue -> device
// workers -> ordered publication) and the blocking result iterator of include/xsearch/xsearch.h under ThreadSanitizer /
// AddressSanitizer, on a box without a GPU: the device is tests/cpp/device_double.cpp (oracle-backed; test
// infrastructure).  The reference keeps such builds for its own pipeline (Makefile:30-38; the iterator they would
// exercise is include/xsearch/ResultTypes.h:48-60, the worker loop include/xsearch/Searcher.h:100-120).
//
// Cases: (1) every tag, join + live iteration, 1..4 workers x 1..4 readers, results equal to a single-threaded pass;
// (2) several jobs at once on the shared slot / buffer pools; (3) handles destroyed while the search is running and
// while a consumer thread is blocked in the iterator's wait; (4) a consumer that stops early; (5) a metafile job with the
// built-in LZ4 codec; (6) the XS_DEVICES fan-out with the host-side sum.
#include <xsearch/xsearch.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <thread>

extern "C" long xsg_double_live_allocations(void);

static int g_fail = 0;
#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++g_fail;                                                           \
    }                                                                     \
  } while (0)

static std::string make_corpus(const std::string& path, size_t bytes, unsigned seed) {
  static const char* words[] = {"the", "of", "and", "Sherlock", "Holmes", "she", "lock", "a", "to", "in", "that", "street", "She"};
  std::mt19937 rng(seed);
  std::string s;
  s.reserve(bytes + 64);
  while (s.size() < bytes) {
    const unsigned r = rng();
    s += (r % 997 == 0) ? "Sherlock" : words[r % 13];
    s += ((r >> 16) % 6 == 0) ? '\n' : ' ';
  }
  s.back() = '\n';
  std::ofstream(path, std::ios::binary).write(s.data(), (std::streamsize)s.size());
  return s;
}

template <class Tag>
static std::vector<typename xs::Result<Tag>::value_type> joined(const std::string& pat, const std::string& file, int threads,
                                                                int readers, const std::string& meta = "") {
  auto h = meta.empty() ? xs::extern_search<Tag>(pat, file, false, threads)
                        : xs::extern_search<Tag>(pat, file, meta, false, threads, readers);
  h->join();
  return h->getResult()->copyResultSafe();
}
template <class Tag>
static std::vector<typename xs::Result<Tag>::value_type> live(const std::string& pat, const std::string& file, int threads) {
  auto h = xs::extern_search<Tag>(pat, file, false, threads);
  std::vector<typename xs::Result<Tag>::value_type> out;
  for (auto const& x : *h->getResult()) out.push_back(x);  // blocks until the next element exists or the search is closed
  return out;
}
template <class T>
static std::vector<T> sorted(std::vector<T> v) {
  std::sort(v.begin(), v.end());
  return v;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const std::string file = dir + "/xsg_sanitize_corpus.txt";
  // small chunks so that a few MB make dozens of them: the queues turn over many times
  setenv("XS_CHUNK_BYTES", "65536", 1);
  make_corpus(file, 3u << 20, 1234);
  const std::string pat = "Sherlock";

  // ---- (1) every tag against the single-threaded pass
  const auto m1 = joined<xs::match_byte_offsets>(pat, file, 1, 1);
  const auto l1 = joined<xs::line_byte_offsets>(pat, file, 1, 1);
  const auto i1 = joined<xs::line_indices>(pat, file, 1, 1);
  const auto s1 = joined<xs::lines>(pat, file, 1, 1);
  CHECK(m1.size() > 100 && l1.size() > 100 && i1.size() == l1.size() && s1.size() == l1.size());
  uint64_t c1 = 0, cl1 = 0;
  {
    auto h = xs::extern_search<xs::count>(pat, file, false, 1);
    h->join();
    c1 = h->getResult()->size();
    auto g = xs::extern_search<xs::count_lines>(pat, file, false, 1);
    g->join();
    cl1 = g->getResult()->size();
  }
  CHECK(c1 == m1.size() && cl1 == l1.size());
  for (int threads = 2; threads <= 4; ++threads) {
    CHECK(sorted(joined<xs::match_byte_offsets>(pat, file, threads, threads)) == m1);
    CHECK(sorted(live<xs::line_byte_offsets>(pat, file, threads)) == l1);
    CHECK(sorted(live<xs::line_indices>(pat, file, threads)) == i1);
    CHECK(sorted(joined<xs::lines>(pat, file, threads, 2)) == sorted(s1));
    const auto running = live<xs::count>(pat, file, threads);  // running totals; the last one is the count
    CHECK(!running.empty() && running.back() == c1);
  }

  // ---- (2) several jobs at once, sharing the per-process slot and buffer pools
  {
    std::vector<std::thread> ts;
    std::atomic<int> ok{0};
    for (int k = 0; k < 6; ++k)
      ts.emplace_back([&, k] {
        if (k % 3 == 0) ok += sorted(joined<xs::match_byte_offsets>(pat, file, 2, 2)) == m1;
        if (k % 3 == 1) ok += sorted(live<xs::line_indices>(pat, file, 3)) == i1;
        if (k % 3 == 2) ok += live<xs::count>(pat, file, 2).back() == c1;
      });
    for (auto& t : ts) t.join();
    CHECK(ok == 6);
  }

  // ---- (3) handles dropped while the search runs; a consumer blocked in the iterator when the handle goes away
  for (int round = 0; round < 8; ++round) {
    auto h = xs::extern_search<xs::lines>(pat, file, false, 1 + round % 3);
    if (round & 1) std::this_thread::sleep_for(std::chrono::microseconds(200 * round));
    h.reset();  // "the threads started for the search are joined on destruction" (README.md:91)
  }
  {
    auto h = xs::extern_search<xs::match_byte_offsets>(pat, file, false, 2);
    std::atomic<size_t> seen{0};
    std::thread consumer([&] {
      for (auto const& x : *h->getResult()) {
        (void)x;
        ++seen;
      }
    });
    consumer.join();  // the iterator ends when the search is closed
    CHECK(seen == m1.size());
  }

  // ---- (4) a consumer that stops early, then the handle is destroyed with results unread
  {
    auto h = xs::extern_search<xs::line_byte_offsets>(pat, file, false, 3);
    size_t n = 0;
    for (auto const& x : *h->getResult()) {
      (void)x;
      if (++n == 5) break;
    }
    CHECK(n == 5);
  }

  // ---- (5) metafile + LZ4 blocks (the built-in codec), readers decode while workers search
  {
    const std::string meta = file + ".meta", data = file + ".xslz4";
    CHECK(xsg_meta_write(file.c_str(), meta.c_str(), data.c_str(), XSG_COMPRESSION_LZ4, 65536, 500, 0) == XSG_OK);
    CHECK(sorted(joined<xs::match_byte_offsets>(pat, data, 3, 3, meta)) == m1);
    CHECK(sorted(joined<xs::line_indices>(pat, data, 2, 4, meta)) == i1);
    std::remove(meta.c_str());
    std::remove(data.c_str());
  }

  // ---- (6) fan-out over two devices of the double, counts added on the host
  {
    setenv("XS_DEVICES", "0,1", 1);
    auto h = xs::extern_search<xs::count>(pat, file, false, 2);
    h->join();
    CHECK(h->getResult()->size() == c1);
    CHECK(h->num_devices() == 2 && !h->reduced_over_rccl());
    CHECK(sorted(joined<xs::match_byte_offsets>(pat, file, 2, 2)) == m1);
    unsetenv("XS_DEVICES");
  }
  std::remove(file.c_str());
  if (g_fail) {
    std::fprintf(stderr, "%d check(s) failed\n", g_fail);
    return 1;
  }
  std::printf("pipeline under the sanitizer: ok (%zu matches, %zu matching lines, %ld allocations of the double alive in its pools)\n",
              m1.size(), l1.size(), xsg_double_live_allocations());
  return 0;
}
