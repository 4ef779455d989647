// seam_cli.cpp -- drives the reference-style searcher functors
// (include/xsearch/tasks/gpu_searchers.h) the way Searcher::run_thread does
// (include/xsearch/Searcher.h:100-120): N threads pull chunks (std::vector<char>,
// the reference's xs::strtype) and call ONE shared const functor concurrently.
//
// usage: seam_cli <index|line_index|line|count> <pattern> <file> <chunk_bytes> <threads>
// prints: per chunk "C <chunk_index> <n>" then the chunk-local values, in chunk order.
#include <xsearch/tasks/gpu_searchers.h>

#include <atomic>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <thread>

using strtype = std::vector<char>;

static std::vector<strtype> newline_aligned_chunks(const std::string& path, size_t target) {
  std::ifstream f(path, std::ios::binary);
  std::vector<char> all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  std::vector<strtype> out;
  size_t pos = 0;
  while (pos < all.size()) {
    size_t end = pos + target;
    if (end >= all.size()) {
      end = all.size();
    } else {
      size_t q = end - 1;
      while (q < all.size() && all[q] != '\n') ++q;
      end = q < all.size() ? q + 1 : all.size();
    }
    out.emplace_back(all.begin() + pos, all.begin() + end);
    pos = end;
  }
  return out;
}

template <class Functor, class Printer>
static void run(const Functor& fn, const std::vector<strtype>& chunks, int threads, Printer print) {
  std::atomic<size_t> next{0};
  std::mutex mu;
  std::map<size_t, std::string> outs;
  auto worker = [&] {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= chunks.size()) break;
      auto r = fn(chunks[i]);  // shared const functor, concurrent calls
      std::string s = print(i, r);
      std::lock_guard<std::mutex> g(mu);
      outs[i] = std::move(s);
    }
  };
  std::vector<std::thread> ts;
  for (int t = 0; t < threads; ++t) ts.emplace_back(worker);
  for (auto& t : ts) t.join();
  for (auto& kv : outs) std::cout << kv.second;
}

int main(int argc, char** argv) {
  if (argc != 6) {
    std::fprintf(stderr, "usage: %s <index|line_index|line|count> <pattern> <file> <chunk_bytes> <threads>\n", argv[0]);
    return 2;
  }
  const std::string what = argv[1], pattern = argv[2], file = argv[3];
  const size_t target = std::stoull(argv[4]);
  const int threads = std::atoi(argv[5]);
  try {
    auto chunks = newline_aligned_chunks(file, target);
    auto print_u64 = [](size_t i, const std::optional<std::vector<uint64_t>>& r) {
      std::string s = "C " + std::to_string(i) + " " + std::to_string(r ? r->size() : 0) + "\n";
      if (r)
        for (uint64_t v : *r) s += std::to_string(v) + "\n";
      return s;
    };
    if (what == "index") {
      run(xs::GpuIndexSearcher<strtype>(pattern), chunks, threads, print_u64);
    } else if (what == "line_index") {
      run(xs::GpuLineIndexSearcher<strtype>(pattern), chunks, threads, print_u64);
    } else if (what == "line") {
      run(xs::GpuLineSearcher<strtype>(pattern), chunks, threads,
          [](size_t i, const std::optional<std::vector<std::string>>& r) {
            std::string s = "C " + std::to_string(i) + " " + std::to_string(r ? r->size() : 0) + "\n";
            if (r)
              for (const auto& l : *r) s += l + "\n";
            return s;
          });
    } else if (what == "count") {
      run(xs::GpuCountSearcher<strtype>(pattern, true), chunks, threads,
          [](size_t i, const std::optional<uint64_t>& r) {
            return "C " + std::to_string(i) + " " + std::to_string(r ? *r : 0) + "\n";
          });
    } else {
      return 2;
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
