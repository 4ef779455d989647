// seam_cli.cpp -- drives the reference-style searcher functors
// (include/xsearch/tasks/gpu_searchers.h) the way Searcher::run_thread does
// (include/xsearch/Searcher.h:100-120): N threads pull chunks (std::vector<char>,
// the reference's xs::strtype) from ONE shared reader functor (tasks/aligned_reader.h)
// and call ONE shared const searcher functor concurrently.
//
// usage: seam_cli <index|line_index|line|count> <pattern> <file> <chunk_bytes> <threads>
// prints: per chunk "C <chunk_index> <n>" then the chunk-local values, in chunk order.
#include <xsearch/tasks/aligned_reader.h>
#include <xsearch/tasks/gpu_searchers.h>

#include <atomic>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <thread>

using strtype = std::vector<char>;

// Searcher::run_thread (Searcher.h:100-120): every worker calls the SHARED reader, then the SHARED searcher
template <class Functor, class Printer>
static void run(const Functor& fn, xs::AlignedFileReader<strtype>& reader, int threads, Printer print) {
  std::mutex mu;
  std::map<size_t, std::string> outs;
  auto worker = [&] {
    for (;;) {
      auto chunk = reader.next();  // operator()() plus the chunk's index, for the ordered printout
      if (!chunk) break;
      auto r = fn(chunk->data);  // shared const functor, concurrent calls
      std::string s = print(chunk->index, r);
      std::lock_guard<std::mutex> g(mu);
      outs[chunk->index] = std::move(s);
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
    xs::AlignedFileReader<strtype> chunks(file, target);  // the ReaderC-shaped reader of this repo
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
