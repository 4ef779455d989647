// extern_search_cli.cpp -- drives the xs::extern_search C++ API exactly the way
// the reference's tests do (join + getResult()->size()/copyResultSafe(), or
// live iteration), and prints the result for tests/test_gpu_pipeline.py.
//
// usage: extern_search_cli <tag> <join|live> <pattern> <file> [meta|-] [threads] [readers]
#include <xsearch/xsearch.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

template <class Tag>
static int run(const std::string& how, const std::string& pattern, const std::string& file, const std::string& meta,
               int threads, int readers) {
  const bool icase = std::getenv("XS_IGNORE_CASE") != nullptr;  // test hook for the ignore_case argument
  auto res = meta.empty() ? xs::extern_search<Tag>(pattern, file, icase, threads)
                          : xs::extern_search<Tag>(pattern, file, meta, icase, threads, readers);
  using V = typename xs::Result<Tag>::value_type;
  constexpr bool is_count = std::is_same<Tag, xs::count>::value || std::is_same<Tag, xs::count_lines>::value;
  if (how == "join") {
    res->join();
    if (is_count) {
      std::cout << res->getResult()->size() << "\n";
    } else {
      auto v = res->getResult()->copyResultSafe();
      for (const V& x : v) std::cout << x << "\n";
    }
  } else {
    if (is_count) {
      V last{};
      for (auto i : *res->getResult()) last = i;
      std::cout << last << "\n";
    } else {
      for (auto const& x : *res->getResult()) std::cout << x << "\n";
    }
  }
  auto st = res->stats();
  std::fprintf(stderr, "stats: chunks=%llu bytes=%llu total=%.4fs read=%.4fs device=%.4fs\n",
               (unsigned long long)st.chunks, (unsigned long long)st.bytes_scanned, st.seconds_total, st.seconds_read,
               st.seconds_device);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s <tag> <join|live> <pattern> <file> [meta|-] [threads] [readers]\n", argv[0]);
    return 2;
  }
  const std::string tag = argv[1], how = argv[2], pattern = argv[3], file = argv[4];
  const std::string meta = argc > 5 && std::strcmp(argv[5], "-") != 0 ? argv[5] : "";
  const int threads = argc > 6 ? std::atoi(argv[6]) : 1;
  const int readers = argc > 7 ? std::atoi(argv[7]) : 1;
  try {
    if (tag == "count" || tag == "count_matches") return run<xs::count_matches>(how, pattern, file, meta, threads, readers);
    if (tag == "count_lines") return run<xs::count_lines>(how, pattern, file, meta, threads, readers);
    if (tag == "match_byte_offsets") return run<xs::match_byte_offsets>(how, pattern, file, meta, threads, readers);
    if (tag == "line_byte_offsets") return run<xs::line_byte_offsets>(how, pattern, file, meta, threads, readers);
    if (tag == "line_indices") return run<xs::line_indices>(how, pattern, file, meta, threads, readers);
    if (tag == "lines") return run<xs::lines>(how, pattern, file, meta, threads, readers);
    std::fprintf(stderr, "unknown tag %s\n", tag.c_str());
    return 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
