// Compile-only: the reference's own call sites of xs::extern_search must compile
// unchanged against include/xsearch/xsearch.h (arguments re-typed from the cited lines).
#include <xsearch/xsearch.h>
#include <xsearch/tasks/gpu_searchers.h>

#include <algorithm>
#include <iostream>
#include <type_traits>

static const std::string pattern("Sherlock");
static const std::string file_path("test/files/sample.txt");
static const std::string meta_file_path("test/files/sample.meta");

static_assert(sizeof(&xs::detail::reference_routes_to_regex) > 0, "regex routing test present");

int callsites(int argc, char** argv) {
  (void)argc;
  {  // README.md:37
    auto searcher = xs::extern_search<xs::lines>(argv[1], argv[2], false, 1);
    for (auto const& line : *searcher->getResult()) std::cout << line << '\n';
  }
  {  // README.md:72 / checkit.cpp:4-5  (five arguments, the third is a path)
    auto res = xs::extern_search<xs::count>(pattern, file_path, meta_file_path, 4, 2);
    res->join();
    auto t = xs::extern_search<xs::lines>(argv[1], argv[2], argv[3], 4, 2);
    for (auto i : *t->getResult()) std::cout << i << std::endl;
  }
  {  // example/grep.cpp:69-79
    bool ignore_case = false;
    int num_threads = 2;
    auto searcher = xs::extern_search<xs::count_lines>(pattern, file_path, ignore_case, num_threads);
    searcher->join();
    std::cout << searcher->getResult()->size() << std::endl;
  }
  {  // test/src/xsearchTest.cpp:344-346
    auto res = xs::extern_search<xs::count_matches>(pattern, file_path, false, 1);
    res->join();
    if (res->getResult()->size() != 46) return 1;
  }
  {  // test/src/xsearchTest.cpp:447-452
    auto res = xs::extern_search<xs::line_byte_offsets>(pattern, file_path, false, 1);
    res->join();
    auto result = res->getResult()->copyResultSafe();
    std::sort(result.begin(), result.end());
    static_assert(std::is_same<decltype(result), std::vector<uint64_t>>::value, "offsets are uint64_t");
  }
  {  // test/src/xsearchTest.cpp:655-676
    auto res = xs::extern_search<xs::lines>(pattern, file_path, false, 4);
    std::vector<std::string> result{};
    for (auto r : *res->getResult()) result.push_back(r);
    std::sort(result.begin(), result.end());
  }
  {  // the read side of ResultTypes.h:91-115, and the regex of xsearchTest.cpp:9,391-396
    auto res = xs::extern_search<xs::match_byte_offsets>("She[r ]lock", file_path, false, 1);
    res->join();
    auto& r = *res->getResult();
    if (r.is_closed() && !r.empty()) {
      std::vector<uint64_t> all = r.get();
      if (all[0] != r[0] || r.at(0) != all.front()) return 1;
    }
  }
  {  // test/src/xsearchTest.cpp:735-739 (live count: the last value is the total)
    auto res = xs::extern_search<xs::count_matches>(pattern, file_path, false, 1);
    uint64_t result = 0;
    for (auto i : *res->getResult()) result = i;
    (void)result;
  }
  {  // test/src/xsearchTest.cpp:1227-1228 (six arguments)
    auto res = xs::extern_search<xs::count_matches>(pattern, file_path, meta_file_path, false, 1, 1);
    res->join();
    auto idx = xs::extern_search<xs::line_indices>(pattern, file_path, meta_file_path, true, 4, 4);
    auto mbo = xs::extern_search<xs::match_byte_offsets>(pattern, file_path, meta_file_path, false, 4, 4);
    (void)idx;
    (void)mbo;
  }
  {  // tasks/searchers.h:38-93 signatures (SearcherC, concepts.h:36-39)
    using strtype = std::vector<char>;
    xs::GpuIndexSearcher<strtype> a(pattern);
    xs::GpuLineIndexSearcher<strtype> b(pattern);
    xs::GpuLineSearcher<strtype> c(pattern);
    static_assert(std::is_move_constructible<xs::GpuLineSearcher<strtype>>::value, "SearcherC");
    strtype data;
    static_assert(std::is_same<decltype(a(data)), std::optional<std::vector<uint64_t>>>::value, "IndexSearcher");
    static_assert(std::is_same<decltype(b(data)), std::optional<std::vector<uint64_t>>>::value, "LineIndexSearcher");
    static_assert(std::is_same<decltype(c(data)), std::optional<std::vector<std::string>>>::value, "LineSearcher");
  }
  return 0;
}
