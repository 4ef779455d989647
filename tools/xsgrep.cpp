// xsgrep -- the reference's example/grep.cpp (PATTERN FILE, -c, -i; lines 23-82)
// on the MI355X engine, without boost::program_options.
//
//   xsgrep [-c] [-i] [-F] [-j THREADS] [-m METAFILE] PATTERN FILE|-
//
// -c  print only a count of matching lines   (grep.cpp:45-46 -> xs::count_lines)
// -i  ignore ASCII case                      (grep.cpp:47-48)
// otherwise print the matching lines, live, as they are found (grep.cpp:74-79).
#include <xsearch/tasks/gpu_searchers.h>
#include <xsearch/xsearch.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

int main(int argc, char** argv) {
  bool count = false, icase = false;
  int threads = 2;  // grep.cpp:21
  std::string meta, pattern, file;
  int pos = 0;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-c" || a == "--count") {
      count = true;
    } else if (a == "-i" || a == "--ignore-case") {
      icase = true;
    } else if (a == "-F" || a == "--fixed-strings") {
      setenv("XS_FORCE_LITERAL", "1", 1);  // like grep -F: never read the pattern as a regex
    } else if ((a == "-j" || a == "--threads") && i + 1 < argc) {
      threads = std::atoi(argv[++i]);
    } else if ((a == "-m" || a == "--meta") && i + 1 < argc) {
      meta = argv[++i];
    } else if (a == "-h" || a == "--help") {
      std::printf("usage: %s [-c] [-i] [-j THREADS] [-m METAFILE] PATTERN FILE\n", argv[0]);
      return 0;
    } else if (pos == 0) {
      pattern = a;
      ++pos;
    } else if (pos == 1) {
      file = a;
      ++pos;
    } else {
      std::fprintf(stderr, "unexpected argument '%s'\n", a.c_str());
      return 2;
    }
  }
  if (pos != 2) {
    std::fprintf(stderr, "usage: %s [-c] [-i] [-j THREADS] [-m METAFILE] PATTERN FILE\n", argv[0]);
    return 2;
  }
  try {
    std::ios::sync_with_stdio(false);
    if (file == "-") {
      // stdin (grep.cpp:37: "input file, stdin if '-' or empty"): no file to plan chunks on, so
      // read newline-aligned chunks here and hand each to the reference-style searcher functors
      // (include/xsearch/tasks/gpu_searchers.h), like Searcher::run_thread does with a reader.
      const uint32_t flags = xs::detail::pattern_flags(pattern, icase);
      xs::GpuLineSearcher<std::vector<char>> lines(pattern, 0, 1, flags);
      xs::GpuCountSearcher<std::vector<char>> counter(pattern, true, 0, 1, flags);
      const size_t target = 16u << 20;
      std::vector<char> buf;  // bytes read and not searched yet
      size_t want = target;
      uint64_t total = 0;
      bool eof = false;
      for (;;) {
        while (!eof && buf.size() < want) {
          const size_t at = buf.size();
          buf.resize(at + (1u << 20));
          const size_t got = std::fread(buf.data() + at, 1, 1u << 20, stdin);
          buf.resize(at + got);
          if (got == 0) eof = true;
        }
        if (buf.empty()) break;
        size_t cut = buf.size();
        if (!eof) {  // cut after the last newline; the rest opens the next chunk
          while (cut > 0 && buf[cut - 1] != '\n') --cut;
          if (cut == 0) {  // one line longer than the chunk target: keep reading
            want = buf.size() + target;
            continue;
          }
        }
        std::vector<char> chunk(buf.begin(), buf.begin() + (ptrdiff_t)cut);
        buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)cut);
        want = target;
        if (count) {
          if (auto c = counter(chunk)) total += *c;
        } else if (auto ls = lines(chunk)) {
          for (const auto& l : *ls) std::cout << l << '\n';
        }
      }
      if (count) std::cout << total << std::endl;
      return 0;
    }
    if (count) {
      auto searcher = meta.empty() ? xs::extern_search<xs::count_lines>(pattern, file, icase, threads)
                                   : xs::extern_search<xs::count_lines>(pattern, file, meta, icase, threads, threads);
      searcher->join();
      std::cout << searcher->getResult()->size() << std::endl;
    } else {
      auto searcher = meta.empty() ? xs::extern_search<xs::lines>(pattern, file, icase, threads)
                                   : xs::extern_search<xs::lines>(pattern, file, meta, icase, threads, threads);
      for (auto const& line : *searcher->getResult()) {
        std::cout << line << '\n';
      }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "xsgrep: %s\n", e.what());
    return 1;
  }
  // Everything is printed; what is left is tearing the HIP runtime down (streams, pinned memory, the device context):
  // 40-90 ms of a process that lives 0.2 s on a small file (profiles/r04_cli_start.txt).  A command-line tool leaves that
  // to the kernel, as grep leaves its buffers: flush and go.  (tools/my_grep.cpp, the README's program, returns normally.)
  std::cout.flush();
  std::fflush(stdout);
  std::_Exit(0);
}
