// ref_grep -- the reference's CPU path as a command line, for bench.py's `cli` block (the timed CPU baseline beside
// xsgrep / my_grep / GNU grep): what README.md:31-41's program does when the library underneath is the reference itself.
//
// TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything under oracle/): never linked or executed by the product.
// Built by oracle/Makefile into oracle/_ref/xsref_grep together with the reference's own
// src/string_search/simd_search.cpp, compiled unmodified from where it lies under /root/reference (findNext /
// findNextNewLine are ITS code); the per-chunk walk above them is the restatement of search_wrappers.h:187-207 (`line`)
// and :163-185 (`count`) in xs_oracle.c, routed through those primitives (xso_use_primitives).
//
// Pipeline (include/xsearch/Searcher.h:100-120: N worker threads, each read -> search -> result.add): T threads pull
// chunk numbers from a shared counter; a chunk is >= 16 MiB extended to just past the next '\n' (the layout of the
// reference's .meta fixtures, SURVEY 5.1 -- the same plan the GPU pipeline cuts); each thread preads its chunk into its
// own buffer (tasks/readers.h:39-48 reads into a fresh zero-filled vector per chunk: reusing the buffer is a kindness to
// the baseline), searches it, and the results are printed in file order.
//
//   xsref_grep [-c] PATTERN FILE [THREADS]
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "xs_oracle.h"

namespace xs::search::simd {  // simd_search.h:52,64 (the definitions are the reference's, linked in)
int64_t findNext(const char* pattern, size_t pattern_len, const char* str, size_t str_len, size_t shift);
int64_t findNextNewLine(const char* str, size_t str_len, size_t shift);
}  // namespace xs::search::simd

static int64_t ref_find_next(const char* pat, size_t plen, const char* str, size_t len, size_t shift) {
  return xs::search::simd::findNext(pat, plen, str, len, shift);
}
static int64_t ref_find_nl(const char* str, size_t len, size_t shift) { return xs::search::simd::findNextNewLine(str, len, shift); }

static bool pread_full(int fd, char* buf, uint64_t n, uint64_t off) {
  while (n) {
    const ssize_t r = pread(fd, buf, n, (off_t)off);
    if (r <= 0) return false;
    buf += r, off += (uint64_t)r, n -= (uint64_t)r;
  }
  return true;
}

int main(int argc, char** argv) {
  bool count_only = false;
  int a = 1;
  if (a < argc && strcmp(argv[a], "-c") == 0) count_only = true, ++a;
  if (argc - a < 2) {
    fprintf(stderr, "usage: %s [-c] PATTERN FILE [THREADS]\n", argv[0]);
    return 2;
  }
  const std::string pattern = argv[a];
  const char* path = argv[a + 1];
  const int T = argc - a > 2 ? std::max(1, atoi(argv[a + 2])) : 1;
  xso_use_primitives(ref_find_next, ref_find_nl);
  const int fd = open(path, O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat(fd, &st) != 0) {
    perror(path);
    return 1;
  }
  const uint64_t size = (uint64_t)st.st_size, target = 16u << 20;
  // newline-aligned plan
  std::vector<std::pair<uint64_t, uint64_t>> plan;
  {
    std::vector<char> probe(1 << 16);
    for (uint64_t pos = 0; pos < size;) {
      uint64_t end = pos + target;
      if (end >= size) {
        end = size;
      } else {
        uint64_t q = end - 1;
        bool found = false;
        while (q < size && !found) {
          const uint64_t n = std::min<uint64_t>(probe.size(), size - q);
          if (!pread_full(fd, probe.data(), n, q)) return 1;
          if (const void* hit = memchr(probe.data(), '\n', n)) {
            end = q + (uint64_t)((const char*)hit - probe.data()) + 1;
            found = true;
          }
          q += n;
        }
        if (!found) end = size;
      }
      plan.push_back({pos, end - pos});
      pos = end;
    }
  }
  std::atomic<uint64_t> next{0};
  std::mutex mu;
  std::condition_variable cv;
  std::map<uint64_t, std::string> done;  // chunk -> its output text (lines + '\n'), or its count as 8 raw bytes
  std::atomic<bool> failed{false};
  auto worker = [&] {
    std::vector<char> buf;
    std::vector<uint64_t> begin, length;
    for (;;) {
      const uint64_t i = next.fetch_add(1);
      if (i >= plan.size()) break;
      const uint64_t off = plan[i].first, len = plan[i].second;
      if (buf.size() < len) buf.resize(len);
      if (!pread_full(fd, buf.data(), len, off)) {
        failed = true;
        break;
      }
      std::string out;
      if (count_only) {
        const uint64_t c = xso_count(buf.data(), len, pattern.data(), pattern.size(), 1);
        out.assign(reinterpret_cast<const char*>(&c), 8);
      } else {
        const uint64_t n = xso_lines(buf.data(), len, pattern.data(), pattern.size(), nullptr, nullptr, 0);
        begin.resize(n), length.resize(n);
        xso_lines(buf.data(), len, pattern.data(), pattern.size(), begin.data(), length.data(), n);
        for (uint64_t k = 0; k < n; ++k) {
          out.append(buf.data() + begin[k], length[k]);
          out.push_back('\n');
        }
      }
      {
        std::lock_guard<std::mutex> g(mu);
        done.emplace(i, std::move(out));
      }
      cv.notify_one();
    }
  };
  std::vector<std::thread> threads;
  for (int t = 0; t < T; ++t) threads.emplace_back(worker);
  uint64_t total = 0;
  for (uint64_t i = 0; i < plan.size() && !failed; ++i) {  // print in file order as chunks complete
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done.count(i) != 0 || failed.load(); });
    if (failed) break;
    std::string out = std::move(done[i]);
    done.erase(i);
    lk.unlock();
    if (count_only) {
      uint64_t c;
      memcpy(&c, out.data(), 8);
      total += c;
    } else {
      fwrite(out.data(), 1, out.size(), stdout);
    }
  }
  for (std::thread& t : threads) t.join();
  if (failed) {
    fprintf(stderr, "read error\n");
    return 1;
  }
  if (count_only) printf("%llu\n", (unsigned long long)total);
  return 0;
}
