// Self-test of the built-in LZ4 block codec (x-search_amd/csrc/xsg_lz4.h), built with
// -fsanitize=address,undefined by tests/cpp/Makefile and run by tests/test_file_host.py:
//  * encode -> decode round trips on empty, tiny, repetitive, text-like and random inputs
//  * interoperability with the host's liblz4 in both directions, when one can be dlopen'ed
//  * the decoder on truncated and bit-flipped blocks: an error or some output, never an
//    out-of-bounds access (the sanitizers watch)
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../x-search_amd/csrc/xsg_lz4.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 11);
}

static bool same(const uint8_t* a, const uint8_t* b, size_t n) { return n == 0 || memcmp(a, b, n) == 0; }

static std::vector<uint8_t> make_input(int kind, size_t n) {
  std::vector<uint8_t> v(n);
  static const char* words[] = {"the ", "Sherlock ", "Holmes ", "detective ", "street\n", "a ", "of ", "and ", "lock", "She"};
  switch (kind) {
    case 0:  // random bytes: incompressible
      for (auto& b : v) b = (uint8_t)rnd();
      break;
    case 1:  // one byte repeated: long overlapping matches
      for (auto& b : v) b = 'a';
      break;
    case 2: {  // words: text-like
      size_t i = 0;
      while (i < n) {
        const char* w = words[rnd() % 10];
        for (; *w && i < n; ++w) v[i++] = (uint8_t)*w;
      }
      break;
    }
    case 3:  // short period
      for (size_t i = 0; i < n; ++i) v[i] = (uint8_t)("abcab"[i % 5]);
      break;
    default:  // small alphabet
      for (auto& b : v) b = (uint8_t)("ab\n"[rnd() % 3]);
  }
  return v;
}

int main() {
  void* h = dlopen("liblz4.so.1", RTLD_NOW);
  if (!h) h = dlopen("/opt/conda/lib/liblz4.so.1", RTLD_NOW);
  auto lib_dec = h ? (int (*)(const char*, char*, int, int))dlsym(h, "LZ4_decompress_safe") : nullptr;
  auto lib_enc = h ? (int (*)(const char*, char*, int, int))dlsym(h, "LZ4_compress_default") : nullptr;
  auto lib_bound = h ? (int (*)(int))dlsym(h, "LZ4_compressBound") : nullptr;
  const size_t sizes[] = {0, 1, 4, 5, 11, 12, 13, 14, 15, 16, 17, 31, 64, 255, 256, 269, 270, 271, 4096, 65535, 65536, 65537, 70000, 300000, 1 << 20};
  long cases = 0, interop = 0, hostile = 0;
  for (int kind = 0; kind < 5; ++kind)
    for (size_t n : sizes) {
      const std::vector<uint8_t> in = make_input(kind, n);
      std::vector<uint8_t> packed((size_t)xsg::lz4_compress_bound((int)n));
      const int pn = xsg::lz4_block_encode(in.data(), (int)n, packed.data(), (int)packed.size());
      if (pn <= 0) return printf("encode failed kind=%d n=%zu\n", kind, n), 1;
      std::vector<uint8_t> out(n + 1, 0xEE);
      if (xsg::lz4_block_decode(packed.data(), (size_t)pn, out.data(), n) != (int64_t)n || !same(out.data(), in.data(), n) ||
          out[n] != 0xEE)
        return printf("round trip failed kind=%d n=%zu\n", kind, n), 1;
      if ((kind == 1 || kind == 3) && n >= 4096 && (size_t)pn > n / 20) return printf("no compression kind=%d n=%zu -> %d\n", kind, n, pn), 1;
      ++cases;
      if (lib_dec && lib_enc && lib_bound) {
        std::vector<uint8_t> o2(n + 1);
        if (lib_dec((const char*)packed.data(), (char*)o2.data(), pn, (int)n) != (int)n || !same(o2.data(), in.data(), n))
          return printf("liblz4 cannot decode our block kind=%d n=%zu\n", kind, n), 1;
        std::vector<uint8_t> p2((size_t)lib_bound((int)n) + 1);
        const int p2n = lib_enc((const char*)in.data(), (char*)p2.data(), (int)n, (int)p2.size());
        if (p2n <= 0 && n > 0) return printf("liblz4 encode failed\n"), 1;
        if (p2n > 0 &&
            (xsg::lz4_block_decode(p2.data(), (size_t)p2n, out.data(), n) != (int64_t)n || !same(out.data(), in.data(), n)))
          return printf("cannot decode liblz4's block kind=%d n=%zu\n", kind, n), 1;
        ++interop;
      }
      // hostile input: truncations and bit flips (heap copies of exact size so that ASan sees any overrun)
      for (int t = 0; t < 40 && pn > 1; ++t) {
        const size_t cut = 1 + rnd() % (size_t)pn;
        std::vector<uint8_t> bad(packed.begin(), packed.begin() + (long)cut);
        if (t & 1) bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() % 8));
        std::vector<uint8_t> o3(n ? n : 1);
        const int64_t r = xsg::lz4_block_decode(bad.data(), bad.size(), o3.data(), n);
        if (r > (int64_t)n) return printf("decoder wrote past the output\n"), 1;
        ++hostile;
      }
    }
  printf("lz4 selftest ok: %ld round trips, %ld interop checks with liblz4, %ld hostile blocks\n", cases, interop, hostile);
  return 0;
}
