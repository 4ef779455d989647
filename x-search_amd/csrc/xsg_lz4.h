// xsg_lz4.h -- a self-contained LZ4 *block* codec (the public block format:
// token = literal length << 4 | match length - 4, length extension bytes of 255,
// literals, 2-byte little-endian offset, last sequence = literals only, the last 5
// bytes are literals and no match starts within the last 12 bytes).
//
// The reference links the system liblz4 (Dockerfile:8) and so does the reader here
// (dlopen); this codec is what the pipeline falls back to when no liblz4 is
// installed on the host, so an .xslz4 corpus still opens.  Host-side IO plumbing:
// nothing here searches.  Decoder: bounds-checked, never reads or writes outside the
// given buffers, returns -1 on malformed input.  Encoder: greedy, one 4-byte hash
// table, always emits a valid block (ratio a little below liblz4's).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace xsg {

inline int lz4_compress_bound(int n) { return n < 0 ? 0 : n + n / 255 + 16; }

// returns the number of bytes written to dst (== the decoded size), or -1
inline int64_t lz4_block_decode(const uint8_t* src, size_t src_n, uint8_t* dst, size_t dst_cap) {
  size_t ip = 0, op = 0;
  if (src_n == 0) return -1;
  for (;;) {
    if (ip >= src_n) return -1;
    const uint32_t token = src[ip++];
    size_t lit = token >> 4;
    if (lit == 15) {
      uint32_t b;
      do {
        if (ip >= src_n) return -1;
        b = src[ip++];
        lit += b;
      } while (b == 255);
    }
    if (lit > src_n - ip || lit > dst_cap - op) return -1;
    if (lit) memcpy(dst + op, src + ip, lit);
    ip += lit;
    op += lit;
    if (ip == src_n) return (int64_t)op;  // the last sequence has no match part
    if (src_n - ip < 2) return -1;
    const size_t off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
    ip += 2;
    if (off == 0 || off > op) return -1;
    size_t mlen = (token & 15u);
    if (mlen == 15) {
      uint32_t b;
      do {
        if (ip >= src_n) return -1;
        b = src[ip++];
        mlen += b;
      } while (b == 255);
    }
    mlen += 4;
    if (mlen > dst_cap - op) return -1;
    const uint8_t* m = dst + op - off;
    if (off >= mlen) {
      memcpy(dst + op, m, mlen);
    } else {
      for (size_t k = 0; k < mlen; ++k) dst[op + k] = m[k];  // overlapping copy replicates the period
    }
    op += mlen;
  }
}

// returns the compressed size, or 0 if dst_cap is too small (lz4_compress_bound(n) always suffices)
inline int lz4_block_encode(const uint8_t* src, int n, uint8_t* dst, int dst_cap) {
  if (n < 0 || dst_cap < lz4_compress_bound(n)) return 0;
  constexpr int kHashBits = 16;
  std::vector<int32_t> table((size_t)1 << kHashBits, -1);
  auto rd32 = [&](int i) {
    uint32_t v;
    memcpy(&v, src + i, 4);
    return v;
  };
  auto hash = [&](uint32_t v) { return (v * 2654435761u) >> (32 - kHashBits); };
  int op = 0, anchor = 0, ip = 0;
  const int mflimit = n - 12;  // no match may start beyond this
  const int matchlimit = n - 5;  // matches end before the last 5 bytes
  auto emit_len = [&](int len) {
    while (len >= 255) dst[op++] = 255, len -= 255;
    dst[op++] = (uint8_t)len;
  };
  while (ip < mflimit) {
    const uint32_t v = rd32(ip);
    const uint32_t h = hash(v);
    const int cand = table[h];
    table[h] = ip;
    if (cand < 0 || ip - cand > 65535 || rd32(cand) != v) {
      ++ip;
      continue;
    }
    int mlen = 4;
    while (ip + mlen < matchlimit && src[cand + mlen] == src[ip + mlen]) ++mlen;
    const int lit = ip - anchor;
    uint8_t* token = dst + op++;
    *token = (uint8_t)((lit >= 15 ? 15 : lit) << 4);
    if (lit >= 15) emit_len(lit - 15);
    if (lit) memcpy(dst + op, src + anchor, (size_t)lit);
    op += lit;
    dst[op++] = (uint8_t)((ip - cand) & 0xff);
    dst[op++] = (uint8_t)((ip - cand) >> 8);
    *token |= (uint8_t)(mlen - 4 >= 15 ? 15 : mlen - 4);
    if (mlen - 4 >= 15) emit_len(mlen - 4 - 15);
    ip += mlen;
    anchor = ip;
  }
  const int lit = n - anchor;  // last literals
  uint8_t* token = dst + op++;
  *token = (uint8_t)((lit >= 15 ? 15 : lit) << 4);
  if (lit >= 15) emit_len(lit - 15);
  if (lit) memcpy(dst + op, src + anchor, (size_t)lit);
  op += lit;
  return op;
}

}  // namespace xsg
