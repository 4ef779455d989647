// host_model.cpp -- TEST INFRASTRUCTURE.  Replays, on the CPU, the exact
// decomposition the GPU path uses, built from the product's own host/device
// headers (xsg_tail.h, xsg_linesum.h):
//
//   bulk  : every occurrence starting at o < limit, decided position by position
//   keep  : greedy non-overlap (match modes) / first occurrence per line (line modes)
//   tail  : the reference walk replayed over [limit, L) by xsg::tail_walk
//   lines : per-16-byte-unit summaries reduced with xsg::sum_combine in tile order
//
// tests/test_host_model.py checks it against the oracle on many random inputs,
// so that the algorithm is known to be right before it ever runs on a GPU.
// It is NOT a fallback: nothing under x-search_amd/ links or loads this.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../x-search_amd/csrc/xsg_linesum.h"
#include "../../x-search_amd/csrc/xsg_tail.h"

using namespace xsg;

static uint64_t bulk_limit(uint64_t L, uint32_t plen, int exact) {
  if (exact) return L >= plen ? L - plen + 1 : 0;
  return tail_zone_begin(L, plen);
}

// bit 1 of `exact` = ignore_case (the pattern passed in is already lowered, like the device copy)
static bool g_icase = false;
static std::vector<uint64_t> bulk_occ(const uint8_t* d, uint64_t L, const uint8_t* p, uint32_t plen, int exact) {
  std::vector<uint64_t> v;
  const uint64_t lim = bulk_limit(L, plen, exact);
  for (uint64_t o = 0; o < lim; ++o)
    if (occurs_at(d, o, p, plen, g_icase)) v.push_back(o);
  return v;
}

static uint64_t line_start(const uint8_t* d, uint64_t o) {
  while (o > 0 && d[o - 1] != '\n') --o;
  return o;
}

static uint64_t walk_entry(const uint8_t* d, uint64_t L, uint64_t last_end, bool skip) {
  if (last_end == 0) return 0;
  if (!skip) return last_end;
  const int64_t nl = next_newline(d, last_end, L);
  return nl < 0 ? UINT64_MAX : (uint64_t)nl + 1;
}

// final list of (line start | match offset, first match of the line)
static void model_list(const uint8_t* d, uint64_t L, const uint8_t* p, uint32_t plen, int exact, int line_mode,
                       std::vector<uint64_t>& pos, std::vector<uint64_t>& first_match) {
  std::vector<uint64_t> occ = bulk_occ(d, L, p, plen, exact);
  uint64_t last_end = 0;
  if (line_mode) {
    uint64_t prev_ls = UINT64_MAX;
    for (uint64_t o : occ) {
      const uint64_t ls = line_start(d, o);
      if (ls != prev_ls) {
        pos.push_back(ls);
        first_match.push_back(o);
        last_end = o + plen;
        prev_ls = ls;
      }
    }
  } else {
    uint64_t last = 0;
    bool have = false;
    for (uint64_t o : occ) {
      if (!have || o >= last + plen) {
        pos.push_back(o);
        first_match.push_back(o);
        last = o;
        have = true;
        last_end = o + plen;
      }
    }
  }
  if (!exact && plen > 1) {
    std::vector<uint64_t> t(tail_max_matches(plen) + 1);
    const uint32_t n = tail_walk(d, L, p, plen, walk_entry(d, L, last_end, line_mode != 0), line_mode != 0, t.data(),
                                 (uint32_t)t.size(), g_icase);
    for (uint32_t k = 0; k < n; ++k) {
      pos.push_back(line_mode ? line_start(d, t[k]) : t[k]);
      first_match.push_back(t[k]);
    }
  }
}

extern "C" {

void hm_set_icase(int on) { g_icase = on != 0; }

uint64_t hm_list(const uint8_t* d, uint64_t L, const uint8_t* p, uint32_t plen, int exact, int line_mode, uint64_t* out,
                 uint64_t cap) {
  std::vector<uint64_t> pos, fm;
  model_list(d, L, p, plen, exact, line_mode, pos, fm);
  for (uint64_t i = 0; i < pos.size() && i < cap; ++i) out[i] = pos[i];
  return pos.size();
}

// XSG_LINES: (begin,len) of reported lines; unterminated last line dropped
uint64_t hm_lines(const uint8_t* d, uint64_t L, const uint8_t* p, uint32_t plen, int exact, uint64_t* beg,
                  uint64_t* len, uint64_t cap) {
  std::vector<uint64_t> pos, fm;
  model_list(d, L, p, plen, exact, 1, pos, fm);
  uint64_t n = 0;
  for (size_t i = 0; i < pos.size(); ++i) {
    const int64_t e = next_newline(d, fm[i] + plen, L);
    if (e < 0) continue;
    if (n < cap) {
      beg[n] = pos[i];
      len[n] = (uint64_t)e - pos[i];
    }
    ++n;
  }
  return n;
}

// the async count path: per-tile counts + chunk_last_end + tail walk
uint64_t hm_count_matches_borderfree(const uint8_t* d, uint64_t L, const uint8_t* p, uint32_t plen, int exact) {
  std::vector<uint64_t> occ = bulk_occ(d, L, p, plen, exact);
  uint64_t n = occ.size();
  const uint64_t last_end = occ.empty() ? 0 : occ.back() + plen;
  if (!exact && plen > 1) n += tail_walk(d, L, p, plen, walk_entry(d, L, last_end, false), false, nullptr, 0, g_icase);
  return n;
}

// the async count_lines path: unit summaries reduced in groups of `group`
// units (any grouping must give the same answer), + tail walk
uint64_t hm_count_lines(const uint8_t* d, uint64_t L, const uint8_t* p, uint32_t plen, int exact, uint32_t group) {
  std::vector<uint64_t> occ = bulk_occ(d, L, p, plen, exact);
  const uint64_t nunits = (L + 15) / 16;
  std::vector<uint32_t> us(nunits, 0);
  std::vector<uint32_t> h(nunits, 0), n(nunits, 0);
  for (uint64_t o : occ) h[o / 16] |= 1u << (o % 16);
  for (uint64_t i = 0; i < L; ++i)
    if (d[i] == '\n') n[i / 16] |= 1u << (i % 16);
  for (uint64_t u = 0; u < nunits; ++u) us[u] = sum_of_unit(h[u], n[u]);
  if (group < 1) group = 1;
  // two-level reduction like lanes -> tiles -> chunk
  uint32_t total = 0;
  bool have = false;
  for (uint64_t g = 0; g < nunits; g += group) {
    uint32_t s = us[g];
    for (uint64_t u = g + 1; u < g + group && u < nunits; ++u) s = sum_combine(s, us[u]);
    total = have ? sum_combine(total, s) : s;
    have = true;
  }
  uint64_t lines = have ? sum_total_lines(total) : 0;
  const uint64_t last_end = occ.empty() ? 0 : occ.back() + plen;
  if (!exact && plen > 1) lines += tail_walk(d, L, p, plen, walk_entry(d, L, last_end, true), true, nullptr, 0, g_icase);
  return lines;
}


// xsg::sum_combine_lanes (the ballot form k_scan uses per wave-load) against the
// ordered reduction of the same 64 unit summaries.  Returns the number of mismatches.
uint64_t hm_check_lane_combine(uint64_t seed, uint64_t iters) {
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1, bad = 0;
  auto rnd = [&]() {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return x;
  };
  for (uint64_t it = 0; it < iters; ++it) {
    uint32_t us[64];
    const int dens = (int)(rnd() % 5);
    for (int l = 0; l < 64; ++l) {
      uint32_t h = 0, n = 0;
      for (int b = 0; b < 16; ++b) {
        const int r = (int)(rnd() % (uint64_t)(4 + dens * 6));
        if (r == 0) n |= 1u << b;
        else if (r == 1) h |= 1u << b;
      }
      if (it % 7 == 0 && (rnd() & 1)) n = 0;
      if (it % 11 == 0) h = 0;
      us[l] = sum_of_unit(h, n);
    }
    uint32_t ref = us[0];
    for (int l = 1; l < 64; ++l) ref = sum_combine(ref, us[l]);
    unsigned long long N = 0, Fm = 0, Lm = 0;
    uint32_t csum = 0;
    for (int l = 0; l < 64; ++l) {
      if (us[l] & kSumNl) N |= 1ull << l;
      if (us[l] & kSumF) Fm |= 1ull << l;
      if (us[l] & kSumL) Lm |= 1ull << l;
      csum += us[l] >> kSumCShift;
    }
    bad += sum_combine_lanes(N, Fm, Lm, csum) != ref;
  }
  return bad;
}

// xsg::tail_walk_masks (k_count_finish: one wave per chunk decides the zone's positions in parallel, the walk then
// runs on bit masks) against xsg::tail_walk on random chunks / patterns / entry points.  Returns #mismatches.
uint64_t hm_check_tail_masks(uint64_t seed, uint64_t iters) {
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + 7, bad = 0;
  auto rnd = [&]() {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return x;
  };
  std::vector<uint8_t> d, p;
  for (uint64_t it = 0; it < iters; ++it) {
    const uint32_t plen = 2 + (uint32_t)(rnd() % (kTailMaskMaxPlen - 1));  // 2..33
    const uint64_t L = rnd() % 4 == 0 ? rnd() % (plen + 40) : plen + rnd() % 200;
    const int alpha = 2 + (int)(rnd() % 2);
    const bool icase = rnd() % 4 == 0;
    d.assign(L + 1, 0);
    p.assign(plen, 0);
    for (uint64_t i = 0; i < L; ++i) {
      const uint64_t r = rnd() % 16;
      d[i] = r == 0 ? '\n' : (uint8_t)((icase && (r & 1) ? 'A' : 'a') + (int)(rnd() % (uint64_t)alpha));
    }
    // a pattern taken from the text (so that partial and full matches happen), sometimes periodic
    if (L >= plen && rnd() % 3 != 0) {
      const uint64_t at = rnd() % (L - plen + 1);
      for (uint32_t k = 0; k < plen; ++k) p[k] = d[at + k] == '\n' ? 'a' : fold(d[at + k], true);
    } else {
      for (uint32_t k = 0; k < plen; ++k) p[k] = (uint8_t)('a' + (int)((k % (1 + rnd() % 3)) % (uint64_t)alpha));
    }
    if (!icase)
      for (uint32_t k = 0; k < plen; ++k) p[k] = p[k];
    const uint64_t Z = tail_zone_begin(L, plen);
    const uint32_t n = (uint32_t)(L - Z);
    uint32_t K[64] = {0};
    uint64_t full = 0, nz = 0, nlm = 0;
    for (uint32_t i = 0; i < n; ++i) {
      const uint64_t o = Z + i;
      uint32_t k = 0;
      if (L - o >= plen)
        while (k < plen && fold(d[o + k], icase) == p[k]) ++k;
      K[i] = k;
      if (k == plen) full |= 1ull << i;
      if (k) nz |= 1ull << i;
      if (d[o] == '\n') nlm |= 1ull << i;
    }
    for (int rep = 0; rep < 4; ++rep) {
      const bool skip = rnd() & 1;
      uint64_t shift0 = rnd() % 5 == 0 ? 0 : rnd() % (L + 2);
      if (rep == 3) shift0 = UINT64_MAX;
      const uint32_t want = tail_walk(d.data(), L, p.data(), plen, shift0, skip, nullptr, 0, icase);
      const uint32_t got = tail_walk_masks(L, plen, shift0, skip, full, nz, nlm, [&](uint32_t j) { return K[j]; });
      bad += want != got;
    }
  }
  return bad;
}

// sum_combine_lanes on summaries of whole SPANS (k_count_finish combines 64 per-wave tile summaries at a time):
// closed-segment counts far above a unit's 7.
uint64_t hm_check_span_combine(uint64_t seed, uint64_t iters) {
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + 3, bad = 0;
  auto rnd = [&]() {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return x;
  };
  for (uint64_t it = 0; it < iters; ++it) {
    uint32_t sp[64];
    for (int l = 0; l < 64; ++l) {
      // a span = a few random units combined (a valid summary by construction)
      const int nu = (int)(rnd() % 6);
      uint32_t s = 0;  // identity
      for (int u = 0; u < nu; ++u) {
        uint32_t h = 0, n = 0;
        for (int b = 0; b < 16; ++b) {
          const int r = (int)(rnd() % 6);
          if (r == 0) n |= 1u << b;
          else if (r == 1) h |= 1u << b;
        }
        if (it % 5 == 0) h = 0;
        s = u ? sum_combine(s, sum_of_unit(h, n)) : sum_of_unit(h, n);
      }
      if (nu == 0 && (rnd() & 1)) s = kSumNl;  // "has a newline, no match": the preset of an untouched wave span
      sp[l] = s;
    }
    uint32_t ref = sp[0];
    for (int l = 1; l < 64; ++l) ref = sum_combine(ref, sp[l]);
    unsigned long long N = 0, Fm = 0, Lm = 0;
    uint32_t csum = 0;
    for (int l = 0; l < 64; ++l) {
      if (sp[l] & kSumNl) N |= 1ull << l;
      if (sp[l] & kSumF) Fm |= 1ull << l;
      if (sp[l] & kSumL) Lm |= 1ull << l;
      csum += sp[l] >> kSumCShift;
    }
    bad += sum_combine_lanes(N, Fm, Lm, csum) != ref;
  }
  return bad;
}

}  // extern "C"
