/*
 * xs_oracle.c -- CPU restatement of the x-search literal hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY -- see xs_oracle.h.  Parity status: PINNED
 * (reference unit-test known answers + cross-check against oracle/_ref).
 *
 * This is a restatement, not a copy: the reference's AVX2 routine is described
 * here by what it computes.  For strstr that is a three-part decomposition
 * (SURVEY 8a, row a2):
 *
 *   plen == 1            -> leftmost byte == pat[0]              (exact)
 *   len  <  32 + plen    -> the reference's LOSSY scalar search  (quirk)
 *   otherwise            -> exact leftmost occurrence starting in the first
 *                           32*floor((len-plen)/32) bytes ("body"), and if
 *                           there is none, the lossy scalar search on the rest.
 *
 * The body is vectorised with AVX2 when the host has it (runtime dispatch) so
 * that the timed "port" CPU baseline is a fair one; the scalar body computes
 * the same function.
 */
#define _GNU_SOURCE
#include "xs_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#define XSO_X86 1
#else
#define XSO_X86 0
#endif

static int g_exact = 0;
static xso_findnext_fn g_findnext = NULL;
static xso_findnl_fn g_findnl = NULL;

void xso_set_exact(int exact) { g_exact = exact ? 1 : 0; }
int xso_get_exact(void) { return g_exact; }
void xso_use_primitives(xso_findnext_fn fn, xso_findnl_fn nl) {
  g_findnext = fn;
  g_findnl = nl;
}

/* ------------------------------------------------------------------------ */
/* simd_search.cpp:58-78.  The reference resumes AFTER the mismatching byte
 * (its read index has already been post-incremented), i.e. at shift+k+1 where
 * k is the number of pattern bytes that matched -- not at shift+1.  That
 * skips occurrences that overlap a partial prefix match. */
const char* xso_scalar_strstr(const char* str, size_t len, const char* pat, size_t plen) {
  size_t shift = 0;
  while (shift < len) {
    if (len - shift < plen) return NULL;
    size_t k = 0;
    while (k < plen && str[shift + k] == pat[k]) ++k;
    if (k == plen) return str + shift;
    shift += k + 1;
  }
  return NULL;
}

/* true leftmost occurrence (exact mode only) */
static const char* exact_strstr(const char* str, size_t len, const char* pat, size_t plen) {
  if (plen == 0 || len < plen) return NULL;
  return (const char*)memmem(str, len, pat, plen);
}

/* simd_search.cpp:107-114 */
const char* xso_scalar_strchr(const char* str, size_t len, int c) {
  for (size_t i = 0; i < len; ++i)
    if (str[i] == c) return str + i;
  return NULL;
}

/* ---- bodies ------------------------------------------------------------- */
/* leftmost i in [0, nblocks*32) with str[i]==c, or -1 */
static int64_t chr_body_scalar(const char* str, size_t nblocks, char c) {
  const char* r = (const char*)memchr(str, (unsigned char)c, nblocks * 32);
  return r ? (int64_t)(r - str) : -1;
}

/* leftmost i in [0, nblocks*32) with str[i..i+plen) == pat, or -1.
 * Caller guarantees nblocks*32 + plen <= len (all reads in range). */
static int64_t str_body_scalar(const char* str, size_t nblocks, const char* pat, size_t plen) {
  const size_t n = nblocks * 32;
  const char first = pat[0], last = pat[plen - 1];
  for (size_t i = 0; i < n; ++i) {
    if (str[i] == first && str[i + plen - 1] == last && memcmp(str + i + 1, pat + 1, plen - 1) == 0) return (int64_t)i;
  }
  return -1;
}

#if XSO_X86
__attribute__((target("avx2"))) static int64_t chr_body_avx2(const char* str, size_t nblocks, char c) {
  const __m256i needle = _mm256_set1_epi8(c);
  for (size_t b = 0; b < nblocks; ++b) {
    const __m256i v = _mm256_loadu_si256((const __m256i*)(str + 32 * b));
    const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, needle));
    if (m) return (int64_t)(32 * b + (size_t)__builtin_ctz(m));
  }
  return -1;
}

__attribute__((target("avx2"))) static int64_t str_body_avx2(const char* str, size_t nblocks, const char* pat,
                                                              size_t plen) {
  const __m256i vf = _mm256_set1_epi8(pat[0]);
  const __m256i vl = _mm256_set1_epi8(pat[plen - 1]);
  for (size_t b = 0; b < nblocks; ++b) {
    const char* p = str + 32 * b;
    const __m256i a = _mm256_loadu_si256((const __m256i*)p);
    const __m256i z = _mm256_loadu_si256((const __m256i*)(p + plen - 1));
    uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpeq_epi8(a, vf), _mm256_cmpeq_epi8(z, vl)));
    while (m) {
      const unsigned bit = (unsigned)__builtin_ctz(m);
      if (memcmp(p + bit + 1, pat + 1, plen - 1) == 0) return (int64_t)(32 * b + bit);
      m &= m - 1;
    }
  }
  return -1;
}
static int have_avx2(void) {
  static int cached = -1;
  if (cached < 0) cached = __builtin_cpu_supports("avx2") ? 1 : 0;
  return cached;
}
#endif

static int64_t chr_body(const char* str, size_t nblocks, char c) {
#if XSO_X86
  if (have_avx2()) return chr_body_avx2(str, nblocks, c);
#endif
  return chr_body_scalar(str, nblocks, c);
}
static int64_t str_body(const char* str, size_t nblocks, const char* pat, size_t plen) {
#if XSO_X86
  if (have_avx2()) return str_body_avx2(str, nblocks, pat, plen);
#endif
  return str_body_scalar(str, nblocks, pat, plen);
}

/* simd_search.cpp:116-144: scalar below 32 bytes, else whole 32-byte blocks
 * then scalar on the remainder.  Exact (leftmost byte == c). */
const char* xso_strchr(const char* str, size_t len, char c) {
  if (len < 32) return xso_scalar_strchr(str, len, c);
  const size_t nblocks = len / 32;
  const int64_t hit = chr_body(str, nblocks, c);
  if (hit >= 0) return str + hit;
  return xso_scalar_strchr(str + 32 * nblocks, len - 32 * nblocks, c);
}

/* simd_search.cpp:162-204 */
const char* xso_strstr(const char* str, size_t len, const char* pat, size_t plen) {
  if (plen == 1) return xso_strchr(str, len, pat[0]); /* :164-166 */
  if (g_exact) return exact_strstr(str, len, pat, plen);
  if (len < 32 + plen) return xso_scalar_strstr(str, len, pat, plen); /* :169-171 */
  /* :179-202 loops while the remaining length is >= 32+plen, 32 bytes a step */
  const size_t nblocks = (len - plen) / 32;
  const int64_t hit = str_body(str, nblocks, pat, plen);
  if (hit >= 0) return str + hit;
  return xso_scalar_strstr(str + 32 * nblocks, len - 32 * nblocks, pat, plen); /* :203 */
}

/* simd_search.cpp:289-295 */
int64_t xso_find_next(const char* pat, size_t plen, const char* str, size_t len, size_t shift) {
  if (shift > len) return -1;
  const char* m = xso_strstr(str + shift, len - shift, pat, plen);
  return m == NULL ? -1 : (int64_t)(m - str);
}

/* simd_search.cpp:297-303 */
int64_t xso_find_next_newline(const char* str, size_t len, size_t shift) {
  if (shift > len) return -1;
  const char* m = xso_strchr(str + shift, len - shift, '\n');
  return m == NULL ? -1 : (int64_t)(m - str);
}

static inline int64_t FN(const char* pat, size_t plen, const char* str, size_t len, size_t shift) {
  return g_findnext ? g_findnext(pat, plen, str, len, shift) : xso_find_next(pat, plen, str, len, shift);
}
static inline int64_t NL(const char* str, size_t len, size_t shift) {
  return g_findnl ? g_findnl(str, len, shift) : xso_find_next_newline(str, len, shift);
}

/* simd_search.cpp:305-322 (declared as findAllPerLine, simd_search.h:77) */
uint64_t xso_count_matching_lines(const char* pat, size_t plen, const char* str, size_t len) {
  uint64_t count = 0;
  size_t shift = 0;
  for (;;) {
    int64_t m = FN(pat, plen, str, len, shift);
    if (m < 0) break;
    ++count;
    shift = (size_t)m + plen;
    m = NL(str, len, shift);
    if (m < 0) break;
    shift = (size_t)m + 1;
  }
  return count;
}

/* simd_search.cpp:324-336 (declared as findAll, simd_search.h:88) */
uint64_t xso_count_matches(const char* pat, size_t plen, const char* str, size_t len) {
  uint64_t count = 0;
  size_t shift = 0;
  for (;;) {
    const int64_t m = FN(pat, plen, str, len, shift);
    if (m < 0) break;
    ++count;
    shift = (size_t)m + plen;
  }
  return count;
}

/* ------------------------------------------------------------------------ */
/* search_wrappers.h:111-123.  Distance from the match back to the byte after
 * the previous '\n' (or to offset 0).  Returns the LINE START (v - distance);
 * the reference returns the distance and the caller subtracts (:151-153).
 * A match whose own first byte is '\n' yields v+1 in the reference
 * (relative_offset - 1 wraps); reproduced here. */
static uint64_t line_start_of(const char* data, uint64_t v) {
  uint64_t rel = 0;
  for (;;) {
    if (data[v - rel] == '\n') return v - (rel - 1); /* rel==0 -> v+1, as the reference */
    if (rel >= v) return 0;
    ++rel;
  }
}

/* search_wrappers.h:29-52 (generic walk) and :136-139 */
uint64_t xso_byte_offsets_match(const char* data, size_t len, const char* pat, size_t plen, int skip_to_nl,
                                uint64_t* out, uint64_t cap) {
  uint64_t n = 0;
  size_t shift = 0;
  while (shift < len) {
    int64_t m = FN(pat, plen, data, len, shift);
    if (m < 0) break;
    if (n < cap) out[n] = (uint64_t)m;
    ++n;
    shift = (size_t)m + plen;
    if (skip_to_nl) {
      m = NL(data, len, shift);
      if (m < 0) break;
      shift = (size_t)m + 1;
    }
  }
  return n;
}

/* search_wrappers.h:149-154 */
uint64_t xso_byte_offsets_line(const char* data, size_t len, const char* pat, size_t plen, uint64_t* out,
                               uint64_t cap) {
  uint64_t n = 0;
  size_t shift = 0;
  while (shift < len) {
    int64_t m = FN(pat, plen, data, len, shift);
    if (m < 0) break;
    if (n < cap) out[n] = line_start_of(data, (uint64_t)m);
    ++n;
    shift = (size_t)m + plen;
    m = NL(data, len, shift);
    if (m < 0) break;
    shift = (size_t)m + 1;
  }
  return n;
}

/* search_wrappers.h:163-185 */
uint64_t xso_count(const char* data, size_t len, const char* pat, size_t plen, int skip_to_nl) {
  uint64_t n = 0;
  size_t shift = 0;
  while (shift < len) {
    int64_t m = FN(pat, plen, data, len, shift);
    if (m < 0) break;
    ++n;
    shift = (size_t)m + plen;
    if (skip_to_nl) {
      m = NL(data, len, shift);
      if (m < 0) break;
      shift = (size_t)m + 1;
    }
  }
  return n;
}

/* search_wrappers.h:187-207.  A match on a last line without '\n' ends the
 * walk WITHOUT emitting that line (:200-202). */
uint64_t xso_lines(const char* data, size_t len, const char* pat, size_t plen, uint64_t* begin, uint64_t* length,
                   uint64_t cap) {
  uint64_t n = 0;
  size_t shift = 0;
  while (shift < len) {
    const int64_t m = FN(pat, plen, data, len, shift);
    if (m < 0) break;
    const uint64_t b = line_start_of(data, (uint64_t)m);
    shift = (size_t)m + plen;
    const int64_t e = NL(data, len, shift);
    if (e < 0) break;
    shift = (size_t)e + 1;
    if (n < cap) {
      begin[n] = b;
      length[n] = (uint64_t)e - b;
    }
    ++n;
  }
  return n;
}

uint64_t xso_count_newlines(const char* data, size_t len) {
  uint64_t n = 0;
  const char* p = data;
  const char* end = data + len;
  while (p < end) {
    const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
    if (!q) break;
    ++n;
    p = q + 1;
  }
  return n;
}

/* xs::line_indices (no reference implementation; see header). */
uint64_t xso_line_indices(const char* data, size_t len, const char* pat, size_t plen, uint64_t line_base,
                          uint64_t* out, uint64_t cap) {
  uint64_t n = 0;
  size_t shift = 0;
  uint64_t counted_to = 0; /* newlines in data[0, counted_to) == nl_seen */
  uint64_t nl_seen = 0;
  while (shift < len) {
    int64_t m = FN(pat, plen, data, len, shift);
    if (m < 0) break;
    const uint64_t b = line_start_of(data, (uint64_t)m);
    if (b > counted_to) {
      nl_seen += xso_count_newlines(data + counted_to, b - counted_to);
      counted_to = b;
    }
    if (n < cap) out[n] = line_base + nl_seen;
    ++n;
    shift = (size_t)m + plen;
    m = NL(data, len, shift);
    if (m < 0) break;
    shift = (size_t)m + 1;
  }
  return n;
}

/* string_utils.cpp:11-33 */
void xso_to_lower(char* buf, size_t len) {
  for (size_t i = 0; i < len; ++i)
    if (buf[i] >= 'A' && buf[i] <= 'Z') buf[i] = (char)(buf[i] + ('a' - 'A'));
}

/* ------------------------------------------------------------------------ */
typedef struct {
  const char* base;
  const uint64_t* offsets;
  const uint64_t* lengths;
  uint64_t n;
  const char* pat;
  size_t plen;
  int skip_to_nl;
  uint64_t next; /* shared work counter */
  uint64_t* counts_out;
  uint64_t total;
  pthread_mutex_t mu;
} mt_job;

static void* mt_worker(void* arg) {
  mt_job* job = (mt_job*)arg;
  uint64_t local = 0;
  for (;;) {
    const uint64_t i = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
    if (i >= job->n) break;
    const uint64_t c = xso_count(job->base + job->offsets[i], job->lengths[i], job->pat, job->plen, job->skip_to_nl);
    if (job->counts_out) job->counts_out[i] = c;
    local += c;
  }
  pthread_mutex_lock(&job->mu);
  job->total += local;
  pthread_mutex_unlock(&job->mu);
  return NULL;
}

/* Searcher.h:100-120: workers loop "next chunk -> search -> add" until the
 * reader is exhausted; here the reader is the shared index. */
uint64_t xso_count_chunks_mt(const char* base, const uint64_t* offsets, const uint64_t* lengths, uint64_t n,
                             const char* pat, size_t plen, int skip_to_nl, int nthreads, uint64_t* counts_out) {
  mt_job job;
  memset(&job, 0, sizeof job);
  job.base = base;
  job.offsets = offsets;
  job.lengths = lengths;
  job.n = n;
  job.pat = pat;
  job.plen = plen;
  job.skip_to_nl = skip_to_nl;
  job.counts_out = counts_out;
  pthread_mutex_init(&job.mu, NULL);
  if (nthreads < 1) nthreads = 1;
  if (nthreads == 1) {
    mt_worker(&job);
  } else {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, mt_worker, &job);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th);
  }
  pthread_mutex_destroy(&job.mu);
  return job.total;
}

/* ------------------------------------------------------------------------ */
/* Persistent worker pool: the same worker loop (Searcher.h:100-120), with    */
/* threads that live across passes the way the reference's Searcher threads   */
/* live for a whole search (Searcher.h:141-145) -- a timed pass pays no       */
/* pthread_create/join.  The corpus is allocated and FIRST-TOUCHED by the     */
/* workers (pages land on the NUMA node of the thread that writes them, spread */
/* over all nodes the pool runs on).  Timing is taken inside, around the       */
/* passes only.                                                                */
/* ------------------------------------------------------------------------ */
#include <sched.h>
#include <sys/mman.h>
#include <time.h>

struct xso_pool {
  int nthreads;
  pthread_t* th;
  pthread_mutex_t mu;
  pthread_cond_t cv_go, cv_done;
  uint64_t generation; /* bumped per dispatched job */
  int running;         /* workers still inside the current job */
  int quit;
  /* the current job */
  int kind; /* 0 = count, 1 = replicate */
  mt_job job;
  char* dst_base;
  const uint64_t* src_idx;
  const char* const* src_ptr;
};

static void pool_run_item(xso_pool* p, uint64_t* local) {
  mt_job* job = &p->job;
  for (;;) {
    const uint64_t i = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
    if (i >= job->n) break;
    if (p->kind == 1) {
      memcpy(p->dst_base + job->offsets[i], p->src_ptr[p->src_idx[i]], job->lengths[i]);
    } else {
      const uint64_t c = xso_count(job->base + job->offsets[i], job->lengths[i], job->pat, job->plen, job->skip_to_nl);
      if (job->counts_out) job->counts_out[i] = c;
      *local += c;
    }
  }
}

static void* pool_worker(void* arg) {
  xso_pool* p = (xso_pool*)arg;
  uint64_t seen = 0;
  for (;;) {
    pthread_mutex_lock(&p->mu);
    while (!p->quit && p->generation == seen) pthread_cond_wait(&p->cv_go, &p->mu);
    if (p->quit) {
      pthread_mutex_unlock(&p->mu);
      return NULL;
    }
    seen = p->generation;
    pthread_mutex_unlock(&p->mu);
    uint64_t local = 0;
    pool_run_item(p, &local);
    pthread_mutex_lock(&p->mu);
    p->job.total += local;
    if (--p->running == 0) pthread_cond_signal(&p->cv_done);
    pthread_mutex_unlock(&p->mu);
  }
}

static void pool_dispatch(xso_pool* p) {
  pthread_mutex_lock(&p->mu);
  p->job.next = 0;
  p->job.total = 0;
  p->running = p->nthreads;
  ++p->generation;
  pthread_cond_broadcast(&p->cv_go);
  while (p->running) pthread_cond_wait(&p->cv_done, &p->mu);
  pthread_mutex_unlock(&p->mu);
}

xso_pool* xso_pool_create(int nthreads, int pin) {
  if (nthreads < 1) nthreads = 1;
  xso_pool* p = (xso_pool*)calloc(1, sizeof *p);
  if (!p) return NULL;
  p->nthreads = nthreads;
  p->th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  pthread_mutex_init(&p->mu, NULL);
  pthread_cond_init(&p->cv_go, NULL);
  pthread_cond_init(&p->cv_done, NULL);
  cpu_set_t allowed;
  int ncpu = 0, cpus[4096];
  if (pin && sched_getaffinity(0, sizeof allowed, &allowed) == 0)
    for (int c = 0; c < CPU_SETSIZE && ncpu < 4096; ++c)
      if (CPU_ISSET(c, &allowed)) cpus[ncpu++] = c;
  for (int t = 0; t < nthreads; ++t) {
    pthread_create(&p->th[t], NULL, pool_worker, p);
    if (pin && ncpu) { /* spread over the allowed CPUs: thread t -> every (ncpu/nthreads)-th CPU */
      cpu_set_t one;
      CPU_ZERO(&one);
      CPU_SET(cpus[(int)(((long long)t * ncpu) / nthreads) % ncpu], &one);
      pthread_setaffinity_np(p->th[t], sizeof one, &one);
    }
  }
  return p;
}

void xso_pool_destroy(xso_pool* p) {
  if (!p) return;
  pthread_mutex_lock(&p->mu);
  p->quit = 1;
  pthread_cond_broadcast(&p->cv_go);
  pthread_mutex_unlock(&p->mu);
  for (int t = 0; t < p->nthreads; ++t) pthread_join(p->th[t], NULL);
  pthread_mutex_destroy(&p->mu);
  pthread_cond_destroy(&p->cv_go);
  pthread_cond_destroy(&p->cv_done);
  free(p->th);
  free(p);
}

/* untouched anonymous memory (no page is resident until a worker writes it) */
char* xso_corpus_alloc(uint64_t bytes) {
  void* m = mmap(NULL, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (m == MAP_FAILED) return NULL;
  madvise(m, bytes, MADV_HUGEPAGE);
  return (char*)m;
}
void xso_corpus_free(char* p, uint64_t bytes) {
  if (p) munmap(p, bytes ? bytes : 1);
}

/* chunk i of the corpus := template src_idx[i] (lengths[i] bytes), written by the pool's workers */
void xso_pool_replicate(xso_pool* p, char* dst_base, const uint64_t* offsets, const uint64_t* lengths, uint64_t n,
                        const uint64_t* src_idx, const char* const* src_ptr) {
  memset(&p->job, 0, sizeof p->job);
  p->kind = 1;
  p->job.offsets = offsets;
  p->job.lengths = lengths;
  p->job.n = n;
  p->dst_base = dst_base;
  p->src_idx = src_idx;
  p->src_ptr = src_ptr;
  pool_dispatch(p);
}

/* `passes` passes of count() over all chunks; returns the total of ONE pass (every pass must agree, else
 * UINT64_MAX) and the wall seconds of all passes in *seconds. */
uint64_t xso_pool_count_chunks(xso_pool* p, const char* base, const uint64_t* offsets, const uint64_t* lengths,
                               uint64_t n, const char* pat, size_t plen, int skip_to_nl, int passes,
                               uint64_t* counts_out, double* seconds) {
  memset(&p->job, 0, sizeof p->job);
  p->kind = 0;
  p->job.base = base;
  p->job.offsets = offsets;
  p->job.lengths = lengths;
  p->job.n = n;
  p->job.pat = pat;
  p->job.plen = plen;
  p->job.skip_to_nl = skip_to_nl;
  p->job.counts_out = counts_out;
  uint64_t want = 0;
  int bad = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int k = 0; k < passes; ++k) {
    pool_dispatch(p);
    if (k == 0) want = p->job.total;
    else if (p->job.total != want) bad = 1;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return bad ? UINT64_MAX : want;
}

/* ------------------------------------------------------------------------ */
/* Regex wrappers for (alternations of) fixed-length class sequences         */
/* (include/xsearch/string_search/search_wrappers.h:63-103, 209-271).        */
/*                                                                           */
/* The reference calls re2::RE2::PartialMatch(input, pattern, &match).  RE2  */
/* is a git submodule of the reference without a recorded commit and is not  */
/* in the snapshot: version UNPINNED.  For the only family    */
/* served -- a fixed number of positions, each accepting a set of bytes,     */
/* the whole expression inside capture group 1 -- RE2's published semantics  */
/* (leftmost match; all alternatives have the same length, so there is       */
/* nothing for "first" or "longest" to decide) reduce to: the smallest       */
/* offset at which every set accepts its byte.  cs_partial_match is that;    */
/* the walks above it restate the reference's loops line by line.  Pinned by */
/* search_wrappersTest.cpp:74-105 ("(a[n|m]t)" -> {2,151,197,507},           */
/* {0,113,176,460}, 4) and cross-checked against CPython's `re`              */
/* (tests/test_oracle_regex.py).                                             */
/* ------------------------------------------------------------------------ */
static int cs_accepts(const xso_classseq* cs, uint32_t a, uint32_t k, unsigned char b) {
  return (cs->sets[a * cs->plen + k][b >> 5] >> (b & 31u)) & 1u;
}

/* stand-in for RE2::PartialMatch(input, "(<expression>)", &match): offset of the match in input or -1.  Every
 * alternative has cs->plen positions, so whichever alternative RE2's leftmost-first rule prefers at the leftmost
 * offset, the match is the same span: match.size() is always cs->plen. */
static int64_t cs_partial_match(const xso_classseq* cs, const char* input, size_t len) {
  if (cs->plen == 0 || len < cs->plen) return -1;
  const size_t last = len - cs->plen;
  const uint32_t nalt = cs->nalt ? cs->nalt : 1u;
  for (size_t o = 0; o <= last; ++o) {
    for (uint32_t a = 0; a < nalt; ++a) {
      uint32_t k = 0;
      while (k < cs->plen && cs_accepts(cs, a, k, (unsigned char)input[o + k])) ++k;
      if (k == cs->plen) return (int64_t)o;
    }
  }
  return -1;
}

/* _regex_byte_offsets (:63-87); as_line_start = the func of regex::byte_offsets_line (:220-225) */
static uint64_t regex_byte_offsets(const char* data, size_t len, const xso_classseq* cs, int skip_to_nl,
                                   int as_line_start, uint64_t* out, uint64_t cap) {
  uint64_t n = 0;
  const char* input = data; /* re2::StringPiece input(data.data(), data.size()) */
  size_t input_len = len;
  size_t total_shift = 0;
  for (;;) {
    const int64_t at = cs_partial_match(cs, input, input_len); /* :72 */
    if (at < 0) break;
    size_t shift = (size_t)at; /* :73 match.data() - input.data() */
    const uint64_t v = shift + total_shift;
    if (n < cap) out[n] = as_line_start ? line_start_of(data, v) : v; /* :74 */
    ++n;
    shift += cs->plen; /* :75 match.size() */
    input += shift;    /* :76 remove_prefix */
    input_len -= shift;
    if (skip_to_nl) { /* :77-85 */
      const char* nl = (const char*)memchr(input, '\n', input_len);
      if (!nl) break;
      const size_t next_nl = (size_t)(nl - input) + 1;
      shift += next_nl;
      input += next_nl;
      input_len -= next_nl;
    }
    total_shift += shift; /* :86 */
  }
  return n;
}

/* regex::byte_offsets_match (:242-245) */
uint64_t xso_regex_byte_offsets_match(const char* data, size_t len, const xso_classseq* cs, int skip_to_nl,
                                      uint64_t* out, uint64_t cap) {
  return regex_byte_offsets(data, len, cs, skip_to_nl, 0, out, cap);
}

/* regex::byte_offsets_line (:220-225) */
uint64_t xso_regex_byte_offsets_line(const char* data, size_t len, const xso_classseq* cs, uint64_t* out,
                                     uint64_t cap) {
  return regex_byte_offsets(data, len, cs, 1, 1, out, cap);
}

/* regex::count (:250-271) */
uint64_t xso_regex_count(const char* data, size_t len, const xso_classseq* cs, int skip_to_nl) {
  uint64_t counter = 0;
  const char* input = data;
  size_t input_len = len;
  for (;;) {
    const int64_t at = cs_partial_match(cs, input, input_len); /* :255 */
    if (at < 0) break;
    size_t shift = (size_t)at;
    counter++;
    shift += cs->plen;
    input += shift;
    input_len -= shift;
    if (skip_to_nl) { /* :260-267 */
      const char* nl = (const char*)memchr(input, '\n', input_len);
      if (!nl) break;
      const size_t next_nl = (size_t)(nl - input) + 1;
      input += next_nl;
      input_len -= next_nl;
    }
  }
  return counter;
}

/* The snapshot has no regex `line` / line_indices wrapper; xs::lines and xs::line_indices with a regex
 * (test/src/xsearchTest.cpp:102-125,173-335) are modelled as the literal walks (:187-207, header note on
 * line_indices) with the regex find in place of findNext. */
uint64_t xso_regex_lines(const char* data, size_t len, const xso_classseq* cs, uint64_t* begin, uint64_t* length,
                         uint64_t cap) {
  uint64_t n = 0;
  size_t shift = 0;
  while (shift < len) {
    const int64_t at = cs_partial_match(cs, data + shift, len - shift);
    if (at < 0) break;
    const uint64_t m = shift + (uint64_t)at;
    const uint64_t b = line_start_of(data, m);
    shift = (size_t)m + cs->plen;
    const char* nl = (const char*)memchr(data + shift, '\n', len - shift);
    if (!nl) break;
    const uint64_t e = (uint64_t)(nl - data);
    shift = (size_t)e + 1;
    if (n < cap) {
      begin[n] = b;
      length[n] = e - b;
    }
    ++n;
  }
  return n;
}

uint64_t xso_regex_line_indices(const char* data, size_t len, const xso_classseq* cs, uint64_t line_base,
                                uint64_t* out, uint64_t cap) {
  uint64_t n = 0, counted_to = 0, nl_seen = 0;
  size_t shift = 0;
  while (shift < len) {
    const int64_t at = cs_partial_match(cs, data + shift, len - shift);
    if (at < 0) break;
    const uint64_t m = shift + (uint64_t)at;
    const uint64_t b = line_start_of(data, m);
    for (; counted_to < b; ++counted_to) nl_seen += data[counted_to] == '\n';
    if (n < cap) out[n] = line_base + nl_seen;
    ++n;
    shift = (size_t)m + cs->plen;
    const char* nl = (const char*)memchr(data + shift, '\n', len - shift);
    if (!nl) break;
    shift = (size_t)(nl - data) + 1;
  }
  return n;
}
