/*
 * xs_oracle.h -- CPU restatement of the x-search literal hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under x-search_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and there only as the checker / the
 * reported CPU baseline.
 *
 * Parity status: PINNED.  Every known-answer value of the reference's own
 * unit tests (test/src/string_search/simd_searchTest.cpp:36-99 and
 * test/src/string_search/search_wrappersTest.cpp:26,39,52,64-69) reproduces
 * (tests/test_oracle_golden.py), and the primitives are cross-checked against
 * oracle/_ref/libxsref.so, which is the reference's own
 * src/string_search/simd_search.cpp compiled unmodified (oracle/Makefile).
 * ignore_case (toLower on both sides) equals that build's simd::strcasestr
 * (tests/test_oracle_golden.py).  The regex wrappers are restated for
 * fixed-length class sequences only: RE2 is an unpinned, absent submodule; that
 * part is pinned by the known answers of search_wrappersTest.cpp:74-105 and
 * cross-checked against CPython's `re` (tests/test_oracle_regex.py).
 *
 * Each function cites the reference file:line whose behaviour it restates.
 * All paths are relative to the reference tree.
 */
#ifndef XS_ORACLE_H
#define XS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- semantics switch -------------------------------------------------- */
/* 0 (default): reference semantics, including the lossy scalar tail of
 *              simd::strstr (src/string_search/simd_search.cpp:58-78,203).
 * 1          : "exact" semantics: every call returns the true leftmost
 *              occurrence (what the reference intends; used to test the
 *              quirk-off mode of the product). */
void xso_set_exact(int exact);
int xso_get_exact(void);

/* Optional: route the wrappers through foreign primitives (used to drive the
 * restated wrappers on top of oracle/_ref's findNext / findNextNewLine).
 * Pass NULL to go back to the restated primitives. */
typedef int64_t (*xso_findnext_fn)(const char* pat, size_t plen, const char* str, size_t len, size_t shift);
typedef int64_t (*xso_findnl_fn)(const char* str, size_t len, size_t shift);
void xso_use_primitives(xso_findnext_fn fn, xso_findnl_fn nl);

/* ---- L0 primitives (src/string_search/simd_search.cpp) ------------------ */
const char* xso_scalar_strstr(const char* str, size_t len, const char* pat, size_t plen);   /* :58-78  */
const char* xso_scalar_strchr(const char* str, size_t len, int c);                          /* :107-114 */
const char* xso_strchr(const char* str, size_t len, char c);                                /* :116-144 */
const char* xso_strstr(const char* str, size_t len, const char* pat, size_t plen);          /* :162-204 */
int64_t xso_find_next(const char* pat, size_t plen, const char* str, size_t len, size_t shift);  /* :289-295 */
int64_t xso_find_next_newline(const char* str, size_t len, size_t shift);                   /* :297-303 */
uint64_t xso_count_matching_lines(const char* pat, size_t plen, const char* str, size_t len);    /* :305-322 */
uint64_t xso_count_matches(const char* pat, size_t plen, const char* str, size_t len);      /* :324-336 */

/* ---- L1 wrappers (include/xsearch/string_search/search_wrappers.h) ------ */
/* All "list" functions write at most `cap` entries and return the number of
 * entries the reference would have produced (call with cap == 0 to size). */

/* byte_offsets_match (:136-139) == _byte_offsets(data, pattern, skip_to_nl) (:29-52) */
uint64_t xso_byte_offsets_match(const char* data, size_t len, const char* pat, size_t plen, int skip_to_nl,
                                uint64_t* out, uint64_t cap);
/* byte_offsets_line (:149-154) with previous_new_line_offset_relative_to_match (:111-123) */
uint64_t xso_byte_offsets_line(const char* data, size_t len, const char* pat, size_t plen, uint64_t* out,
                               uint64_t cap);
/* count (:163-185) */
uint64_t xso_count(const char* data, size_t len, const char* pat, size_t plen, int skip_to_nl);
/* line (:187-207): line i is data[begin[i], begin[i]+length[i]) -- excludes '\n' (:204) */
uint64_t xso_lines(const char* data, size_t len, const char* pat, size_t plen, uint64_t* begin, uint64_t* length,
                   uint64_t cap);
/* xs::line_indices -- no implementation in the reference snapshot; semantics
 * pinned by test/src/xsearchTest.cpp:95-125 + the .meta fixtures:
 * 0-based count of '\n' before the start of each matching line, plus
 * `line_base` (the number of '\n' in all preceding chunks). */
uint64_t xso_line_indices(const char* data, size_t len, const char* pat, size_t plen, uint64_t line_base,
                          uint64_t* out, uint64_t cap);
uint64_t xso_count_newlines(const char* data, size_t len);

/* ---- regex wrappers, class-sequence family only (search_wrappers.h:63-103,209-271) ---- */
/* RE2 (unpinned submodule, absent) is restated for alternations of fixed-length sequences of byte sets
 * (all alternatives of one length) only: see the block comment in xs_oracle.c.  The expression -> sets parser of the oracle lives in xs_oracle.py
 * (compile_class_sequence); bit b of sets[k] set <=> position k accepts byte b. */
typedef struct xso_classseq {
  uint32_t plen;          /* 1..32 positions */
  uint32_t nalt;          /* alternatives of that common length (0 is read as 1) */
  uint32_t sets[256][8];  /* alternative a, position k: sets[a * plen + k] */
} xso_classseq;
uint64_t xso_regex_byte_offsets_match(const char* data, size_t len, const xso_classseq* cs, int skip_to_nl,
                                      uint64_t* out, uint64_t cap);                              /* :242-245 */
uint64_t xso_regex_byte_offsets_line(const char* data, size_t len, const xso_classseq* cs, uint64_t* out,
                                     uint64_t cap);                                               /* :220-225 */
uint64_t xso_regex_count(const char* data, size_t len, const xso_classseq* cs, int skip_to_nl); /* :250-271 */
/* no regex line()/line_indices in the snapshot: the literal walks with the regex find (inferred) */
uint64_t xso_regex_lines(const char* data, size_t len, const xso_classseq* cs, uint64_t* begin, uint64_t* length,
                         uint64_t cap);
uint64_t xso_regex_line_indices(const char* data, size_t len, const xso_classseq* cs, uint64_t line_base,
                                uint64_t* out, uint64_t cap);

/* ---- ignore_case building block (src/utils/string_utils.cpp:11-33) ------- */
/* simd::toLower: bytes 'A'..'Z' += 32 in place (signed compares in the AVX2 body,
 * std::tolower in the C locale for the remainder: ASCII only, bytes >= 0x80 untouched).
 * The snapshot has no case-insensitive wrapper; ignore_case is modelled as
 * search(toLower(chunk), toLower(pattern)) -- offsets unchanged. */
void xso_to_lower(char* buf, size_t len);

/* ---- chunk driver (include/xsearch/Searcher.h:100-120 worker loop) ------- */
/* N worker threads pull chunk indices from a shared counter (work stealing,
 * like run_thread), each runs count() on its chunk; returns the sum.
 * counts_out (optional, n entries) receives the per-chunk results. */
uint64_t xso_count_chunks_mt(const char* base, const uint64_t* offsets, const uint64_t* lengths, uint64_t n,
                             const char* pat, size_t plen, int skip_to_nl, int nthreads, uint64_t* counts_out);

/* The same worker loop on a persistent pool (threads live across passes, like the reference's Searcher
 * threads live for a whole search, Searcher.h:141-145); pin != 0 spreads the threads over the allowed CPUs.
 * bench.py's cpu_baseline leg: corpus allocated untouched (xso_corpus_alloc), first-touched by the workers
 * (xso_pool_replicate), timed inside xso_pool_count_chunks. */
typedef struct xso_pool xso_pool;
xso_pool* xso_pool_create(int nthreads, int pin);
void xso_pool_destroy(xso_pool* p);
char* xso_corpus_alloc(uint64_t bytes);
void xso_corpus_free(char* p, uint64_t bytes);
void xso_pool_replicate(xso_pool* p, char* dst_base, const uint64_t* offsets, const uint64_t* lengths, uint64_t n,
                        const uint64_t* src_idx, const char* const* src_ptr);
uint64_t xso_pool_count_chunks(xso_pool* p, const char* base, const uint64_t* offsets, const uint64_t* lengths,
                               uint64_t n, const char* pat, size_t plen, int skip_to_nl, int passes,
                               uint64_t* counts_out, double* seconds);

#ifdef __cplusplus
}
#endif
#endif /* XS_ORACLE_H */
