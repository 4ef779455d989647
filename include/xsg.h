/*
 * xsg.h -- C ABI of the MI355X literal-scan engine (libxsg.so).
 *
 * This is the drop-in boundary for the hot path of lfreist/x-search: the
 * per-chunk literal search that the reference implements in
 *   src/string_search/simd_search.cpp                       (AVX2 primitives)
 *   include/xsearch/string_search/search_wrappers.h:29-207  (per-chunk walks)
 * and calls from the task functors include/xsearch/tasks/searchers.h:38-93
 * inside the worker loop include/xsearch/Searcher.h:100-120.  With
 * XSG_FLAG_REGEX the same entry points stand in for the regex walks
 * (search_wrappers.h:63-103, 209-271) on fixed-length class-sequence expressions.
 *
 * Everything here is plain C: opaque handles, pointers and sizes.  No torch
 * or C++ types cross this boundary.  The C++ surface the reference's users see
 * (xs::extern_search<Tag>(...)->join()/getResult()) is the header-only layer
 * include/xsearch/xsearch.h on top of this ABI; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Results are bit-identical to the reference's CPU path on the same chunk
 * bytes, including its end-of-chunk behaviour (see XSG_FLAG_EXACT_TAIL).
 *
 * Threading: a ctx may be used from several host threads as long as each
 * thread works on its own shard; calls on ONE shard must not overlap.
 * All functions return XSG_OK (0) or a negative XSG_E* code;
 * xsg_last_error() gives the message of the calling thread's last failure.
 * There is NO CPU fallback: without a usable HIP device every compute entry
 * point fails with XSG_ENODEV.
 */
#ifndef XSG_H
#define XSG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: xsg_shard_invalidate and xsg_result_lines_view joined (round 3); 4: xsg_count_async_status joined, XSG_MAX_PATTERN
 * grew and the line tags accept a literal that contains '\n' (round 4); nothing was removed or changed in meaning */
#define XSG_ABI_VERSION 4

/* ---- status codes ------------------------------------------------------- */
#define XSG_OK 0
#define XSG_EINVAL (-1)   /* bad argument (empty pattern, misaligned chunk, ...) */
#define XSG_ENODEV (-2)   /* no HIP device / device index out of range */
#define XSG_EHIP (-3)     /* a HIP runtime call failed (message has the HIP error) */
#define XSG_ENOMEM (-4)   /* host or device allocation failed */
#define XSG_ENOTSUP (-5)  /* valid request this entry point cannot serve (see docs) */
#define XSG_EIO (-6)      /* file open/read/parse failure */
#define XSG_ESTATE (-7)   /* call sequence error (e.g. result fetched before search) */

/* ---- what to compute: one value per xs:: tag ----------------------------- */
/* README.md:71-81 / test/src/xsearchTest.cpp:344 name the tags; the function
 * of search_wrappers.h each one maps to is given on the right. */
enum xsg_mode {
  XSG_COUNT_MATCHES = 0,      /* xs::count, xs::count_matches -> search::count(data, p, false)   (:163-185) */
  XSG_COUNT_LINES = 1,        /* xs::count_lines              -> search::count(data, p, true)    (:163-185) */
  XSG_MATCH_BYTE_OFFSETS = 2, /* xs::match_byte_offsets       -> search::byte_offsets_match      (:136-139) */
  XSG_LINE_BYTE_OFFSETS = 3,  /* xs::line_byte_offsets        -> search::byte_offsets_line       (:149-154) */
  XSG_LINE_INDICES = 4,       /* xs::line_indices             -> (no reference impl)     */
  XSG_LINES = 5               /* xs::lines                    -> search::line                    (:187-207) */
};

/* ---- pattern flags ------------------------------------------------------- */
/* Default (0): reproduce the reference exactly, including the lossy scalar
 * search it applies to the last <plen+32 bytes of every chunk
 * (simd_search.cpp:58-78,203).  With XSG_FLAG_EXACT_TAIL every occurrence in
 * the chunk is reported (what the reference intends). */
#define XSG_FLAG_EXACT_TAIL 0x1u
/* ASCII case-insensitive search (the `ignore_case` argument of xs::extern_search,
 * example/grep.cpp:69, test/src/xsearchTest.cpp:350).  The snapshot holds no
 * implementation of the case-insensitive wrappers; the semantics here are the
 * reference's building block applied to both sides: simd::toLower
 * (src/utils/string_utils.cpp:11-33: 'A'..'Z' += 32, other bytes unchanged) on
 * the chunk and on the pattern, then the normal search.  That is also exactly
 * what the reference's own (uncalled) case-insensitive primitive computes:
 * simd::strcasestr (src/string_search/simd_search.cpp:220-287) returns the same
 * offset as strstr on the lowered chunk and pattern, end-of-chunk behaviour
 * included (checked on the compiled reference, tests/test_oracle_golden.py).
 * Offsets are unchanged and xs::lines returns the ORIGINAL bytes (goldens keep
 * their case, test/src/xsearchTest.cpp:227-240). */
#define XSG_FLAG_IGNORE_CASE 0x2u
/* The pattern is a regular expression.  The reference hands a pattern that "does
 * not match itself as a regex" (include/xsearch/utils/utils.h:17-25) to RE2 and
 * walks the chunk with RE2::PartialMatch
 * (include/xsearch/string_search/search_wrappers.h:63-87,209-271).  Served here,
 * decided position by position in the scan kernel: alternations of fixed-length
 * sequences of byte classes -- literals, [ASCII classes] incl. [:alpha:] & co,
 * \d \w \s, escapes, x{n}, groups ( ) (?: ), `a|b` at any depth as long as all
 * alternatives of the whole expression have one length (then leftmost-first has
 * nothing to choose) -- which covers every regex the reference's tests use
 * (`She[r ]lock`, test/src/xsearchTest.cpp:9; `(a[n|m]t)`,
 * test/src/string_search/search_wrappersTest.cpp:78).  Up to 32 positions, 8
 * alternatives, 64 sets in all.
 * '.', negated classes and \D \W \S are accepted with their ASCII meaning ('.' =
 * any ASCII byte but '\n'); RE2 matches whole code points there, so a search with
 * such an expression REFUSES data that holds a byte >= 0x80 (XSG_ENOTSUP from the
 * search call; xsg_count_async hands out UINT64_MAX in all four counters) rather
 * than decide it differently.
 * Expressions of VARIABLE length -- x* x+ x? x{n,} x{n,m}, the lazy forms, alternatives
 * of different lengths, more than 32 positions or 8 alternatives -- take a second
 * route: a forward leftmost-first DFA and a reverse longest-match DFA (what RE2
 * itself runs for such a search), compiled on the host, and a kernel that walks every
 * line of the shard with them, one line per lane (csrc/xsg_regex.h,
 * csrc/xsg_rx_kernels.hip).  All six tags; same refusal of non-ASCII data for '.'
 * and negated classes.  An expression of this kind whose sets accept '\n' (\s+,
 * [^,]*) can match across lines: it is walked chunk by chunk, one lane per chunk
 * (slow), and serves XSG_COUNT_MATCHES / XSG_MATCH_BYTE_OFFSETS only.  Refused on
 * this route (XSG_ENOTSUP from xsg_set_pattern, never approximated): expressions
 * that can match the empty string, automata over 16384 table entries.  Refused on both routes: anchors ^ $ \b \A \z, (?flags), \p, \C.
 * CAPTURE GROUPS: the reference's walk is RE2::PartialMatch(input, pattern, &match) (search_wrappers.h:71-75,
 * 254-257), where `match` receives capture group ONE; its tests hand over an expression that IS one group,
 * re2::RE2("(a[n|m]t)") (test/src/string_search/search_wrappersTest.cpp:78).  XSG_FLAG_REGEX(expr) is that call with
 * "(" + expr + ")": offset, length and resume point are those of the WHOLE match; groups inside `expr` only group
 * (`Sher(lock|wood)` reports where `Sher` starts, not where `lock` does; an expression the caller already wrapped,
 * `(a[n|m]t)`, is the same search).  A maintainer who binds this flag passes the expression without relying on an inner
 * group being reported (tests/test_oracle_regex.py::test_the_whole_match_is_group_one_of_the_wrapped_expression).
 * No tail quirk applies (RE2 is exact); XSG_FLAG_IGNORE_CASE folds ASCII letters in
 * the data and in every class, as for literals (automaton route: the classes are
 * closed under case instead, the same thing). */
#define XSG_FLAG_REGEX 0x4u

/* A literal may be up to 32 KiB long (the reference's walk takes any std::string; what bounds it here is one 16-bit
 * field: the end of a tile's last match, relative to the tile, must stay below 2^16).  The scan kernel keeps the first
 * KiB of a long pattern in LDS and verifies the rest of a candidate from the device copy of the pattern.  A regular
 * expression (XSG_FLAG_REGEX) is at most XSG_MAX_REGEX bytes of text. */
#define XSG_MAX_PATTERN 32768u
#define XSG_MAX_REGEX 1024u

/* ---- counters written by the count entry points --------------------------- */
#define XSG_CTR_MATCHES 0  /* XSG_COUNT_MATCHES result */
#define XSG_CTR_LINES 1    /* XSG_COUNT_LINES result   */
#define XSG_CTR_NEWLINES 2 /* number of '\n' in the shard (only if requested) */
#define XSG_CTR_BYTES 3    /* bytes scanned */
#define XSG_NUM_COUNTERS 4

/* OR into `mode` of the count entry points to also count '\n' (needed to
 * derive line-index bases across GPUs). */
#define XSG_WITH_NEWLINES 0x100u

/* line_base value meaning "derive from the preceding chunks of this shard" */
#define XSG_LINE_BASE_AUTO UINT64_MAX

typedef struct xsg_ctx xsg_ctx;     /* one per (process, device): streams, pattern, scratch */
typedef struct xsg_shard xsg_shard; /* a set of chunks resident in device memory */

/* A chunk = what the reference's reader hands to a searcher functor
 * (tasks/readers.h:39-48 -> tasks/searchers.h:49): a run of bytes that is
 * searched on its own.  No match or line spans two chunks. */
typedef struct xsg_chunk {
  uint64_t offset;        /* byte offset of the chunk inside the shard buffer; multiple of 16 */
  uint64_t length;        /* chunk length in bytes (< 2^40) */
  uint64_t global_offset; /* offset of the chunk in the (uncompressed) file: added to every reported byte offset */
  uint64_t line_base;     /* number of '\n' in the file before this chunk, or XSG_LINE_BASE_AUTO */
} xsg_chunk;

/* ---- library ------------------------------------------------------------- */
int xsg_abi_version(void);
const char* xsg_strerror(int code);
const char* xsg_last_error(void);
int xsg_device_count(int* count);

/* ---- context -------------------------------------------------------------- */
int xsg_ctx_create(int device, xsg_ctx** out);
void xsg_ctx_destroy(xsg_ctx* ctx);
/* Literal pattern, 1..XSG_MAX_PATTERN bytes, any byte values (or, with
 * XSG_FLAG_REGEX, a regular expression of the families above, at most XSG_MAX_REGEX bytes).  A LITERAL that contains
 * '\n' is served by every tag: the reference's line walk (search_wrappers.h:29-50,163-207: after a match, on to the
 * first '\n' at or behind its end) is then a chain from occurrence to occurrence, resolved on the device; the line
 * it reports for a match begins behind the last '\n' before the match (at the byte after the match's first byte if
 * that byte is a '\n' itself: previous_new_line_offset_relative_to_match, :111-123) and ends at the first '\n' at or
 * behind the match's end -- so xs::lines hands out strings that contain newlines.  xsg_count_async does not serve
 * XSG_COUNT_LINES for such a pattern (XSG_ENOTSUP: use xsg_count).  A regular EXPRESSION that can match '\n' keeps
 * to XSG_COUNT_MATCHES / XSG_MATCH_BYTE_OFFSETS (the line modes return XSG_ENOTSUP for it). */
int xsg_set_pattern(xsg_ctx* ctx, const void* pattern, size_t plen, uint32_t flags);
/* Would XSG_FLAG_REGEX accept this expression?  XSG_OK, or XSG_ENOTSUP with the
 * reason in xsg_last_error().  Needs no device (callers route on it the way the
 * reference routes on use_str_as_regex, utils/utils.h:17-25).  Optional outputs:
 * the number of byte positions, and their 256-bit sets (room for 32 x 8 uint32;
 * bit b of sets[8*k + b/32] set <=> position k accepts byte b; folded if
 * XSG_FLAG_IGNORE_CASE is in flags).  *positions == 0: an expression of variable
 * length, served by the automaton route below (no position-wise sets; whether it
 * can match a '\n' is xsg_regex_dfa.multiline). */
int xsg_regex_check(const void* expr, size_t n, uint32_t flags, uint32_t* positions, uint32_t* sets);
/* The same with the whole structure: number of alternatives, whether the expression is exact on ASCII data only
 * ('.', negated classes), and every alternative's sets (room for 64 x 8 uint32, alternative-major).  xsg_regex_check
 * returns the position-wise UNION of the alternatives. */
int xsg_regex_info(const void* expr, size_t n, uint32_t flags, uint32_t* positions, uint32_t* alternatives,
                   uint32_t* ascii_only, uint32_t* sets);
/* Expressions of VARIABLE length (x* x+ x? x{n,m}, lazy forms, alternatives of different lengths) are served by a
 * second route: the library compiles them into a forward (leftmost-first) and a reverse (longest) byte-class DFA
 * and k_rx_scan walks every line with them (csrc/xsg_regex.h; restrictions there: no empty matches, no anchors; an
 * expression with a set that accepts '\n' is walked chunk by chunk instead and serves the match tags only).
 * xsg_regex_check / xsg_regex_info return XSG_OK with *positions == 0 for such an expression.  This call hands out the automata (for inspection and for the host-side tests, which drive the tables
 * against another regex engine without a GPU): `info` always; `fwd` / `rev` (row-major, states x ncls entries, each
 * the NEXT STATE'S ROW OFFSET = state * ncls; state 0 is dead, states >= *_first_acc hold a match) if they have room
 * for the tables (cap_entries each).  XSG_ENOTSUP: the expression is not served by this route (it may still be a
 * fixed-length one that xsg_regex_info describes). */
typedef struct xsg_regex_dfa {
  uint32_t ncls, minlen, ascii_only;
  uint32_t multiline; /* a set accepts '\n': matches may span lines; the match tags only */
  /* PREFILTER: if prefix_positions != 0, every match starts with that many bytes accepted by one of
   * prefix_alternatives class sequences; the synchronous entry points (xsg_count, xsg_search, the jobs) then find
   * candidate positions with the scan kernel's class-sequence matcher and run the automaton at candidates only. */
  uint32_t prefix_positions, prefix_alternatives;
  /* FACTOR (expressions without a selective start that cannot match across lines): if factor_positions != 0, every match
   * CONTAINS that many consecutive bytes accepted by one class sequence (`\w+ing`: `\wing`); the synchronous entry
   * points mark the tiles in which a line with an occurrence starts, once per binding and pattern, and the line-walking
   * kernel skips every other tile. */
  uint32_t factor_positions;
  uint32_t fwd_states, fwd_start, fwd_first_acc;
  uint32_t rev_states, rev_start, rev_first_acc;
  uint8_t class_of[256];
} xsg_regex_dfa;
int xsg_regex_dfa_info(const void* expr, size_t n, uint32_t flags, xsg_regex_dfa* info, uint16_t* fwd, uint16_t* rev,
                       size_t cap_entries);
/* The prefilter of such an expression in the layout of xsg_regex_info (room for 64 x 8 uint32, alternative-major):
 * every match starts with *positions bytes that one of the *alternatives class sequences accepts.  *positions == 0:
 * the expression has no selective start, every entry point walks all lines (k_rx_scan). */
int xsg_regex_prefix(const void* expr, size_t n, uint32_t flags, uint32_t* positions, uint32_t* alternatives, uint32_t* sets);
/* ... and its factor (xsg_regex_dfa.factor_positions): one class sequence, room for 32 x 8 uint32. */
int xsg_regex_factor(const void* expr, size_t n, uint32_t flags, uint32_t* positions, uint32_t* sets);

/* ---- shards ---------------------------------------------------------------- */
/* d_base/capacity: device memory owned by the caller (hipMalloc, a torch
 * tensor's data_ptr(), ...), must stay valid, and its BYTES ARE IMMUTABLE FOR THE LIFETIME OF THE BINDING: the library
 * keeps results derived from them per binding (newline counts per tile, the measured hot filter and filter window of
 * a pattern, the tile marks and density verdicts of the regex prefilters, whether a pattern's lists fit the one-sync
 * route) and every later pass -- the stream-ordered xsg_count_async included -- relies on them.  A caller that refills
 * the buffer in place calls xsg_shard_rebind (same pointer and table are fine) or xsg_shard_invalidate before the
 * next search; searching rewritten bytes under an old binding can silently lose matches.
 * Requirements: d_base 16-byte aligned; for every chunk offset % 16 == 0 and
 * offset + round_up(length,16) <= capacity; chunks must not overlap and must
 * be listed in increasing offset order.  The table is copied. */
int xsg_shard_create(xsg_ctx* ctx, const void* d_base, uint64_t capacity, const xsg_chunk* chunks, uint64_t nchunks,
                     xsg_shard** out);
/* Re-point an existing shard at a new chunk table (same or different buffer)
 * without re-allocating its scratch when the new table is not larger. */
int xsg_shard_rebind(xsg_shard* shard, const void* d_base, uint64_t capacity, const xsg_chunk* chunks,
                     uint64_t nchunks);
/* The bytes behind the binding were rewritten in place (same buffer, same chunk table): forget everything derived
 * from the old bytes.  Cheaper than a rebind (no table upload); a rebind implies it. */
int xsg_shard_invalidate(xsg_shard* shard);
void xsg_shard_destroy(xsg_shard* shard);
/* line-index base of the whole shard for XSG_LINE_BASE_AUTO chunks (default 0):
 * the number of '\n' in all shards that precede this one in the file. */
int xsg_shard_set_line_base(xsg_shard* shard, uint64_t line_base);

/* ---- counting (replaces search::count, search_wrappers.h:163-185) ---------- */
/* Asynchronous: enqueues the scan on `stream` (a hipStream_t, NULL = the
 * ctx's own stream) and returns.  d_counters: device memory for
 * XSG_NUM_COUNTERS uint64 values, overwritten by the call.  Serves
 * XSG_COUNT_LINES and XSG_COUNT_MATCHES, each optionally | XSG_WITH_NEWLINES,
 * with one exception.  A pattern that can overlap itself -- a literal with a
 * border (`that`, `aa`, `abab`) or a class sequence two of whose occurrences may
 * overlap (`[a-z]{4}`, `t.e`; the test is conservative) -- needs the greedy
 * non-overlap walk over the ordered occurrence list.  Its XSG_COUNT_MATCHES
 * runs that list route entirely on the device, into arrays whose capacity is
 * twice the number of raw occurrences the last list pass on this shard saw (at
 * least 2^20): if there are more, all four counters read UINT64_MAX and
 * xsg_count (synchronous, sizes exactly, remembers the size) is the call to
 * make; | XSG_WITH_NEWLINES is not served for such a pattern (XSG_ENOTSUP).
 * For a LITERAL with a border the exception lapses once a synchronous call
 * (xsg_count, xsg_search) on this binding has established that its occurrences
 * do not overlap in the bound data -- the usual case in text; see xsg_count.
 * The first pass of a (binding, pattern) on a shard of 64 MiB or more also runs
 * the library's hot-filter probe (a few short launches and one stream sync,
 * DESIGN.md 3.1) -- unless another binding of the same buffer already measured
 * this pattern on this ctx; every later call only enqueues. */
int xsg_count_async(xsg_shard* shard, uint32_t mode, void* stream, uint64_t* d_counters);
/* The same pass with an error channel: *d_status (device memory, one uint64, overwritten by the call) receives 0 or a
 * combination of the XSG_STATUS_* bits below, and when it is not 0 all four counters read ZERO -- so a caller that adds
 * up or all-reduces counters (and the status word with them: the bits of different ranks add up to something non-zero)
 * can never mistake a refusal for a number.  xsg_count_async itself keeps its older convention (UINT64_MAX in all four
 * counters); new callers should use this form. */
#define XSG_STATUS_OK 0u
#define XSG_STATUS_OVERFLOW 1u  /* a self-overlapping pattern's bounded device-side list ran out of capacity: call xsg_count */
#define XSG_STATUS_NONASCII 2u  /* an expression that is exact on ASCII data only met a byte >= 0x80 (see XSG_FLAG_REGEX) */
int xsg_count_async_status(xsg_shard* shard, uint32_t mode, void* stream, uint64_t* d_counters, uint64_t* d_status);
/* Synchronous, any pattern, XSG_COUNT_MATCHES or XSG_COUNT_LINES
 * (| XSG_WITH_NEWLINES): result in host memory.  The first XSG_COUNT_MATCHES
 * (or match-offset search) of a literal with a border on a binding also runs
 * one count pass per border (at most three) for the word two overlapping
 * occurrences would spell: if the data holds none, every occurrence is a
 * match and the pattern is served like one without a border from then on. */
int xsg_count(xsg_shard* shard, uint32_t mode, uint64_t counters[XSG_NUM_COUNTERS]);

/* Split-phase form of xsg_count for host pipelines: _begin enqueues the pass on the ctx's stream and returns,
 * _end blocks until it is done and hands out the counters.  One pass in flight per shard. */
int xsg_count_begin(xsg_shard* shard, uint32_t mode);
int xsg_count_end(xsg_shard* shard, uint64_t counters[XSG_NUM_COUNTERS]);

/* ---- list results (replace byte_offsets_match/_line, line; :136-154,187-207) */
/* Runs the search for one of the list modes on the ctx stream and waits.
 * *n_results = number of elements (offsets / indices / lines).  Results stay
 * in device memory owned by the shard until the next search on it. */
int xsg_search(xsg_shard* shard, uint32_t mode, uint64_t* n_results);
/* Copy the uint64 results of the last XSG_MATCH_BYTE_OFFSETS /
 * XSG_LINE_BYTE_OFFSETS / XSG_LINE_INDICES search, ascending, to host memory. */
int xsg_result_u64(xsg_shard* shard, uint64_t* out, uint64_t cap);
/* The same without the copy into caller memory: *out points at *n values in pinned host memory owned by the shard,
 * valid until the next search on it (a dense needle returns hundreds of MB: D2H into pageable memory runs at a
 * sixth of the pinned rate). */
int xsg_result_u64_view(xsg_shard* shard, const uint64_t** out, uint64_t* n);
/* After an XSG_LINES search: total number of line bytes (no '\n's). */
int xsg_result_lines_size(xsg_shard* shard, uint64_t* n_lines, uint64_t* total_bytes);
/* line i = bytes[ starts[i] .. starts[i] + lengths[i] ), in file order;
 * `offsets` (optional) receives the global byte offset of each line start. */
int xsg_result_lines(xsg_shard* shard, uint64_t* lengths, char* bytes, uint64_t bytes_cap, uint64_t* offsets);

/* The same without copies into caller memory: *lengths / *offsets (n_lines values each) and *bytes (total_bytes, the lines
 * back to back) point into pinned host memory owned by the shard, valid until the next search on it.  A list that fit
 * the one-sync route is already there; a larger one is moved in three pinned D2H copies (a needle in most lines of
 * 10 GiB returns 3 GB: into pageable memory that is 0.3 s, pinned 0.07).  XSG_ENOMEM if the result needs more than
 * 16 GiB of pinned memory (use xsg_result_lines). */
int xsg_result_lines_view(xsg_shard* shard, const uint64_t** lengths, const char** bytes, const uint64_t** offsets,
                          uint64_t* n_lines, uint64_t* total_bytes);

/* newline count of the shard, available after an XSG_LINE_INDICES search (lets a
 * caller chain line-index bases from chunk to chunk without a second pass). */
int xsg_result_newlines(xsg_shard* shard, uint64_t* newlines);

/* ======================================================================== */
/* File searches: the host pipeline behind xs::extern_search                 */
/* ======================================================================== */
/* Replaces, for the literal path, what the reference runs inside
 * include/xsearch/Searcher.h:100-120 (worker loop: read -> search -> result.add)
 * with include/xsearch/tasks/readers.h:29-54 as the reader and
 * include/xsearch/ResultTypes.h:31-130 as the result container:
 *   - chunks are cut at '\n' (>= chunk_bytes, extended to the next newline; the
 *     layout the reference's .meta fixtures show) or taken from a metafile;
 *   - num_max_readers reader threads pread (and LZ4/ZSTD-decode) chunks into
 *     pinned host buffers; num_threads device workers take the filled buffers,
 *     hipMemcpyAsync them on their own HIP stream and run the scan.  The pinned
 *     buffers circulate between the two stages (readers never run ahead by more
 *     than num_threads + 1 chunks);
 *   - partial results are published in chunk order; a consumer may read them
 *     while the search is still running (blocking cursor below).
 * Errors: start fails for an unreadable file / bad metafile / no device; a
 * failure inside a worker stops the job and is returned by xsg_job_join. */
typedef struct xsg_job xsg_job;

typedef struct xsg_job_opts {
  uint32_t struct_size;     /* sizeof(xsg_job_opts), for ABI growth */
  uint32_t mode;            /* enum xsg_mode */
  uint32_t pattern_flags;   /* XSG_FLAG_* */
  int32_t device;           /* HIP device index */
  int32_t num_threads;      /* worker threads, >= 1 */
  int32_t num_max_readers;  /* concurrent reads, >= 1 */
  uint64_t chunk_bytes;     /* target chunk size without a metafile (default 16 MiB) */
  uint64_t chunk_begin;     /* search only chunks [chunk_begin, chunk_end) of the plan; 0,0 = all. */
  uint64_t chunk_end;       /* (how one rank of a multi-GPU job takes its contiguous range) */
} xsg_job_opts;

typedef struct xsg_job_stats {
  uint64_t bytes_scanned;  /* uncompressed bytes handed to the GPU */
  uint64_t bytes_read;     /* bytes read from disk/page cache */
  uint64_t chunks;
  double seconds_total;    /* start -> last worker done */
  double seconds_read;     /* summed over workers */
  double seconds_decompress;
  double seconds_device;   /* H2D + kernels + D2H, summed over workers */
  uint64_t newlines;       /* XSG_LINE_INDICES without metafile bases: '\n' in the searched chunks */
  uint64_t plan_chunks;    /* chunks in the whole plan of the file (before the range was applied) */
} xsg_job_stats;

void xsg_job_opts_init(xsg_job_opts* opts);
/* meta_file_path may be NULL.  Returns immediately; the search runs in the
 * job's own threads. */
int xsg_job_start(const void* pattern, size_t plen, const char* file_path, const char* meta_file_path,
                  const xsg_job_opts* opts, xsg_job** out);
/* Blocks until every worker has finished; returns the first worker error. */
int xsg_job_join(xsg_job* job);
/* Joins, then frees the job and everything it returned pointers into. */
void xsg_job_destroy(xsg_job* job);
/* Count tags: the count so far (final after join).  List tags: elements so far. */
int xsg_job_total(xsg_job* job, uint64_t* total);
/* Blocks until element `index` of the result sequence exists or the job has
 * finished.  *available = number of elements that exist now; *finished = 1 once
 * no more will come.  Count tags: element k = running total after k+1 chunks. */
int xsg_job_wait(xsg_job* job, uint64_t index, uint64_t* available, int* finished);
/* Non-blocking form of xsg_job_wait. */
int xsg_job_poll(xsg_job* job, uint64_t* available, int* finished);
int xsg_job_get_u64(xsg_job* job, uint64_t first, uint64_t n, uint64_t* out);
/* XSG_LINES: line `index`; *data stays valid until xsg_job_destroy. */
int xsg_job_get_line(xsg_job* job, uint64_t index, const char** data, uint64_t* len);
int xsg_job_stats_get(xsg_job* job, xsg_job_stats* stats);

/* ======================================================================== */
/* The lower seam: one chunk held in HOST memory                             */
/* ======================================================================== */
/* What a reference-style searcher functor calls for the chunk its reader just
 * produced (include/xsearch/tasks/searchers.h:38-93: IndexSearcher ->
 * byte_offsets_match, LineIndexSearcher -> byte_offsets_line, LineSearcher ->
 * line; concepts.h:36-39 SearcherC).  Results are CHUNK-LOCAL, exactly what the
 * search_wrappers.h functions return.  Thread-safe like the reference's shared
 * const functor (Searcher.h:110): concurrent callers use separate internal
 * slots (pinned staging buffer + device buffer + stream); callers beyond
 * max_slots wait for a free one.  include/xsearch/tasks/gpu_searchers.h wraps
 * these in functors with the reference's call signature. */
typedef struct xsg_host_searcher xsg_host_searcher;
int xsg_host_searcher_create(int device, const void* pattern, size_t plen, uint32_t flags, int max_slots,
                             xsg_host_searcher** out);
void xsg_host_searcher_destroy(xsg_host_searcher* hs);
/* search::count(data, pattern, skip_to_nl)  (search_wrappers.h:163-185) */
int xsg_host_count(xsg_host_searcher* hs, const void* data, uint64_t len, int skip_to_nl, uint64_t* count);
/* mode = XSG_MATCH_BYTE_OFFSETS / XSG_LINE_BYTE_OFFSETS / XSG_LINE_INDICES; *out is malloc'ed (xsg_free) */
int xsg_host_offsets(xsg_host_searcher* hs, uint32_t mode, const void* data, uint64_t len, uint64_t** out,
                     uint64_t* n);
/* search::line (:187-207): n lines, lengths[i] bytes each, packed in *bytes; both malloc'ed (xsg_free) */
int xsg_host_lines(xsg_host_searcher* hs, const void* data, uint64_t len, uint64_t** lengths, char** bytes,
                   uint64_t* n, uint64_t* nbytes);

/* ---- chunk plans and metafiles (host only: usable without a GPU) ------------- */
/* One record of the reference's metafile (decoded from test/files/ *.meta,
 * field names from metafile_cat.cpp:39-49). */
typedef struct xsg_file_chunk {
  uint64_t original_offset; /* offset in the uncompressed stream */
  uint64_t actual_offset;   /* offset in the file on disk */
  uint64_t original_size;
  uint64_t actual_size;     /* == original_size when not compressed */
  uint64_t first_line;      /* global index of the first line of the chunk, or XSG_LINE_BASE_AUTO */
  uint64_t n_mappings;      /* metafile only: number of (byte offset, line index) pairs */
} xsg_file_chunk;

#define XSG_COMPRESSION_NONE 1
#define XSG_COMPRESSION_ZSTD 2
#define XSG_COMPRESSION_LZ4 3

/* Newline-aligned plan of a plain file: every chunk is >= target_bytes long and
 * ends just after a '\n' (the last one ends at EOF).  *chunks is malloc'ed:
 * release with xsg_free. */
int xsg_plan_chunks(const char* file_path, uint64_t target_bytes, xsg_file_chunk** chunks, uint64_t* n);
/* Parse a metafile.  mappings (optional) receives all (globalByteOffset,
 * globalLineIndex) pairs back to back, 2 uint64 each, n_mappings per chunk. */
int xsg_meta_read(const char* meta_path, int32_t* compression, xsg_file_chunk** chunks, uint64_t* n,
                  uint64_t** mappings, uint64_t* n_mapping_pairs);
/* Preprocess `file_path` into the reference's on-disk layout: writes the
 * metafile and, for ZSTD/LZ4, the chunk-wise compressed data file
 * (data_out_path; ignored for XSG_COMPRESSION_NONE).  mapping_gap = minimum
 * distance in bytes between two mapping entries (fixtures: 500). */
int xsg_meta_write(const char* file_path, const char* meta_out_path, const char* data_out_path, int32_t compression,
                   uint64_t chunk_bytes, uint64_t mapping_gap, int hc);
void xsg_free(void* p);
/* Which decoder serves a compression type on this host (loads it if need be): "liblz4" / "libzstd" (the system's
 * shared library, what the reference links: Dockerfile:8), "built-in LZ4 block codec" (csrc/xsg_lz4.h, when the host
 * has no liblz4 or XSG_NO_LIBLZ4=1), "none" for XSG_COMPRESSION_NONE, "" if the type cannot be served here. */
const char* xsg_codec_name(int32_t compression);

/* ======================================================================== */
/* Multi-GPU: the one exchange step, RCCL over xGMI                           */
/* ======================================================================== */
/* The reference's only parallelism is N threads over chunks (include/xsearch/Searcher.h:141-145).  Across GPUs
 * the chunk list is cut into contiguous ranges (xsg_job_opts.chunk_begin/_end, or one shard per device) and
 * searched with no data-path collective; what is exchanged once per search is
 *   - the counter vector (xs::count / xs::count_lines): ncclAllReduce(sum) of XSG_NUM_COUNTERS uint64
 *   - one uint64 per device, its '\n' total (xs::line_indices without a metafile): ncclAllGather
 * 8-32 byte messages: latency-bound.  librccl is bound at run time (the copy the process already carries --
 * e.g. PyTorch-ROCm's -- else the system's); without one the create calls return XSG_ENOTSUP and the caller sums
 * on the host instead (include/xsearch/xsearch.h does exactly that). */
typedef struct xsg_comm xsg_comm;
#define XSG_COMM_ID_BYTES 128
/* rank form (one process per GPU): rank 0 makes an id, hands it to the others by any means (file, socket,
 * torch.distributed store), every rank then creates its end. */
int xsg_comm_unique_id(void* id, size_t cap);
int xsg_comm_create_rank(xsg_ctx* ctx, int nranks, int rank, const void* id, xsg_comm** out);
/* local form (one process, n GPUs): one communicator over the devices of ctxs[0..n), all distinct. */
int xsg_comm_create_local(xsg_ctx* const* ctxs, int n, xsg_comm** out);
/* Destroy a communicator BEFORE the contexts it was made from are used for nothing else: the collectives run on the
 * contexts' streams, so a ctx must outlive every collective still queued on it (destroying the ctx first is tolerated by
 * xsg_comm_destroy itself -- it keeps the device numbers by value -- but not by a collective in flight). */
void xsg_comm_destroy(xsg_comm* comm);
int xsg_comm_size(xsg_comm* comm, int* nranks, int* rank /* -1 in the local form */);
/* path of the librccl in use ("" if none) */
const char* xsg_comm_library(void);
/* rank form: in-place sum over the ranks of k uint64 device words (e.g. what xsg_count_async wrote), enqueued
 * on `stream` (a hipStream_t; NULL = the ctx's own) -- stream-ordered behind the count that produced them. */
int xsg_reduce_counts_async(xsg_comm* comm, uint64_t* d_counters, int k, void* stream);
/* either form, blocking: totals[0..k) = the sums; the device vectors are left summed in place.
 * rank form: d_counters[0] is this rank's vector; local form: d_counters[i] lives on ctxs[i]'s device. */
int xsg_reduce_counts(xsg_comm* comm, uint64_t* const* d_counters, int k, uint64_t* totals);
/* either form, blocking: out[0..nranks) = every rank's / device's value in rank order (mine[0], or mine[i] for
 * ctxs[i]); the exclusive prefix -- the line-index base of each range -- is the caller's to take. */
int xsg_allgather_u64(xsg_comm* comm, const uint64_t* mine, uint64_t* out);

/* NUMA placement of a device: its node (-1 if the platform does not say) and the CPUs next to it (the PCI device's
 * local_cpulist, e.g. "0-63,128-191").  The reader and device-worker threads of a job are bound to those CPUs
 * (XSG_NUMA=0 switches that off); its pinned buffers are allocated on that node by hipHostMalloc. */
int xsg_device_numa(int device, int* node, char* cpulist, size_t cap);

/* The count of a search that fanned out over several devices (one job each, contiguous chunk ranges): the sum of
 * the jobs' totals, exchanged over RCCL (one ncclAllReduce of one uint64 over a per-process communicator of the
 * jobs' devices, created at first use).  *via_rccl (optional) = 1 if it went that way, 0 if the sum was taken on
 * the host instead: no librccl, a device listed twice, a single job, or XS_REDUCE=host.  Call after the joins. */
int xsg_jobs_reduce_total(xsg_job* const* jobs, int n, uint64_t* total, int* via_rccl);

/* ---- diagnostics ------------------------------------------------------------ */
/* Name of the device the ctx is bound to (e.g. "gfx950..."), CU count. */
int xsg_ctx_info(xsg_ctx* ctx, char* arch, size_t arch_cap, int* compute_units, uint64_t* hbm_bytes);
/* Time the dominant kernel alone: runs the bulk scan kernel of the given count
 * mode `iters` times back to back on the ctx stream between two HIP events and
 * returns the average milliseconds per launch (bench.py's roofline figure). */
int xsg_time_scan_kernel(xsg_shard* shard, uint32_t mode, int iters, float* avg_ms);
/* Name of the bulk-kernel instantiation the next pass of `mode` launches on this shard with the current pattern
 * ("xsg::k_scan<KIND, WANT_NL, WANT_LINES, EMIT, LOADS, ICASE> stagger=N"): what a profile of that pass shows. */
int xsg_scan_kernel_name(xsg_shard* shard, uint32_t mode, char* out, size_t cap);
/* Measure and fix the bulk kernel's wave stagger for this shard, the current pattern and `mode` (a handful of
 * launches; replaces the per-variant default until the shard is destroyed or tuned again).  *chosen (optional)
 * receives the value, or UINT32_MAX when the default was kept (shard under 1 GiB, or XSG_TUNE set). */
int xsg_shard_tune(xsg_shard* shard, uint32_t mode, uint32_t* chosen);
/* (The read-only HBM probes of round 1 -- xsg_time_read_ceiling and friends -- moved to libxsg_diag.so,
 * x-search_amd/csrc/diag/xsg_diag.hip: the product library holds search code only.) */

#ifdef __cplusplus
}
#endif
#endif /* XSG_H */
