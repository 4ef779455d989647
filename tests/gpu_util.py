"""Helpers for the -m gpu tests: put chunks into device memory (torch is only
the allocator here) and run the C-ABI searches through x-search_amd/xsg.py."""
import numpy as np

import corpus
import xsg


def upload(blocks, global_offsets=None, line_bases=None):
    """blocks: list of uint8 arrays -> (device tensor, chunk table)."""
    import torch
    lengths = [int(b.size) for b in blocks]
    off, ln, cap = corpus.chunk_table(lengths)
    host = np.zeros(max(cap, 256), dtype=np.uint8)
    for o, b in zip(off, blocks):
        host[int(o):int(o) + b.size] = b
    t = torch.from_numpy(host).to("cuda:0")
    chunks = xsg.make_chunks(off, ln, global_offsets, line_bases)
    return t, chunks


class GpuSearch:
    """One context + one shard, re-bound per case."""

    def __init__(self, hot=None, probe=False):
        """hot=0/1 pins the hot filter of the 8-byte-window kinds (XSG_HOT, read when the context is created):
        the test shards are far below the size at which the library measures and picks one itself.
        probe=True lets it measure on shards of any size (XSG_PROBE_MIN_BYTES=0): on kilobyte shards the timings are
        noise, so both filters and every candidate window of a long pattern get picked at random -- which is the
        point: whatever the probe chooses, the results must not change."""
        import os
        env = {}
        if hot is not None:
            env["XSG_HOT"] = str(hot)
        if probe:
            env["XSG_PROBE_MIN_BYTES"] = "0"
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            self.ctx = xsg.Context(0)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        self.shard = None
        self.keep = None

    def bind(self, blocks, global_offsets=None, line_bases=None):
        t, chunks = upload(blocks, global_offsets, line_bases)
        self.keep = t
        if self.shard is None:
            self.shard = xsg.Shard(self.ctx, t.data_ptr(), t.numel(), chunks)
        else:
            self.shard.rebind(t.data_ptr(), t.numel(), chunks)
        return chunks

    def all_modes(self, pattern: bytes, flags=0, lines=True):
        """-> dict with every xs:: tag's result for the bound shard (lines=False: the pattern can match a
        newline, so only the match tags apply)."""
        self.ctx.set_pattern(pattern, flags)
        s = self.shard
        out = {}
        c = s.count(xsg.COUNT_MATCHES | xsg.WITH_NEWLINES)
        out["count_matches"] = int(c[xsg.CTR_MATCHES])
        out["newlines"] = int(c[xsg.CTR_NEWLINES])
        out["bytes"] = int(c[xsg.CTR_BYTES])
        out["match_byte_offsets"] = s.search_u64(xsg.MATCH_BYTE_OFFSETS).tolist()
        if lines:  # (a LITERAL that contains '\n' is served by the line tags too: its walk is a chain of occurrences)
            out["count_lines"] = int(s.count(xsg.COUNT_LINES)[xsg.CTR_LINES])
            out["line_byte_offsets"] = s.search_u64(xsg.LINE_BYTE_OFFSETS).tolist()
            out["line_indices"] = s.search_u64(xsg.LINE_INDICES).tolist()
            ls, lo = s.search_lines()
            out["lines"] = ls
            out["lines_offsets"] = lo.tolist()
            # the zero-copy form of the same result (xsg_result_lines_view) must hold the same lines
            vl, vb, vo = s.search_lines_view()
            ends = np.cumsum(vl.astype(np.int64)) if vl.size else np.zeros(0, dtype=np.int64)
            raw = vb.tobytes()
            assert [raw[int(e) - int(n):int(e)] for e, n in zip(ends, vl)] == ls, "xsg_result_lines_view: lines differ"
            assert vo.tolist() == out["lines_offsets"], "xsg_result_lines_view: offsets differ"
        # once more: what a SECOND count launches may differ from the first (a needle the first count found dense runs
        # with another stagger, a 4..8-byte one byte-parallel: x-search_amd/csrc/xsg_kernels.hip, dense_bytes_route)
        c2 = s.count(xsg.COUNT_MATCHES | xsg.WITH_NEWLINES)
        assert [int(x) for x in c2] == [int(x) for x in c], "the second count differs from the first"
        if "count_lines" in out:
            assert int(s.count(xsg.COUNT_LINES)[xsg.CTR_LINES]) == out["count_lines"], "the second count_lines differs"
        return out


def oracle_all_modes(oracle, blocks, pattern: bytes, exact=False, global_offsets=None, line_bases=None,
                     ignore_case=False):
    """The same dict from the CPU oracle, chunk by chunk (chunks are independent
    units: include/xsearch/Searcher.h:100-120 hands each to the searcher alone)."""
    orig_blocks = blocks
    if ignore_case:  # search(toLower(chunk), toLower(pattern)); reported lines keep their original bytes
        blocks = [oracle.lower(b) for b in blocks]
        pattern = oracle.lower(pattern).tobytes()
    oracle.set_exact(bool(exact))
    try:
        out = {"count_matches": 0, "newlines": 0, "bytes": 0, "match_byte_offsets": [], "count_lines": 0,
               "line_byte_offsets": [], "line_indices": [], "lines": [], "lines_offsets": []}
        goff, nl_before = 0, 0
        for i, b in enumerate(blocks):
            g = goff if global_offsets is None else int(global_offsets[i])
            lb = nl_before if line_bases is None else int(line_bases[i])
            out["count_matches"] += oracle.count(b, pattern, False)
            out["match_byte_offsets"] += [int(x) + g for x in oracle.byte_offsets_match(b, pattern)]
            if True:  # (patterns that contain '\n' included: the reference's walk is defined for any string)
                out["count_lines"] += oracle.count(b, pattern, True)
                out["line_byte_offsets"] += [int(x) + g for x in oracle.byte_offsets_line(b, pattern)]
                out["line_indices"] += [int(x) for x in oracle.line_indices(b, pattern, lb)]
                beg, ln = oracle.lines_spans(b, pattern)
                out["lines"] += [orig_blocks[i][int(s):int(s + l)].tobytes() for s, l in zip(beg, ln)]
                out["lines_offsets"] += [int(s) + g for s in beg]
            nl = oracle.count_newlines(b)
            out["newlines"] += nl
            out["bytes"] += int(b.size)
            goff += int(b.size)
            nl_before += nl
        return out
    finally:
        oracle.set_exact(False)


def oracle_regex_all_modes(oracle, blocks, expr: bytes, ignore_case=False, global_offsets=None, line_bases=None):
    """oracle_all_modes for a class-sequence regex (search_wrappers.h:63-103,209-271 restated in oracle/)."""
    from xs_oracle import RegexProgram, UnsupportedRegex, compile_class_sequence
    try:
        cs = compile_class_sequence(expr, ignore_case)
    except UnsupportedRegex:
        # not a class sequence: the variable-length family (the product's automaton route); raises if that refuses too
        prog = RegexProgram(expr, ignore_case)
        return oracle_rx_all_modes(oracle, blocks, prog, global_offsets, line_bases), not prog.multiline
    orig_blocks = blocks
    if ignore_case:
        blocks = [oracle.lower(b) for b in blocks]
    with_lines = not any(cs.accepts(k, 10) for k in range(cs.plen))
    out = {"count_matches": 0, "newlines": 0, "bytes": 0, "match_byte_offsets": []}
    if with_lines:
        out.update({"count_lines": 0, "line_byte_offsets": [], "line_indices": [], "lines": [], "lines_offsets": []})
    goff, nl_before = 0, 0
    for i, b in enumerate(blocks):
        g = goff if global_offsets is None else int(global_offsets[i])
        lb = nl_before if line_bases is None else int(line_bases[i])
        out["count_matches"] += oracle.regex_count(b, cs, False)
        out["match_byte_offsets"] += [int(x) + g for x in oracle.regex_byte_offsets_match(b, cs)]
        if with_lines:
            out["count_lines"] += oracle.regex_count(b, cs, True)
            out["line_byte_offsets"] += [int(x) + g for x in oracle.regex_byte_offsets_line(b, cs)]
            out["line_indices"] += [int(x) for x in oracle.regex_line_indices(b, cs, lb)]
            beg, ln = oracle.regex_lines_spans(b, cs)
            out["lines"] += [orig_blocks[i][int(s):int(s + l)].tobytes() for s, l in zip(beg, ln)]
            out["lines_offsets"] += [int(s) + g for s in beg]
        nl = oracle.count_newlines(b)
        out["newlines"] += nl
        out["bytes"] += int(b.size)
        goff += int(b.size)
        nl_before += nl
    return out, with_lines


def oracle_rx_all_modes(oracle, blocks, prog, global_offsets=None, line_bases=None):
    """the same dict for a variable-length expression (oracle/xs_oracle.py: RegexProgram; the sets are closed under
    case, so the data is searched as it is)"""
    out = {"count_matches": 0, "newlines": 0, "bytes": 0, "match_byte_offsets": [], "count_lines": 0,
           "line_byte_offsets": [], "line_indices": [], "lines": [], "lines_offsets": []}
    goff, nl_before = 0, 0
    for i, b in enumerate(blocks):
        g = goff if global_offsets is None else int(global_offsets[i])
        lb = nl_before if line_bases is None else int(line_bases[i])
        m = oracle.rx_byte_offsets(b, prog, False)
        out["count_matches"] += int(m.size)
        out["match_byte_offsets"] += [int(x) + g for x in m]
        nl = oracle.count_newlines(b)
        out["newlines"] += nl
        out["bytes"] += int(b.size)
        goff += int(b.size)
        nl_before += nl
        if prog.multiline:
            continue
        out["count_lines"] += oracle.rx_count(b, prog, True)
        out["line_byte_offsets"] += [int(x) + g for x in oracle.rx_byte_offsets(b, prog, True, True)]
        out["line_indices"] += [int(x) for x in oracle.rx_line_indices(b, prog, lb)]
        beg, ln = oracle.rx_lines_spans(b, prog)
        out["lines"] += [b[int(s):int(s + l)].tobytes() for s, l in zip(beg, ln)]
        out["lines_offsets"] += [int(s) + g for s in beg]
    if prog.multiline:
        for k in ("count_lines", "line_byte_offsets", "line_indices", "lines", "lines_offsets"):
            out.pop(k)
    return out
