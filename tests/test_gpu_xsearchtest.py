"""The reference's integration suite (test/src/xsearchTest.cpp: 96 TESTs, all inside a comment there because the
100 MB corpus is missing) replayed on the GPU against ITS OWN golden vectors, on the stand-in corpus
(tests/sample_standin.py: a file that agrees with everything the snapshot records about test/files/sample.txt).

Matrix, as in the reference: {join, live} x {plain, plain + meta, lz4 + meta, zst + meta} x {count_matches, count_lines,
line_byte_offsets, match_byte_offsets, line_indices, lines} x {literal `Sherlock`, regex `She[r ]lock`} x
{case, ignore-case} x {1, 4 threads}.  Expected values: tests/golden/ref_xsearchtest_vectors.json = xsearchTest.cpp:17-335
verbatim.  The compressed payloads and the metafiles are written by the product's own preprocessor (xsg_meta_write; no
payload fixture exists: SURVEY 8c) -- and the chunk table it writes for this file is the reference's.
"""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import golden_util as G
import sample_standin as S
import xsg

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
CLI = ROOT / "tests" / "cpp" / "build" / "extern_search_cli"
MAP = G.load("ref_xsearchtest_mapping_brackets.json")
TAGS = {"count_matches": xsg.COUNT_MATCHES, "count_lines": xsg.COUNT_LINES, "line_byte_offsets": xsg.LINE_BYTE_OFFSETS,
        "match_byte_offsets": xsg.MATCH_BYTE_OFFSETS, "line_indices": xsg.LINE_INDICES, "lines": xsg.LINES}
FAMILIES = {"literal_case": (b"Sherlock", 0), "literal_icase": (b"Sherlock", xsg.FLAG_IGNORE_CASE),
            "regex_case": (b"She[r ]lock", xsg.FLAG_REGEX), "regex_icase": (b"She[r ]lock", xsg.FLAG_REGEX | xsg.FLAG_IGNORE_CASE)}


@pytest.fixture(scope="module")
def sample():
    d = Path("/dev/shm") / f"xsg_standin_{os.getpid()}"
    d.mkdir(exist_ok=True)
    txt = d / "sample.txt"
    S.build(str(txt))
    files = {"plain": (str(txt), None)}
    for comp, name in ((xsg.COMPRESSION_NONE, "meta"), (xsg.COMPRESSION_LZ4, "xslz4"), (xsg.COMPRESSION_ZSTD, "xszst")):
        data, meta = d / f"sample.{name}", d / f"sample.{name}.meta"
        xsg.meta_write(str(txt), str(meta), str(data), comp, 16 << 20, 500)
        files[name] = (str(txt) if comp == xsg.COMPRESSION_NONE else str(data), str(meta))
    yield files
    for f in d.iterdir():
        f.unlink()
    d.rmdir()


def as_py(tag, value):
    if tag.startswith("count"):
        return int(value)
    if tag == "lines":
        return [x.decode("latin-1") for x in value]
    return [int(x) for x in value]


def test_the_written_metafiles_carry_the_references_chunk_table(sample):
    for name in ("meta", "xslz4", "xszst"):
        comp, chunks = xsg.meta_read(sample[name][1])
        assert [(int(c["original_offset"]), int(c["original_size"]), int(c["first_line"])) for c in chunks] == \
               [(c["original_offset"], c["original_size"], c["first_line"]) for c in MAP["chunks"]], name


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("source", ["plain", "meta", "xslz4", "xszst"])
@pytest.mark.parametrize("family", list(FAMILIES))
def test_join_search_and_copy_results(sample, family, source, threads):
    """xsearchTest.cpp "join search and copy results": every tag, joined, against the reference's vectors"""
    pattern, flags = FAMILIES[family]
    want = S.expected(family)
    path, meta = sample[source]
    for tag, mode in TAGS.items():
        j = xsg.Job(pattern, path, mode, meta_path=meta, num_threads=threads, num_max_readers=threads, flags=flags)
        got = as_py(tag, j.result())
        j.close()
        assert got == want[tag], (family, source, threads, tag)


@pytest.mark.parametrize("source", ["plain", "xslz4"])
@pytest.mark.parametrize("family", list(FAMILIES))
def test_live_iteration(sample, family, source):
    """xsearchTest.cpp "live search": the iterator blocks per element; count tags yield running totals whose last value
    is the count (:735-739), vector tags yield flat elements (:888-890)"""
    pattern, flags = FAMILIES[family]
    want = S.expected(family)
    path, meta = sample[source]
    for tag, mode in TAGS.items():
        j = xsg.Job(pattern, path, mode, meta_path=meta, num_threads=4, num_max_readers=2, flags=flags)
        seen = list(j)
        j.join()
        j.close()
        if tag.startswith("count"):
            assert len(seen) == len(MAP["chunks"]) and seen[-1] == want[tag] and seen == sorted(seen)
        else:
            assert as_py(tag, seen) == want[tag], (family, source, tag)


@pytest.mark.parametrize("how", ["join", "live"])
def test_through_the_cpp_api(sample, how):
    """the same through xs::extern_search<Tag> (include/xsearch/xsearch.h), which routes `She[r ]lock` to the regex
    matcher the way the reference does (utils/utils.h:17-25)"""
    if not CLI.exists():
        pytest.fail(f"{CLI} not built (make -C tests/cpp)")
    for family, (pattern, flags) in FAMILIES.items():
        want = S.expected(family)
        env = dict(os.environ)
        if flags & xsg.FLAG_IGNORE_CASE:
            env["XS_IGNORE_CASE"] = "1"
        for tag in TAGS:
            for src, threads in (("plain", "1"), ("xszst", "4")):
                path, meta = sample[src]
                r = subprocess.run([str(CLI), tag, how, pattern.decode(), path, meta or "-", threads, "2"],
                                   capture_output=True, env=env, timeout=300)
                assert r.returncode == 0, r.stderr.decode()
                out = r.stdout.split(b"\n")[:-1]
                if tag.startswith("count"):
                    assert int(out[0]) == want[tag], (family, tag, src)
                elif tag == "lines":
                    assert [x.decode("latin-1") for x in out] == want[tag], (family, tag, src)
                else:
                    assert [int(x) for x in out] == want[tag], (family, tag, src)
