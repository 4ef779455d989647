"""Host-side file logic of the pipeline (no GPU): newline-aligned chunk plans,
metafile reader/writer.  The metafile format is pinned by the reference's four
.meta fixtures (test/files/): they are parsed in place when the
reference tree is mounted and compared with tests/golden/ref_metafile_summary.json."""
import hashlib
import json
import os
from pathlib import Path

import numpy as np
import pytest

import corpus
import golden_util as G
import xsg

REF_FILES = Path("/root/reference/test/files")


@pytest.fixture(scope="module")
def text_file(tmp_path_factory):
    d = tmp_path_factory.mktemp("xsg")
    blocks = [corpus.text_block(31, i, 700_000 + 13 * i, needle_rate=1e-3) for i in range(4)]
    data = np.concatenate(blocks)
    p = d / "corpus.txt"
    data.tofile(p)
    return p, data


def test_plan_is_newline_aligned_and_covers_the_file(text_file):
    p, data = text_file
    for target in (1 << 16, 1 << 20, 1 << 30):
        pl = xsg.plan_chunks(str(p), target)
        assert int(pl["original_size"].sum()) == data.size
        pos = 0
        for i, c in enumerate(pl):
            assert int(c["original_offset"]) == pos == int(c["actual_offset"])
            n = int(c["original_size"])
            last = i == len(pl) - 1
            assert n >= target or last
            assert data[pos + n - 1] == 10  # ends just after a newline (the corpus is '\n'-terminated)
            if not last:
                # minimal: no newline in [pos+target-1, end-1)
                assert not (data[pos + target - 1:pos + n - 1] == 10).any()
            pos += n


def test_plan_edge_cases(tmp_path):
    e = tmp_path / "empty"
    e.write_bytes(b"")
    assert len(xsg.plan_chunks(str(e), 100)) == 0
    f = tmp_path / "no_nl"
    f.write_bytes(b"x" * 1000)
    pl = xsg.plan_chunks(str(f), 100)
    assert len(pl) == 1 and int(pl[0]["original_size"]) == 1000
    g = tmp_path / "unterminated"
    g.write_bytes(b"ab\n" * 100 + b"tail")
    pl = xsg.plan_chunks(str(g), 30)
    assert int(pl["original_size"].sum()) == 304
    assert all(int(c["original_size"]) % 3 == 0 for c in pl[:-1])
    with pytest.raises(xsg.XsgError) as ei:
        xsg.plan_chunks(str(tmp_path / "missing"), 100)
    assert ei.value.code == xsg.EIO


@pytest.mark.parametrize("comp,name", [(xsg.COMPRESSION_NONE, "none"), (xsg.COMPRESSION_LZ4, "lz4"),
                                       (xsg.COMPRESSION_ZSTD, "zst")])
def test_meta_write_read_roundtrip(text_file, tmp_path, comp, name):
    p, data = text_file
    meta, out = tmp_path / f"c.{name}.meta", tmp_path / f"c.{name}"
    xsg.meta_write(str(p), str(meta), str(out), comp, 1 << 19, 500)
    ctype, chunks, maps = xsg.meta_read(str(meta), True)
    assert ctype == comp
    plan = xsg.plan_chunks(str(p), 1 << 19)
    assert chunks["original_offset"].tolist() == plan["original_offset"].tolist()
    assert chunks["original_size"].tolist() == plan["original_size"].tolist()
    # mapping entries: (line start offset, 0-based line index); first entry of a chunk = chunk start;
    # consecutive entries of a chunk >= 500 bytes apart (fixtures: min 500 / max 818)
    nl = np.flatnonzero(data == 10)
    line_starts = np.concatenate([[0], nl + 1])
    at = 0
    for c in chunks:
        m = maps[at:at + int(c["n_mappings"])]
        at += int(c["n_mappings"])
        assert int(m[0][0]) == int(c["original_offset"]) and int(c["first_line"]) == int(m[0][1])
        idx = np.searchsorted(line_starts, m[:, 0])
        assert (line_starts[idx] == m[:, 0]).all()       # every entry is a line start
        assert (idx == m[:, 1]).all()                     # ... with its 0-based line index
        gaps = np.diff(m[:, 0].astype(np.int64))
        assert (gaps[1:] >= 500).all() if len(gaps) > 1 else True
    if comp == xsg.COMPRESSION_NONE:
        assert chunks["actual_size"].tolist() == chunks["original_size"].tolist()
        assert chunks["actual_offset"].tolist() == chunks["original_offset"].tolist()
    else:
        assert int(chunks["actual_size"].sum()) == out.stat().st_size < data.size
        assert chunks["actual_offset"].tolist() == np.concatenate([[0], np.cumsum(chunks["actual_size"])[:-1]]).tolist()


def test_meta_read_rejects_garbage(tmp_path):
    bad = tmp_path / "bad.meta"
    bad.write_bytes(b"\x07\x00\x00\x00")
    with pytest.raises(xsg.XsgError) as e:
        xsg.meta_read(str(bad))
    assert e.value.code == xsg.EIO
    bad.write_bytes(b"\x01\x00\x00\x00" + b"\x00" * 17)
    with pytest.raises(xsg.XsgError):
        xsg.meta_read(str(bad))


def _meta_bytes(ctype, records):
    """records: (original_offset, actual_offset, original_size, actual_size) -- no mapping entries"""
    import struct
    out = struct.pack("<i", ctype)
    for r in records:
        out += struct.pack("<5Q", *r, 0)
    return out


@pytest.mark.parametrize("name,ctype,records,data_len", [
    # uncompressed file whose record asks for far more bytes than the chunk it declares (heap overflow of the
    # pinned buffer if trusted: the buffer is sized from original_size, the read from actual_size)
    ("actual_ne_original", 1, [(0, 0, 16, 1 << 20)], 1 << 21),
    # actual_offset + actual_size wraps around 2^64
    ("offset_wraps", 1, [(0, (1 << 64) - 8, 16, 16)], 4096),
    # sizes near 2^64: the rounded-up buffer size would wrap to a few hundred bytes
    ("size_wraps", 1, [(0, 0, (1 << 64) - 10, (1 << 64) - 10)], 4096),
    ("lz4_block_over_int32", 3, [(0, 0, 1 << 33, 64)], 4096),
    ("beyond_eof", 2, [(0, 4000, 100, 200)], 4096),
    ("compressed_chunk_without_bytes", 3, [(0, 0, 100, 0)], 4096),
])
def test_job_start_refuses_a_malformed_metafile(tmp_path, name, ctype, records, data_len):
    """Every metafile record is validated before a buffer is sized from it or a byte is read through it
    (x-search_amd/csrc/xsg_file.cpp, job_start_impl); the file checks run before the device check, so this holds
    on a host without a GPU too."""
    data = tmp_path / f"{name}.txt"
    data.write_bytes(b"a line of text\n" * (data_len // 15 + 1))
    meta = tmp_path / f"{name}.meta"
    meta.write_bytes(_meta_bytes(ctype, records))
    with pytest.raises(xsg.XsgError) as e:
        xsg.Job(b"text", str(data), xsg.COUNT_MATCHES, meta_path=str(meta))
    assert e.value.code == xsg.EIO, str(e.value)


def _summary(path):
    ctype, chunks, maps = xsg.meta_read(str(path), True)
    return {
        "compression_type": ctype,
        "file_bytes": os.path.getsize(path),
        "original_size": [int(x) for x in chunks["original_size"]],
        "actual_size_total": int(chunks["actual_size"].sum()),
        "n_mappings": [int(x) for x in chunks["n_mappings"]],
        "first_line": [int(x) for x in chunks["first_line"]],
        "last_mapping": [int(x) for x in maps[-1]],
        "mapping_gap_min_max": [int(np.diff(maps[:int(chunks["n_mappings"][0]), 0].astype(np.int64))[1:].min()),
                                int(np.diff(maps[:int(chunks["n_mappings"][0]), 0].astype(np.int64)).max())],
        "mappings_sha256": hashlib.sha256(maps.tobytes()).hexdigest(),
    }


def test_reference_metafile_fixtures():
    """The four fixtures the reference's tests use must parse to exactly EOF with
    the values recorded in tests/golden/ref_metafile_summary.json."""
    want = G.load("ref_metafile_summary.json")
    assert want["sample.meta"]["compression_type"] == 1 and want["sample.xszst.meta"]["compression_type"] == 2
    assert want["sample.xslz4.meta"]["compression_type"] == want["sample.xslz4hc.meta"]["compression_type"] == 3
    for v in want.values():
        assert sum(v["original_size"]) == 100_000_000 and v["file_bytes"] == 3_060_484
        assert v["last_mapping"] == [99_999_691, 3_447_129]
    if not REF_FILES.exists():
        pytest.skip("reference tree not mounted: summary checked, fixtures not re-parsed")
    for name, v in want.items():
        assert _summary(REF_FILES / name) == v, name


if __name__ == "__main__":  # regenerate the summary (build container only)
    out = {n: _summary(REF_FILES / n) for n in ("sample.meta", "sample.xslz4.meta", "sample.xslz4hc.meta",
                                                  "sample.xszst.meta")}
    (G.GOLDEN / "ref_metafile_summary.json").write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1)[:600])


def test_xsmeta_cli_cat_and_write(text_file, tmp_path):
    """tools/xsmeta: the metafile_cat-like dump (metafile_cat.cpp:23-52 field names)
    and the preprocessor, both GPU-free."""
    import subprocess
    exe = Path(__file__).resolve().parents[1] / "tools" / "build" / "xsmeta"
    if not exe.exists():
        pytest.skip("tools/build/xsmeta not built (make -C tools)")
    p, data = text_file
    meta, out = tmp_path / "t.meta", tmp_path / "t.xslz4"
    r = subprocess.run([str(exe), "write", str(p), "--meta", str(meta), "--data", str(out), "--lz4", "--chunk-bytes",
                        str(1 << 19)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([str(exe), "cat", str(meta)], capture_output=True)
    assert r.returncode == 0
    txt = r.stdout.decode()
    ctype, chunks = xsg.meta_read(str(meta))
    assert txt.startswith("Compression Type: LZ4 (4)")
    assert txt.count("Chunk ") == len(chunks)
    assert f"  original: {int(chunks[1]['original_offset'])}" in txt
    r = subprocess.run([str(exe), "cat", str(tmp_path / "nope.meta")], capture_output=True)
    assert r.returncode == 1 and b"cannot open" in r.stderr


def test_builtin_lz4_codec_selftest():
    """x-search_amd/csrc/xsg_lz4.h (the fallback when the host has no liblz4): round trips, interop with the host's
    liblz4 in both directions, hostile blocks -- under AddressSanitizer + UBSan (tests/cpp/lz4_selftest.cpp)."""
    import subprocess
    exe = Path(__file__).resolve().parent / "cpp" / "build" / "lz4_selftest"
    if not exe.exists():
        pytest.fail(f"{exe} not built (make -C tests/cpp)")
    r = subprocess.run([str(exe)], capture_output=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr).decode()
    assert b"lz4 selftest ok" in r.stdout


def test_builtin_lz4_writes_what_liblz4_reads(text_file, tmp_path):
    """xsmeta write with XSG_NO_LIBLZ4=1 (built-in encoder): same chunk table as with liblz4, a valid data file."""
    import os
    import subprocess
    exe = Path(__file__).resolve().parents[1] / "tools" / "build" / "xsmeta"
    p, data = text_file
    outs = {}
    for tag, env in (("lib", {}), ("own", {"XSG_NO_LIBLZ4": "1"})):
        meta, out = tmp_path / f"{tag}.meta", tmp_path / f"{tag}.xslz4"
        r = subprocess.run([str(exe), "write", str(p), "--meta", str(meta), "--data", str(out), "--lz4", "--chunk-bytes",
                            str(1 << 19)], capture_output=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()
        outs[tag] = (xsg.meta_read(str(meta))[1], out.stat().st_size)
    a, b = outs["lib"][0], outs["own"][0]
    assert len(a) == len(b) and (a["original_offset"] == b["original_offset"]).all() and (a["original_size"] == b["original_size"]).all()
    assert 0 < outs["own"][1] < data.size  # it does compress text


def test_aligned_file_reader_functor(text_file, tmp_path):
    """include/xsearch/tasks/aligned_reader.h: the ReaderC-shaped chunk reader (reference: tasks/readers.h:29-54,
    concepts.h:24-27) hands out the newline-aligned plan, from 4 threads sharing one instance, with and without
    a metafile (tests/cpp/reader_selftest.cpp)."""
    import subprocess
    exe = Path(__file__).resolve().parent / "cpp" / "build" / "reader_selftest"
    if not exe.exists():
        pytest.fail(f"{exe} not built (make -C tests/cpp)")
    p, data = text_file
    for target in (1 << 16, 1 << 19, 1 << 30):
        r = subprocess.run([str(exe), str(p), str(target)], capture_output=True, timeout=120)
        assert r.returncode == 0 and b"reader selftest ok" in r.stdout, (r.stdout + r.stderr).decode()
    meta = tmp_path / "plain.meta"
    xsg.meta_write(str(p), str(meta), None, xsg.COMPRESSION_NONE, 1 << 18, 500)
    r = subprocess.run([str(exe), str(p), str(1 << 18), str(meta)], capture_output=True, timeout=120)
    assert r.returncode == 0 and b"reader selftest ok" in r.stdout, (r.stdout + r.stderr).decode()
    lz = tmp_path / "c.meta"
    xsg.meta_write(str(p), str(lz), str(tmp_path / "c.xslz4"), xsg.COMPRESSION_LZ4, 1 << 18, 500)
    r = subprocess.run([str(exe), str(tmp_path / "c.xslz4"), str(1 << 18), str(lz)], capture_output=True, timeout=120)
    assert r.returncode == 1 and b"compressed" in r.stdout
    empty = tmp_path / "empty.txt"
    empty.write_bytes(b"")
    r = subprocess.run([str(exe), str(empty), "65536"], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stdout.decode()
