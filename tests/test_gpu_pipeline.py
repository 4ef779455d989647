"""The drop-in boundary end to end on the GPU box: xs::extern_search (C++,
include/xsearch/xsearch.h) and its C ABI (xsg_job_*) on real files, plain and
through metafiles (none / LZ4 / ZSTD), against the oracle run chunk by chunk
over the same chunk plan.  Reads like the reference's test/src/xsearchTest.cpp:
{join, live} x {plain, meta} x {six tags} x {1, 4 threads}."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import corpus
import xsg
from gpu_util import oracle_all_modes

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
CLI = ROOT / "tests" / "cpp" / "build" / "extern_search_cli"
TAGS = {"count": xsg.COUNT_MATCHES, "count_lines": xsg.COUNT_LINES, "match_byte_offsets": xsg.MATCH_BYTE_OFFSETS,
        "line_byte_offsets": xsg.LINE_BYTE_OFFSETS, "line_indices": xsg.LINE_INDICES, "lines": xsg.LINES}
KEY = {"count": "count_matches", "count_lines": "count_lines", "match_byte_offsets": "match_byte_offsets",
       "line_byte_offsets": "line_byte_offsets", "line_indices": "line_indices", "lines": "lines"}
CHUNK = 2 << 20


@pytest.fixture(scope="module")
def files(tmp_path_factory, oracle):
    d = tmp_path_factory.mktemp("xsjobs")
    blocks = [corpus.text_block(77, i, 3_000_000 + 17 * i, needle_rate=4e-4) for i in range(6)]
    data = np.concatenate(blocks)
    data = np.concatenate([data[:-1], np.frombuffer(b" SheSherlock", dtype=np.uint8)])  # decoy + no final newline
    txt = d / "sample.txt"
    data.tofile(txt)
    plan = xsg.plan_chunks(str(txt), CHUNK)
    chunks = [data[int(c["original_offset"]):int(c["original_offset"] + c["original_size"])] for c in plan]
    want = {p: oracle_all_modes(oracle, chunks, p) for p in (b"Sherlock", b"She", b"detective street")}
    metas = {}
    for comp, name in ((xsg.COMPRESSION_NONE, "none"), (xsg.COMPRESSION_LZ4, "xslz4"), (xsg.COMPRESSION_ZSTD, "xszst")):
        out = d / f"sample.{name}"
        meta = d / f"sample.{name}.meta"
        xsg.meta_write(str(txt), str(meta), str(out), comp, CHUNK, 500)
        metas[name] = (str(txt) if comp == xsg.COMPRESSION_NONE else str(out), str(meta))
    return {"txt": str(txt), "want": want, "metas": metas, "nchunks": len(plan), "size": data.size}


def as_py(tag, value):
    if tag in ("count", "count_lines"):
        return int(value)
    if tag == "lines":
        return list(value)
    return [int(x) for x in value]


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("tag", list(TAGS))
def test_job_join_plain(files, tag, threads):
    for pat, want in files["want"].items():
        j = xsg.Job(pat, files["txt"], TAGS[tag], num_threads=threads, num_max_readers=threads, chunk_bytes=CHUNK)
        got = as_py(tag, j.result())
        assert got == want[KEY[tag]], (tag, pat, threads)
        st = j.stats()
        assert st["chunks"] == files["nchunks"] and st["bytes_scanned"] == files["size"]
        j.close()


@pytest.mark.parametrize("tag", list(TAGS))
def test_job_live_iteration(files, tag):
    pat = b"She"
    want = files["want"][pat][KEY[tag]]
    j = xsg.Job(pat, files["txt"], TAGS[tag], num_threads=3, num_max_readers=2, chunk_bytes=CHUNK)
    seen = list(j)  # blocks per element until the job closes
    if tag in ("count", "count_lines"):
        assert len(seen) == files["nchunks"] and seen[-1] == want and seen == sorted(seen)
    else:
        assert as_py(tag, seen) == want
    j.join()
    j.close()


@pytest.mark.parametrize("meta", ["none", "xslz4", "xszst"])
@pytest.mark.parametrize("tag", list(TAGS))
def test_job_with_metafile(files, tag, meta):
    data_path, meta_path = files["metas"][meta]
    pat = b"Sherlock"
    j = xsg.Job(pat, data_path, TAGS[tag], meta_path=meta_path, num_threads=2, num_max_readers=2)
    assert as_py(tag, j.result()) == files["want"][pat][KEY[tag]], (tag, meta)
    st = j.stats()
    if meta != "none":
        assert st["bytes_read"] < st["bytes_scanned"] == files["size"]
    j.close()


def test_job_errors(files, tmp_path):
    with pytest.raises(xsg.XsgError) as e:
        xsg.Job(b"x", str(tmp_path / "missing.txt"), xsg.COUNT_MATCHES)
    assert e.value.code == xsg.EIO
    with pytest.raises(xsg.XsgError) as e:
        xsg.Job(b"", files["txt"], xsg.COUNT_MATCHES)
    assert e.value.code == xsg.EINVAL
    bad = tmp_path / "bad.meta"
    bad.write_bytes(b"\x09\x00\x00\x00")
    with pytest.raises(xsg.XsgError) as e:
        xsg.Job(b"x", files["txt"], xsg.COUNT_MATCHES, meta_path=str(bad))
    assert e.value.code == xsg.EIO
    with pytest.raises(xsg.XsgError) as e:  # an EXPRESSION that can match '\n': the line tags refuse (a literal is served)
        xsg.Job(b"a[^x]b", files["txt"], xsg.LINES, flags=xsg.FLAG_REGEX)
    assert e.value.code == xsg.ENOTSUP
    empty = tmp_path / "empty.txt"
    empty.write_bytes(b"")
    j = xsg.Job(b"x", str(empty), xsg.COUNT_MATCHES)
    assert j.result() == 0 and list(j) == []
    j.close()


def test_job_with_a_pattern_that_contains_a_newline(files, oracle):
    """every tag through the file pipeline for a literal with '\\n' in it (the reference's walk takes any string)"""
    data = np.fromfile(files["txt"], dtype=np.uint8)
    plan = xsg.plan_chunks(files["txt"], CHUNK)
    chunks = [data[int(c["original_offset"]):int(c["original_offset"] + c["original_size"])] for c in plan]
    for pat in (b"e\nthe", b"\nShe"):
        want = oracle_all_modes(oracle, chunks, pat)
        assert want["count_lines"] > 0
        for tag in TAGS:
            j = xsg.Job(pat, files["txt"], TAGS[tag], num_threads=2, num_max_readers=2, chunk_bytes=CHUNK)
            assert as_py(tag, j.result()) == want[KEY[tag]], (pat, tag)
            j.close()


def run_cli(*args, env=None):
    import os
    e = dict(os.environ)
    e["XS_CHUNK_BYTES"] = str(CHUNK)
    if env:
        e.update(env)
    return subprocess.run([str(CLI), *args], capture_output=True, env=e, timeout=300)


@pytest.mark.parametrize("how", ["join", "live"])
@pytest.mark.parametrize("tag", list(TAGS))
def test_cpp_extern_search(files, tag, how):
    """xs::extern_search<Tag>(pattern, file, false, threads) as README.md:37 / xsearchTest.cpp:344 call it."""
    if not CLI.exists():
        pytest.fail(f"{CLI} not built (make -C tests/cpp)")
    pat = b"Sherlock"
    want = files["want"][pat][KEY[tag]]
    for threads in ("1", "4"):
        r = run_cli(tag, how, pat.decode(), files["txt"], "-", threads)
        assert r.returncode == 0, r.stderr.decode()
        out = r.stdout.split(b"\n")[:-1]
        if tag in ("count", "count_lines"):
            assert int(out[0]) == want
        elif tag == "lines":
            assert out == want
        else:
            assert [int(x) for x in out] == want


def test_cpp_extern_search_with_meta_and_errors(files):
    data_path, meta_path = files["metas"]["xslz4"]
    r = run_cli("count", "join", "Sherlock", data_path, meta_path, "4", "2")  # README.md:72 / checkit.cpp:4
    assert r.returncode == 0 and int(r.stdout) == files["want"][b"Sherlock"]["count_matches"]
    r = run_cli("lines", "live", "Sherlock", data_path, meta_path, "4", "2")
    assert r.returncode == 0 and r.stdout.split(b"\n")[:-1] == files["want"][b"Sherlock"]["lines"]
    r = run_cli("count", "join", "Sherlock", "/nonexistent/file.txt")
    assert r.returncode == 1 and b"cannot open" in r.stderr
    # the same LZ4 corpus through the built-in block decoder (what a host without liblz4 uses)
    r = run_cli("match_byte_offsets", "join", "Sherlock", data_path, meta_path, "4", "4", env={"XSG_NO_LIBLZ4": "1"})
    assert r.returncode == 0 and [int(x) for x in r.stdout.split()] == files["want"][b"Sherlock"]["match_byte_offsets"]


def test_config1_shape_100mb_six_chunks(oracle, tmp_path):
    """BASELINE configs[0]: xs::count 'Sherlock' on a 100 000 000-byte file -> 6 chunks of 16 MiB(+) like
    test/files/sample.meta."""
    blocks = [corpus.text_block(123, i, 10_000_000) for i in range(10)]
    data = np.concatenate(blocks)
    assert data.size == 100_000_000
    p = tmp_path / "sample.txt"
    data.tofile(p)
    plan = xsg.plan_chunks(str(p))
    assert len(plan) == 6 and all(16777216 <= int(c["original_size"]) < 16777216 + 200 for c in plan[:-1])
    chunks = [data[int(c["original_offset"]):int(c["original_offset"] + c["original_size"])] for c in plan]
    want = sum(oracle.count(c, b"Sherlock", False) for c in chunks)
    j = xsg.Job(b"Sherlock", str(p), xsg.COUNT_MATCHES, num_threads=1)
    assert j.result() == want > 0
    j.close()


SEAM = ROOT / "tests" / "cpp" / "build" / "seam_cli"


@pytest.mark.parametrize("threads", ["1", "4"])
def test_reference_style_searcher_functors(files, oracle, threads):
    """GpuIndexSearcher / GpuLineIndexSearcher / GpuLineSearcher called like
    Searcher::run_thread calls the reference's functors (Searcher.h:100-120,
    tasks/searchers.h:38-93): chunk-local results, nullopt for hit-less chunks,
    one shared const functor used by N threads."""
    if not SEAM.exists():
        pytest.fail(f"{SEAM} not built (make -C tests/cpp)")
    data = np.fromfile(files["txt"], dtype=np.uint8)
    plan = xsg.plan_chunks(files["txt"], CHUNK)
    chunks = [data[int(c["original_offset"]):int(c["original_offset"] + c["original_size"])] for c in plan]
    pat = b"She"

    def parse(out):
        res, cur = {}, None
        for line in out.split(b"\n")[:-1]:
            if line.startswith(b"C "):
                _, i, n = line.split()
                cur = res.setdefault(int(i), [])
            else:
                cur.append(line)
        return res

    for what, fn in (("index", lambda b: [str(int(x)).encode() for x in oracle.byte_offsets_match(b, pat)]),
                     ("line_index", lambda b: [str(int(x)).encode() for x in oracle.byte_offsets_line(b, pat)]),
                     ("line", lambda b: oracle.lines(b, pat))):
        r = subprocess.run([str(SEAM), what, pat.decode(), files["txt"], str(CHUNK), threads], capture_output=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr.decode()
        got = parse(r.stdout)
        assert sorted(got) == list(range(len(chunks)))
        for i, b in enumerate(chunks):
            assert got[i] == fn(b), (what, i)
    r = subprocess.run([str(SEAM), "count", pat.decode(), files["txt"], str(CHUNK), threads], capture_output=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    counts = {int(l.split()[1]): int(l.split()[2]) for l in r.stdout.split(b"\n")[:-1]}
    assert [counts[i] for i in range(len(chunks))] == [oracle.count(b, pat, True) for b in chunks]


def test_cpp_extern_search_ignore_case(files, oracle):
    """xs::extern_search<Tag>(pattern, file, /*ignore_case=*/true, threads) -- test/src/xsearchTest.cpp:350"""
    data = np.fromfile(files["txt"], dtype=np.uint8)
    plan = xsg.plan_chunks(files["txt"], CHUNK)
    chunks = [data[int(c["original_offset"]):int(c["original_offset"] + c["original_size"])] for c in plan]
    want = oracle_all_modes(oracle, chunks, b"sHERLOCK", ignore_case=True)
    assert want["count_matches"] >= files["want"][b"Sherlock"]["count_matches"] > 0
    r = run_cli("count", "join", "sHERLOCK", files["txt"], "-", "2", env={"XS_IGNORE_CASE": "1"})
    assert r.returncode == 0 and int(r.stdout) == want["count_matches"]
    r = run_cli("lines", "live", "sHERLOCK", files["txt"], "-", "2", env={"XS_IGNORE_CASE": "1"})
    assert r.returncode == 0 and r.stdout.split(b"\n")[:-1] == want["lines"]
    r = run_cli("count", "join", "sHERLOCK", files["txt"], "-", "2")
    assert r.returncode == 0 and int(r.stdout) == 0


def test_xsgrep_equals_gnu_grep(tmp_path):
    """tools/xsgrep (the reference's example/grep.cpp on this engine) against GNU grep
    on a clean, newline-terminated file: same lines, same -c count, same -i behaviour."""
    import shutil
    exe = ROOT / "tools" / "build" / "xsgrep"
    if not exe.exists():
        pytest.fail(f"{exe} not built (make -C tools)")
    if not shutil.which("grep"):
        pytest.skip("no GNU grep on this host")
    data = np.concatenate([corpus.text_block(404, i, 2_500_000, needle_rate=3e-4) for i in range(4)])
    # pad the end so that no needle sits in the reference's lossy end-of-chunk zone of the LAST chunk
    data = np.concatenate([data, np.frombuffer(b"the end of the file is plain text without the needle\n" * 2, dtype=np.uint8)])
    p = tmp_path / "g.txt"
    data.tofile(p)
    import os
    env = dict(os.environ, XS_CHUNK_BYTES=str(1 << 30), LC_ALL="C")  # one chunk: grep has no chunk-end quirk
    for args in (["Sherlock"], ["-i", "sherlock"], ["-i", "HOLMES"], ["locked"]):
        want = subprocess.run(["grep", "-F", *args, str(p)], capture_output=True, env=env).stdout
        got = subprocess.run([str(exe), "-j", "2", *args, str(p)], capture_output=True, env=env, timeout=120)
        assert got.returncode == 0, got.stderr.decode()
        assert got.stdout == want, args
        wc = subprocess.run(["grep", "-F", "-c", *args, str(p)], capture_output=True, env=env).stdout
        gc = subprocess.run([str(exe), "-c", *args, str(p)], capture_output=True, env=env, timeout=120).stdout
        assert gc == wc, args
    # regular expressions of the class-sequence family: grep reads them as basic regexes the same way
    # ... and of the variable-length family (the automaton route), which grep -E reads the same way
    for args in (["She[r ]lock"], ["-i", "she[r ]lock"], ["[Hh]olmes[ ,.]"], ["^She"], ["Sher.*[lm]es"], ["lock(ed|s)? "],
                 ["-i", "holmes +[a-z]+ed"], ["[A-Z][a-z]+ [A-Z][a-z]+"]):
        got = subprocess.run([str(exe), "-j", "2", *args, str(p)], capture_output=True, env=env, timeout=120)
        if args == ["^She"]:  # an anchor: refused, not searched as text
            assert got.returncode == 1 and b"does not serve" in got.stderr
            continue
        want = subprocess.run(["grep", "-E", *args, str(p)], capture_output=True, env=env).stdout
        assert got.returncode == 0, got.stderr.decode()
        assert got.stdout == want and len(want) > 0, args
        wc = subprocess.run(["grep", "-E", "-c", *args, str(p)], capture_output=True, env=env).stdout
        gc = subprocess.run([str(exe), "-c", *args, str(p)], capture_output=True, env=env, timeout=120).stdout
        assert gc == wc, args


def _dist_gpu_worker(rank, world, port, path, chunk_bytes, q):
    import os
    import sys
    for p in (ROOT / "x-search_amd", ROOT / "oracle", ROOT / "tests"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import dist_search
    import xsg as x
    dist.init_process_group("gloo", rank=rank, world_size=world)  # one GPU on this box: collectives via gloo
    out = {}
    for mode in (x.COUNT_MATCHES, x.COUNT_LINES, x.LINE_INDICES, x.MATCH_BYTE_OFFSETS, x.LINES):
        r = dist_search.distributed_search(b"Sherlock", path, mode, dist=dist, device=0, num_threads=2,
                                           chunk_bytes=chunk_bytes)
        out[mode] = r if isinstance(r, int) else [v if isinstance(v, bytes) else int(v) for v in r]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_the_file_gpu_scanner(files):
    """dist_search with the real per-rank scanner (xsg.Job on a chunk range): two
    processes on this box's one GPU, contiguous chunk ranges, all_reduce / all_gather
    through gloo (RCCL needs one GPU per rank)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_gpu_worker, args=(r, 2, port, files["txt"], CHUNK, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = files["want"][b"Sherlock"]
    assert got[0][xsg.COUNT_MATCHES] == got[1][xsg.COUNT_MATCHES] == want["count_matches"]
    assert got[0][xsg.COUNT_LINES] == got[1][xsg.COUNT_LINES] == want["count_lines"]
    assert got[0][xsg.MATCH_BYTE_OFFSETS] + got[1][xsg.MATCH_BYTE_OFFSETS] == want["match_byte_offsets"]
    assert got[0][xsg.LINE_INDICES] + got[1][xsg.LINE_INDICES] == want["line_indices"]
    assert got[0][xsg.LINES] + got[1][xsg.LINES] == want["lines"]


def test_xsgrep_stdin(tmp_path):
    """xsgrep PATTERN - : chunks cut on the fly from stdin, searched through the functor seam."""
    import os
    exe = ROOT / "tools" / "build" / "xsgrep"
    data = np.concatenate([corpus.text_block(515, i, 9_000_000, needle_rate=3e-4) for i in range(4)])  # > 2 chunks of 16 MiB
    data = np.concatenate([data, np.frombuffer(b"plain closing line\n" * 4, dtype=np.uint8)])
    p = tmp_path / "s.txt"
    data.tofile(p)
    env = dict(os.environ, LC_ALL="C")
    for args in (["Sherlock"], ["-i", "holmes"]):
        want = subprocess.run(["grep", "-F", *args, str(p)], capture_output=True, env=env).stdout
        with open(p, "rb") as f:
            got = subprocess.run([str(exe), *args, "-"], stdin=f, capture_output=True, env=env, timeout=120)
        assert got.returncode == 0, got.stderr.decode()
        # grep has no end-of-chunk quirk; the chunks here end on line boundaries and the needle never sits in
        # the last 40 bytes of a 16 MiB chunk in this corpus (checked by the equality itself)
        assert got.stdout == want, args
        wc = subprocess.run(["grep", "-F", "-c", *args, str(p)], capture_output=True, env=env).stdout
        with open(p, "rb") as f:
            gc = subprocess.run([str(exe), "-c", *args, "-"], stdin=f, capture_output=True, env=env, timeout=120).stdout
        assert gc == wc, args


def test_concurrent_jobs_and_early_destroy(files):
    """Several searches at once in one process (each with its own workers, slots from the
    shared pool), a job destroyed while it is still running, and many short jobs in a row."""
    import threading
    want = files["want"][b"Sherlock"]
    results, errors = {}, []

    def run(i, tag):
        try:
            j = xsg.Job(b"Sherlock", files["txt"], TAGS[tag], num_threads=1 + i % 3, num_max_readers=2, chunk_bytes=CHUNK)
            results[(i, tag)] = as_py(tag, j.result())
            j.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=run, args=(i, tag)) for i, tag in enumerate(list(TAGS) * 2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    for (i, tag), got in results.items():
        assert got == want[KEY[tag]], (i, tag)
    # destroy without join, while workers are busy: must stop and free cleanly
    for _ in range(5):
        j = xsg.Job(b"e", files["txt"], xsg.LINES, num_threads=4, num_max_readers=4, chunk_bytes=CHUNK)
        j.close()
    # many short jobs reuse pooled slots
    for _ in range(40):
        j = xsg.Job(b"Sherlock", files["txt"], xsg.COUNT_MATCHES, num_threads=2, chunk_bytes=CHUNK)
        assert j.result() == want["count_matches"]
        j.close()


def test_regex_routing_of_extern_search(files, oracle):
    """'She[r ]lock' is a regex for the reference (utils/utils.h:17-25; 53 matches in its goldens,
    xsearchTest.cpp:19): xs::extern_search serves it through the kernel's class-sequence matcher, for every
    tag, with and without ignore_case; expressions of variable length through the automaton route; a regex outside
    both families is refused loudly, never searched as text."""
    from gpu_util import oracle_regex_all_modes
    data = np.fromfile(files["txt"], dtype=np.uint8)
    plan = xsg.plan_chunks(files["txt"], CHUNK)
    chunks = [data[int(c["original_offset"]):int(c["original_offset"] + c["original_size"])] for c in plan]
    for icase in (False, True):
        want, _ = oracle_regex_all_modes(oracle, chunks, b"She[r ]lock", icase)
        lit = oracle_all_modes(oracle, chunks, b"Sherlock", ignore_case=icase)
        assert want["count_matches"] >= lit["count_matches"] > 0
        for tag in TAGS:
            r = run_cli(tag, "join", "She[r ]lock", files["txt"], env={"XS_IGNORE_CASE": "1"} if icase else None)
            assert r.returncode == 0, r.stderr
            if tag in ("count", "count_lines"):
                assert int(r.stdout) == want[KEY[tag]], (tag, icase)
            elif tag == "lines":
                assert r.stdout.split(b"\n")[:-1] == want["lines"], icase
            else:
                assert [int(x) for x in r.stdout.split()] == want[KEY[tag]], (tag, icase)
    r = run_cli("count", "join", "She[r ]lock", files["txt"], env={"XS_FORCE_LITERAL": "1"})
    assert r.returncode == 0 and int(r.stdout) == 0
    # expressions of variable length take the automaton route (k_rx_scan), every tag
    for expr in ("Sherlock|Holmes", "Sher?lock", "Sher.*?k +[a-z]+"):
        want, _ = oracle_regex_all_modes(oracle, chunks, expr.encode(), False)
        assert want["count_matches"] > 0
        for tag in TAGS:
            r = run_cli(tag, "join", expr, files["txt"])
            assert r.returncode == 0, r.stderr
            if tag in ("count", "count_lines"):
                assert int(r.stdout) == want[KEY[tag]], (expr, tag)
            elif tag == "lines":
                assert r.stdout.split(b"\n")[:-1] == want["lines"], expr
            else:
                assert [int(x) for x in r.stdout.split()] == want[KEY[tag]], (expr, tag)
    # a set that accepts '\n' under a repetition: matches may span lines -- the match tags are served, the line tags refuse
    want, with_lines = oracle_regex_all_modes(oracle, chunks, b"She[^r]+lock", False)
    assert not with_lines
    r = run_cli("count", "join", "She[^r]+lock", files["txt"])
    assert r.returncode == 0 and int(r.stdout) == want["count_matches"], r.stderr
    r = run_cli("count_lines", "join", "She[^r]+lock", files["txt"])
    assert r.returncode == 1 and b"\\n" in r.stderr, r.stderr
    for expr in ("^Sherlock", "Sher\\b", "(Sher)*"):
        r = run_cli("count", "join", expr, files["txt"])
        assert r.returncode == 1 and b"regular expression" in r.stderr and b"does not serve" in r.stderr, (expr, r.stderr)
    for expr in ("a.b", "She.*lock"):  # these match themselves as regexes -> plain text for the reference too
        r = run_cli("count", "join", expr, files["txt"])
        assert r.returncode == 0 and int(r.stdout) == 0


@pytest.mark.parametrize("how", ["join", "live"])
def test_extern_search_fans_out_over_devices(files, how):
    """XS_DEVICES: one xs::extern_search call runs one job per listed device on contiguous chunk ranges
    and merges on the host -- counts add up, a live count ends on the global total, list tags come back in file
    order with global offsets, line indices carry the newlines of the earlier ranges.  One GPU here, so the same
    device is listed two and three times: the merge logic does not care."""
    want = files["want"][b"Sherlock"]
    data_path, meta_path = files["metas"]["xslz4"]
    for devs in ("0,0", "0,0,0"):
        for tag in TAGS:
            for src, meta in ((files["txt"], "-"), (data_path, meta_path)):
                r = run_cli(tag, how, "Sherlock", src, meta, "2", "2", env={"XS_DEVICES": devs})
                assert r.returncode == 0, r.stderr
                if tag in ("count", "count_lines"):
                    assert int(r.stdout) == want[KEY[tag]], (tag, devs, meta)
                elif tag == "lines":
                    assert r.stdout.split(b"\n")[:-1] == want["lines"], (devs, meta)
                else:
                    assert [int(x) for x in r.stdout.split()] == want[KEY[tag]], (tag, devs, meta)
    r = run_cli("count", "join", "Sherlock", files["txt"], env={"XS_DEVICES": "0,x"})
    assert r.returncode == 1 and b"XS_DEVICES" in r.stderr
    r = run_cli("count", "join", "Sherlock", files["txt"], env={"XS_DEVICES": "0,99"})
    assert r.returncode == 1 and b"device" in r.stderr


@pytest.mark.parametrize("threads,readers", [(1, 1), (2, 6), (4, 8)])
@pytest.mark.parametrize("tag", list(TAGS))
def test_job_over_a_chunk_range(files, oracle, tag, threads, readers):
    """One rank's share of a file (xsg_job_opts.chunk_begin/_end; bench.py --gpus N, xs::extern_search with XS_DEVICES):
    byte offsets stay global, line indices count from the range's first line."""
    data = np.fromfile(files["txt"], dtype=np.uint8)
    plan = xsg.plan_chunks(files["txt"], CHUNK)
    chunks = [data[int(c["original_offset"]):int(c["original_offset"] + c["original_size"])] for c in plan]
    n = len(chunks)
    goff = [int(c["original_offset"]) for c in plan]
    for lo, hi in ((0, n), (1, n), (n // 2, n), (2, 4), (n - 1, n)):
        for pat in (b"Sherlock", b"She"):
            want = oracle_all_modes(oracle, chunks[lo:hi], pat, global_offsets=goff[lo:hi])
            j = xsg.Job(pat, files["txt"], TAGS[tag], num_threads=threads, num_max_readers=readers, chunk_bytes=CHUNK,
                        chunk_range=(lo, hi))
            got = as_py(tag, j.result())
            j.close()
            assert got == want[KEY[tag]], (tag, pat, lo, hi, threads, readers)
