"""The threaded host code under ThreadSanitizer and AddressSanitizer (the reference keeps such builds for its pipeline:
Makefile:30-38).  x-search_amd/csrc/xsg_file.cpp (reader -> pinned queue -> device worker -> ordered publication) and
the blocking iterator of include/xsearch/xsearch.h (reference: include/xsearch/ResultTypes.h:48-60) are compiled with
g++ -fsanitize=... against tests/cpp/device_double.cpp, an oracle-backed stand-in for the GPU (test infrastructure;
the product library has no CPU path), and run the concurrent-jobs / early-destroy / live-iteration cases of
tests/cpp/pipeline_sanitize.cpp.  CPU box only: GPU sanitizers are not available on the pool."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
BUILD = ROOT / "tests" / "cpp" / "build"


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", str(ROOT / "tests" / "cpp"), "--no-print-directory", "sanitize"])
    return BUILD


@pytest.mark.parametrize("lanes", ["one lane", "two lanes"])
@pytest.mark.parametrize("name,env", [
    ("pipeline_tsan", {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"}),
    ("pipeline_asan", {"ASAN_OPTIONS": "detect_leaks=1 abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1"}),
])
def test_host_pipeline_under_the_sanitizer(built, tmp_path, name, env, lanes):
    e = dict(os.environ)
    e.update(env)
    # a worker alternates between two lanes (streams) only in jobs long enough to pay for the second one; the test
    # files are small, so the two-lane schedule is forced in one of the two runs; readers: the pipeline's own count
    e["XSG_LANE2_MIN_CHUNKS"] = "1" if lanes == "two lanes" else "1000000"
    e["XSG_READ_PIECE"] = "65536"  # several pieces per chunk: the readers share chunks
    e.pop("XS_DEVICES", None)
    r = subprocess.run([str(built / name), str(tmp_path)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "pipeline under the sanitizer: ok" in r.stdout
