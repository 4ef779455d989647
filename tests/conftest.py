import os
import sys
from pathlib import Path

# the library reads its XSG_* toggles once per process; the route tests switch them between searches (x-search_amd/csrc/xsg_objects.h)
os.environ.setdefault("XSG_TEST_HOOKS", "1")

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle", ROOT / "x-search_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from xs_oracle import Oracle
    o = Oracle()
    o.set_exact(False)
    o.use_reference_primitives(None)
    return o


@pytest.fixture(scope="session")
def reference():
    from xs_oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libxsref.so not built or host lacks AVX2")
    return Reference()


def _ensure_built(target_dir, artefact):
    """C++ test drivers / tools are built in-tree (g++ only); build them on a fresh checkout."""
    import subprocess
    if not artefact.exists():
        subprocess.run(["make", "-C", str(target_dir), "--no-print-directory"], check=False)
    return artefact.exists()


@pytest.fixture(scope="session", autouse=True)
def _cpp_artefacts():
    import xsg
    try:
        xsg.load()  # builds libxsg.so if it is missing
    except FileNotFoundError:
        return
    _ensure_built(ROOT / "tests" / "cpp", ROOT / "tests" / "cpp" / "build" / "extern_search_cli")
    _ensure_built(ROOT / "tests" / "cpp", ROOT / "tests" / "cpp" / "build" / "seam_cli")
    _ensure_built(ROOT / "tools", ROOT / "tools" / "build" / "xsgrep")
