import sys
from pathlib import Path

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
