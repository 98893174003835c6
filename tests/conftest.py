import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """Make sure the in-tree HIP library matches the sources before any test imports it (hipcc cross-compiles gfx950 without a
    GPU).  This only (re)builds the product; nothing here substitutes for it - without hipcc the ABI / GPU tests fail loudly."""
    try:
        from q1physrl_amd import build
        if build.is_stale():
            build.build_lib()
    except Exception as ex:   # noqa: BLE001 - reported by the tests that need the library
        sys.stderr.write(f"conftest: libq1env.so not (re)built: {ex}\n")
