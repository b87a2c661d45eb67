import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _fresh_hip_library():
    """(Re)build liblego_hip.so before any test: a stale .so behind a changed header is the one
    failure mode ctypes cannot detect.  `make` is a no-op when everything is up to date."""
    from legommenders_amd import _lib
    _lib.build()
