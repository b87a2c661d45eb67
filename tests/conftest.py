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


def spawn_ranks(fn, args, nprocs, deadline=420.0):
    """`torch.multiprocessing.spawn` with a wall-clock bound: ranks that neither finish nor fail within `deadline`
    seconds are killed by PID and the test fails, instead of the suite waiting on a stuck child for ever."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    t0 = time.time()
    while not ctx.join(timeout=5.0):                                  # raises when a rank failed
        if time.time() - t0 > deadline:
            for proc in ctx.processes:
                if proc.is_alive():
                    proc.kill()
            pytest.fail(f"{fn.__name__}: {nprocs} ranks still running after {deadline:.0f} s (killed)")
