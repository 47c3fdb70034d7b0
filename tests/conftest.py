import os
import sys

import pytest

# the oracle's OpenMP loops are tiny in the tests; hundreds of host threads only add overhead
os.environ.setdefault("OMP_NUM_THREADS", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _gpu_available():
        pytest.fail("GPU test selected but no HIP device is visible (the product path has no CPU fallback)")
    return True
