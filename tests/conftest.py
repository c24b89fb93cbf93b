import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the parity tests force every kernel / split form through OZIMMU_HIP_* switches that they flip between calls of this one
# process: make the library follow the environment on every use (csrc/config.h; production reads them once per process).
# Subprocess tests (LD_PRELOAD drivers) strip OZIMMU_* and run the production behaviour.
os.environ.setdefault("OZIMMU_HIP_ENV_PER_CALL", "1")
# The measured kernel choice (csrc/kernel_tuner.h) is OFF for the suite: one session-wide handle sees the same shapes again and
# again, and the tests that assert WHICH kernel the cost model picks must not depend on what a timing decides.
# tests/test_gpu_tuner.py switches it on for its own handles; the LD_PRELOAD subprocess tests run with it on (production).
os.environ.setdefault("OZIMMU_HIP_AUTOTUNE", "0")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oz():
    """the product bindings + one library handle on cuda:0 (GPU tests only)"""
    import torch  # noqa: F401  (first: one shared HIP runtime)
    import ozimmu_amd
    h = ozimmu_amd.create()
    yield ozimmu_amd, h
    torch.cuda.synchronize()
    ozimmu_amd.destroy(h)


@pytest.fixture(scope="session")
def ozh():
    """the bindings over libozimmu_hip_test.so (-DOZIMMU_HIP_TEST_HOOKS) + a handle of that library: for the tests that read
    the kernels' INT32 diagonal sums, inject a failed launch or jump the exponent epoch.  Every other test uses `oz`, the
    library that ships."""
    import torch  # noqa: F401
    import ozimmu_amd
    t = ozimmu_amd.test_flavour()
    h = t.create()
    yield t, h
    torch.cuda.synchronize()
    t.destroy(h)
