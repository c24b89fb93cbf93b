"""An end-to-end caller through the LD_PRELOAD boundary (tools/lu_preload.py): right-looking blocked LU whose trailing updates are
rocblas_dgemm calls with K = NB under a shrinking trailing matrix - the call stream /root/reference/README.md:17-20 ("LD_PRELOAD an
application") and /root/reference/src/cublas.cu:280-295 (cublasDgemm_v2) exist for.  The child processes run the PRODUCTION
defaults (tuner on, OZIMMU_HIP_* switches read once: the parent's OZIMMU_* environment is stripped).

Asserted: every trailing update is intercepted and runs on the Ozaki path (none failed, none left to the vendor but the ones
below the thresholds), and the factorisation's backward error ||PA - LU|| / ||A|| with fp64_int8_9 is no worse than the vendor
DGEMM's (fp64_int8_3 is visibly worse: the check can tell the paths apart)."""
import pytest

from tools import lu_preload

pytestmark = pytest.mark.gpu

N, NB = 4096, 512


@pytest.fixture(scope="module")
def native():
    return lu_preload.run_mode("native", N, NB, timeout=900)


@pytest.mark.parametrize("mode", ["fp64_int8_9", "fp64_int8_auto"])
def test_blocked_lu_through_the_preload(native, mode):
    r = lu_preload.run_mode(mode, N, NB, timeout=900)
    issued = r["dgemm_calls_issued"]
    assert issued == N // NB - 1
    shim = r["shim"]
    assert shim["failed"] == 0
    assert shim["taken"] == issued, (shim, issued)       # every trailing update ran here (thresholds = NB: all of them qualify)
    assert shim["seen"] == shim["taken"] + shim["declined"]
    assert sum(r["kernels"].values()) >= issued            # ... as slice-GEMM launches of this library
    if mode == "fp64_int8_9":
        # FP64-grade: the Ozaki product's error (8.5e-17 relative at S = 9) is below the DGEMM's own rounding
        assert r["backward_error"] <= 1.25 * native["backward_error"] + 1e-17, (r["backward_error"], native["backward_error"])
    else:
        # threshold 1.5 (tools/lu_preload.py) selects fp64_int8_8 for these trailing matrices: one slice less, 2^7 coarser
        # products, a factorisation ~7 x less accurate than the DGEMM's - what the threshold asks for, and far from 3 slices
        assert r["backward_error"] <= 16 * native["backward_error"], (r["backward_error"], native["backward_error"])
    assert native["backward_error"] < 1e-13


def test_three_slices_are_visibly_coarser():
    r = lu_preload.run_mode("fp64_int8_3", N, NB, timeout=900)
    assert r["shim"]["taken"] == r["dgemm_calls_issued"]
    assert r["backward_error"] > 1e-9
