"""GPU parity of every kernel the launch policy can pick, each FORCED and its identity asserted.

tests/test_gpu_wide_kernel.py / test_gpu_k2_kernel.py force a kernel FAMILY (OZIMMU_HIP_GEMM_KERNEL) and let the cost model
choose inside it: whether the k64 tile ran with its B fragments through LDS or in named registers (the kernel behind the headline
number), whether a persistent launch drew its tickets one tile ahead, whether a few exact rounds of tiles ran as a static grid -
all of that depended on what the model happened to predict for the test's shape, and a refit could silently drop the coverage.
Here every such form is an explicit arm (FORCED_ARMS: switch values -> the name ozimmu_hip_last_kernel must report), checked
bit for bit against the oracle (OZ_ORDER_DIAGONAL: the kernels' summation grouping): INT32 diagonal sums and the FP64 result.
tests/test_abi.py::test_every_pick_has_a_forced_gpu_arm (CPU) fails when a Pick value has no arm here.

The reference's own pin for this path is the residual gate of test/main_test.cu:702-746 (< 1e-15 for fp64_int8_8..16); every arm
is under it as well (test_forced_kernel_passes_the_reference_gate)."""
import functools

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import ColMajor, exp_rand, operand, uniform01, uniform_pm1

pytestmark = pytest.mark.gpu

# name reported by ozimmu_hip_last_kernel -> the switches that force it (on a shape / mode where the kernel exists)
FORCED_ARMS = {
    "k2": {"OZIMMU_HIP_GEMM_KERNEL": "k2", "OZIMMU_HIP_ONE_LAUNCH": "0"},
    # split + K-split tile in ONE kernel (csrc/slice_gemm_one_launch.hip): what a small real GEMM (K <= 2048, S <= 9) runs by default
    "k2_one_launch": {"OZIMMU_HIP_GEMM_KERNEL": "k2", "OZIMMU_HIP_ONE_LAUNCH": "1"},
    "classic": {"OZIMMU_HIP_GEMM_KERNEL": "classic"},
    "wide": {"OZIMMU_HIP_GEMM_KERNEL": "wide", "OZIMMU_HIP_PAIRED_TILE": "0"},
    "x16": {"OZIMMU_HIP_GEMM_KERNEL": "x16"},
    "k64": {"OZIMMU_HIP_GEMM_KERNEL": "k64", "OZIMMU_HIP_K64_BREG": "0"},
    "k64_breg": {"OZIMMU_HIP_GEMM_KERNEL": "k64", "OZIMMU_HIP_K64_BREG": "1"},
}
# a single-pass mode every arm is built for (k64_breg: 9 staged diagonals; x16 at S = 9 is not instantiated: S = 12)
# (k64 at S = 11, 12: the round-5 form with named accumulator registers and the highest B slices refilled in place)
ARM_MODES = {"k2": [4, 9], "k2_one_launch": [3, 4, 7, 9], "classic": [3, 9, 12], "wide": [4, 9, 11], "x16": [12, 10, 11], "k64": [4, 9, 10, 11, 12], "k64_breg": [9]}


def _sync():
    import torch
    torch.cuda.synchronize()


def _force(monkeypatch, arm, **extra):
    for k in ("OZIMMU_HIP_GEMM_KERNEL", "OZIMMU_HIP_PAIRED_TILE", "OZIMMU_HIP_K64_BREG", "OZIMMU_HIP_K64_TILE", "OZIMMU_HIP_ONE_LAUNCH"):
        monkeypatch.delenv(k, raising=False)
    for k, v in {**FORCED_ARMS[arm], **extra}.items():
        monkeypatch.setenv(k, str(v))


def _arm_cases():
    return [(arm, S) for arm, modes in ARM_MODES.items() for S in modes]


# k-block counts = 0 mod 4 (the register form's loop body holds two 64-k steps): 128 = 4, 256 = 8, 1024 = 32, 1120 -> 36
SHAPES = [(200, 130, 128), (389, 257, 256), (130, 200, 1024), (70, 129, 1120)]


@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("arm,S", _arm_cases())
def test_forced_kernel_diagonal_sums_bit_exact(ozh, monkeypatch, arm, S, m, n, k):
    import torch
    m_, h = ozh   # the INT32 dump is a test hook: libozimmu_hip_test.so
    _force(monkeypatch, arm)
    rng = np.random.default_rng(m * 5 + n * 3 + k + S)
    a = operand("N", m, k, rng, fill=exp_rand(2.0))
    b = operand("T", k, n, rng, fill=exp_rand(2.0))
    L = O.bits_per_int8(k)
    pa, _ = O.split("A", "N", a.view, S, L)
    pb, _ = O.split("B", "T", b.view, S, L)
    d_ref = O.diagonal_sums(pa, pb)
    out = torch.full((S, n, m), 12345, dtype=torch.int32, device="cuda")
    assert m_.diagonal_sums(h, "N", "T", m, n, k, a.dev, a.ld, b.dev, b.ld, S, out) == 0
    _sync()
    assert m_.last_kernel(h)[0] == arm
    np.testing.assert_array_equal(out.cpu().numpy().transpose(0, 2, 1).astype(np.int64), d_ref)


@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "T"), ("N", "T")])
@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("arm,S", _arm_cases())
def test_forced_kernel_gemm_bit_exact_vs_oracle(oz, monkeypatch, arm, S, m, n, k, op_a, op_b):
    m_, h = oz    # the library that ships
    _force(monkeypatch, arm)
    rng = np.random.default_rng(m + 2 * n + 3 * k + S)
    a = operand(op_a, m, k, rng, pad=1)
    b = operand(op_b, k, n, rng, pad=2)
    c = ColMajor(m, n, ld=m + 3, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n, ld=m + 3)
    c_ref.buf[...] = c.buf
    st = m_.gemm(h, op_a, op_b, m, n, k, -0.75, a.dev, a.ld, b.dev, b.ld, 1.25, c.dev, c.ld, f"fp64_int8_{S}")
    _sync()
    assert st == 0
    assert m_.last_kernel(h)[0] == arm
    assert O.gemm(op_a, op_b, m, n, k, -0.75, a.view, b.view, 1.25, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))
    assert np.isnan(c.buf[:, m:]).all()  # ld padding untouched


@pytest.mark.parametrize("arm", sorted(FORCED_ARMS))
def test_forced_kernel_passes_the_reference_gate(oz, monkeypatch, arm):
    """test/main_test.cu:702-746: uniform (0, 1], m = n = k in {1023, 1024, 1025} -> here 1024 (k-blocks = 0 mod 4), residual
    < 1e-15 at fp64_int8_9 (fp64_int8_12 for the paired tile, which has no S = 9 instantiation)"""
    m_, h = oz
    S = 12 if arm == "x16" else 9
    _force(monkeypatch, arm)
    n = 1024
    rng = np.random.default_rng(99)
    a = operand("N", n, n, rng, fill=uniform01)
    b = operand("N", n, n, rng, fill=uniform01)
    c = ColMajor(n, n)
    assert m_.gemm(h, "N", "N", n, n, n, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld, f"fp64_int8_{S}") == 0
    _sync()
    assert m_.last_kernel(h)[0] == arm
    r = O.relative_residual_sampled("N", "N", n, n, n, a.view, b.view, c.download(), ns=512)
    assert r < 1e-15, r


# ---- persistent workgroups: claims, stealing, the ticket drawn one tile ahead, the static grid -------------------------------
# Shapes of more tiles than CUs (m * n >= 2.5 M, K >= 1024: the calls that get phase lines and claim counters), oracle
# computed once per shape and compared with every launch form.
@functools.lru_cache(maxsize=None)
def _big_case(m, n, k, S):
    rng = np.random.default_rng(m + n + k + S)
    a = operand("N", m, k, rng)
    b = operand("N", k, n, rng)
    c0 = ColMajor(m, n, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n)
    c_ref.buf[...] = c0.buf
    assert O.gemm("N", "N", m, n, k, 1.5, a.view, b.view, -0.5, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    return a, b, c0, c_ref


# 2048 x 1536: 384 tiles of 64 x 128 (1.5 rounds: queue + stealing); 2048 x 2048: 512 tiles = 2 exact rounds (static grid by
# default); K = 1024 and 2048 = 32 / 64 k-blocks (both within OZIMMU_HIP_SPEC_CLAIM_KB's default of 64)
@pytest.mark.parametrize("spec_kb", [0, 64])
@pytest.mark.parametrize("static_rounds", [0, 20])
@pytest.mark.parametrize("grid", [0, 7])
@pytest.mark.parametrize("breg", [0, 1])
@pytest.mark.parametrize("m,n,k", [(2048, 1536, 1024), (2048, 2048, 1024), (1600, 1664, 2048)])
def test_persistent_launch_forms_bit_exact_vs_oracle(oz, monkeypatch, m, n, k, breg, grid, static_rounds, spec_kb):
    m_, h = oz
    S = 9
    a, b, c0, c_ref = _big_case(m, n, k, S)
    _force(monkeypatch, "k64_breg" if breg else "k64", OZIMMU_HIP_WIDE_GRID=grid, OZIMMU_HIP_STATIC_ROUNDS=static_rounds,
           OZIMMU_HIP_SPEC_CLAIM_KB=spec_kb)
    c = ColMajor(m, n)
    c.buf[...] = c0.buf
    c._dev = None
    assert m_.gemm(h, "N", "N", m, n, k, 1.5, a.dev, a.ld, b.dev, b.ld, -0.5, c.dev, c.ld, f"fp64_int8_{S}") == 0
    _sync()
    assert m_.last_kernel(h)[0] == ("k64_breg" if breg else "k64")
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


@pytest.mark.parametrize("grid", [1, 3, 7])
@pytest.mark.parametrize("arm,S", [("wide", 9), ("wide", 6), ("x16", 12), ("k64", 9), ("k64_breg", 9), ("k64", 6), ("k64", 11), ("k64", 12)])
@pytest.mark.parametrize("m,n,k", [(700, 520, 128), (333, 900, 256)])
def test_few_persistent_workgroups_walk_many_tiles(oz, monkeypatch, arm, S, m, n, k, grid):
    """OZIMMU_HIP_WIDE_GRID = g on a small problem: the call gets phase lines and claim counters (api.cpp: wants_phase) and g
    persistent workgroups walk all its tiles of both heights - claims, stealing across the XCD runs, the speculative ticket
    (K <= 2048) and the re-use of LDS and registers from tile to tile, on shapes the oracle checks in a second"""
    m_, h = oz
    _force(monkeypatch, arm, OZIMMU_HIP_WIDE_GRID=grid)
    rng = np.random.default_rng(m + n + k + S + grid)
    a = operand("T", m, k, rng, pad=1)
    b = operand("N", k, n, rng)
    c = ColMajor(m, n, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n)
    c_ref.buf[...] = c.buf
    assert m_.gemm(h, "T", "N", m, n, k, 1.5, a.dev, a.ld, b.dev, b.ld, -0.5, c.dev, c.ld, f"fp64_int8_{S}") == 0
    _sync()
    assert m_.last_kernel(h)[0] == arm
    assert O.gemm("T", "N", m, n, k, 1.5, a.view, b.view, -0.5, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


# ---- the register kernel's store forms inside one launch -----------------------------------------------------------------------
# Interior tiles of a real, final product with an even ldc and a 16-byte aligned C store 16 bytes per lane through the lane-pair
# exchange; edge tiles, odd ldc and misaligned C take the 8-byte form - every case below mixes both inside one launch, with
# several tiles per workgroup where a grid is given.  (Written for the round-5 attempt to recombine under the last k-step's MFMAs,
# profiles/r5_ablate/r5c_*: the attempt is out, the cases stay.)
@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (-0.75, 1.25)])
@pytest.mark.parametrize("m,n,k,ld_pad,grid", [(389, 257, 256, 3, 0), (389, 257, 256, 2, 0), (200, 300, 128, 0, 0), (640, 512, 512, 0, 0),
                                               (700, 520, 1024, 4, 3), (96, 256, 2048, 0, 0), (2048, 1536, 256, 0, 0)])
def test_register_kernel_store_forms_bit_exact_vs_oracle(oz, monkeypatch, m, n, k, ld_pad, grid, alpha, beta):
    m_, h = oz
    S = 9
    extra = {}
    if grid:
        extra["OZIMMU_HIP_WIDE_GRID"] = grid
    _force(monkeypatch, "k64_breg", **extra)
    rng = np.random.default_rng(m + 3 * n + 5 * k + ld_pad)
    a = operand("N", m, k, rng)
    b = operand("T", k, n, rng, pad=1)
    c = ColMajor(m, n, ld=m + ld_pad, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n, ld=m + ld_pad)
    c_ref.buf[...] = c.buf
    assert m_.gemm(h, "N", "T", m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c.dev, c.ld, f"fp64_int8_{S}") == 0
    _sync()
    assert m_.last_kernel(h)[0] == "k64_breg"
    assert O.gemm("N", "T", m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))
    if ld_pad:
        assert np.isnan(c.buf[:, m:]).all()  # ld padding untouched
