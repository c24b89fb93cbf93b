"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (task ③): bit-exact for the integer stages (slices, max_exp, INT32 diagonal sums) and for the FP64
result against the oracle evaluated in the kernel's own summation grouping (OZ_ORDER_DIAGONAL); against
the reference's pair-by-pair grouping (OZ_ORDER_REFERENCE) the FP64 result may differ only by the
rounding of the regrouped sum -- tolerance stated in each test.
"""
import os

import numpy as np
import pytest

import ozimmu_amd

from oracle import oracle as O
from tests.util import ColMajor, exp_rand, operand, uniform01, uniform_pm1, wide_exponent

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["resident_split", "resident_split_two_launches", "resident_split_16", "resident_split_32", "resident_split_general", "multi_view_split", "banded_split", "one_pass_split"])
def split_kernel(request, monkeypatch):
    """every test runs with every split implementation: the resident one-read kernel (what the library does for K <= 2048;
    strips of 8 rows by policy at these sizes - inside the slice GEMM's own launch where the K-split tile runs, csrc/
    slice_gemm_one_launch.hip, and as a launch of its own with OZIMMU_HIP_ONE_LAUNCH=0 - and forced to 16 and 32; compiled for the
    one or two unit blocks a wave of a short strip holds, round 6, and - resident_split_general - always for four), one launch per pass for all operand views (what it does
    for longer K), one launch per view walking row bands (operands larger than the Infinity Cache; the band size is forced
    down so that these shapes have several bands), and the one-pass kernel that re-reads its strip from L2"""
    if request.param == "resident_split_two_launches":
        monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "0")
    elif request.param in ("resident_split_16", "resident_split_32"):
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_RESIDENT", request.param[-2:])
    elif request.param == "resident_split_general":
        monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "0")
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_RESIDENT_UNITS", "4")
    elif request.param != "resident_split":
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_RESIDENT", "0")
    if request.param == "banded_split":
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_MULTI_BYTES", "0")
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_BAND_BYTES", "65536")
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_STRIP", "3")  # the k-contiguous cut's strip + prefetch form (A/B switch)
        # the per-view cut pass walks the operand from its end by default (split.hip: cut_kernel); half of the cases from its start
        import zlib
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_REVERSE", str(zlib.crc32(request.node.name.encode()) & 1))
    elif request.param == "one_pass_split":
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_ONE_PASS_BYTES", str(1 << 40))

OPS = ["N", "T"]


def _sync():
    import torch
    torch.cuda.synchronize()


# ---------------------------------------------------------------- split (rows A1-A5 of SURVEY §8a)

@pytest.mark.parametrize("matrix", ["A", "B"])
@pytest.mark.parametrize("op", OPS)
# (40, 1050): 33 k-blocks, stored as 34 (layout.h: k_blocks keeps the count even beyond 32) - the re-layout must not see it
# (300, 1536), (272, 2048): 40 / 36 strips of 8 rows, two unit blocks per wave - the resident split pairs neighbouring strips of a
# row-contiguous operand on one XCD there (split.hip, round 6), and its last group of strips is a partial one
@pytest.mark.parametrize("rows,k", [(1, 1), (5, 3), (33, 37), (64, 32), (70, 100), (130, 257), (40, 1050), (300, 1536), (272, 2048)])
@pytest.mark.parametrize("S", [3, 9, 18])
def test_split_bit_exact(oz, matrix, op, rows, k, S):
    import torch
    m_, h = oz
    rng = np.random.default_rng(1000 * rows + k + S)
    L = O.bits_per_int8(k)
    if matrix == "A":
        x = operand(op, rows, k, rng, fill=wide_exponent(6), pad=3)
        mm, nn = rows, k
    else:
        x = operand(op, k, rows, rng, fill=wide_exponent(6), pad=3)
        mm, nn = k, rows
    planes_ref, mx_ref = O.split(matrix, op, x.view, S, L)
    ldo = O.pad4(k)
    out = torch.full((S, rows, ldo), 77, dtype=torch.int8, device="cuda")
    mx = torch.full((rows,), -1.0, dtype=torch.float64, device="cuda")
    st = m_.split_int8(h, out, ldo, mx, mm, nn, x.dev, x.ld, op, m_.matrix_A if matrix == "A" else m_.matrix_B,
                       S, L)
    _sync()
    assert st == 0
    np.testing.assert_array_equal(mx.cpu().numpy().view(np.uint64), mx_ref.view(np.uint64))
    np.testing.assert_array_equal(out.cpu().numpy(), planes_ref)


def test_split_special_rows(oz):
    """zero row, subnormal-only row, subnormal inside a tiny row, Inf row, NaN row, huge row, off >= 128"""
    import torch
    m_, h = oz
    rng = np.random.default_rng(7)
    rows, k, S = 40, 50, 12
    a = operand("T", rows, k, rng)  # k-contiguous rows
    v = a.view  # (k, rows) storage of A^T: column r is row r of op(A)
    v[:, 0] = 0.0
    v[:, 1] = 5e-324 * rng.integers(1, 1000, k)       # all subnormal -> max_exp 0, zero slices
    v[:, 2] = 1e-300 * rng.uniform(-1, 1, k)          # tiny normal row ...
    v[3, 2] = 3e-310                                  # ... with a subnormal element
    v[5, 3] = np.inf
    v[6, 4] = np.nan
    v[:, 5] = rng.uniform(-1, 1, k) * 1e308           # near the top of the exponent range
    v[0, 5] = 1.7e308                                 # exponent field 0x7FE -> poisoned (2^(e+1) overflows)
    v[:, 6] = rng.uniform(-1, 1, k) * 1e-200
    v[0, 6] = 1e100                                   # spread > 2^127: small elements vanish (shift >= 128)
    v[:, 7] = -np.abs(v[:, 7])                        # all negative
    v[2, 8] = -0.0
    L = O.bits_per_int8(k)
    planes_ref, mx_ref = O.split("A", "T", a.view, S, L)
    out = torch.zeros((S, rows, O.pad4(k)), dtype=torch.int8, device="cuda")
    mx = torch.zeros((rows,), dtype=torch.float64, device="cuda")
    assert m_.split_int8(h, out, O.pad4(k), mx, rows, k, a.dev, a.ld, "T", m_.matrix_A, S, L) == 0
    _sync()
    got_mx = mx.cpu().numpy()
    assert got_mx[0] == 0 and got_mx[1] == 0
    assert np.isnan(got_mx[3]) and np.isnan(got_mx[4]) and np.isnan(got_mx[5])
    np.testing.assert_array_equal(got_mx.view(np.uint64), mx_ref.view(np.uint64))
    np.testing.assert_array_equal(out.cpu().numpy(), planes_ref)


# ---------------------------------------------------------------- INT32 slice products (row A6)

@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
@pytest.mark.parametrize("m,n,k,S", [(64, 64, 32, 3), (100, 70, 130, 6), (65, 129, 257, 9), (33, 31, 65, 13),
                                     (40, 50, 40, 18)])
def test_diagonal_sums_bit_exact(ozh, op_a, op_b, m, n, k, S):
    import torch
    m_, h = ozh   # the INT32 dump is a test hook: libozimmu_hip_test.so
    rng = np.random.default_rng(m * 7 + n * 3 + k + S)
    a = operand(op_a, m, k, rng, fill=exp_rand(2.0))
    b = operand(op_b, k, n, rng, fill=exp_rand(2.0))
    L = O.bits_per_int8(k)
    pa, _ = O.split("A", op_a, a.view, S, L)
    pb, _ = O.split("B", op_b, b.view, S, L)
    d_ref = O.diagonal_sums(pa, pb)  # [S][m][n] int64
    out = torch.full((S, n, m), 12345, dtype=torch.int32, device="cuda")
    assert m_.diagonal_sums(h, op_a, op_b, m, n, k, a.dev, a.ld, b.dev, b.ld, S, out) == 0
    _sync()
    got = out.cpu().numpy().transpose(0, 2, 1).astype(np.int64)
    np.testing.assert_array_equal(got, d_ref)


# ---------------------------------------------------------------- full DGEMM (rows A7-A10)

def _run_gemm(m_, h, op_a, op_b, m, n, k, alpha, a, b, beta, c, mode):
    st = m_.gemm(h, op_a, op_b, m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c.dev, c.ld, mode)
    _sync()
    return st


@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
@pytest.mark.parametrize("m,n,k", [(64, 64, 64), (1, 1, 1), (3, 200, 5), (127, 65, 333), (256, 192, 1030)])
@pytest.mark.parametrize("S", [3, 6, 9, 10, 11, 14, 18])
def test_gemm_bit_exact_vs_oracle_diagonal_order(oz, op_a, op_b, m, n, k, S):
    m_, h = oz
    rng = np.random.default_rng(m + 2 * n + 3 * k + S)
    a = operand(op_a, m, k, rng, pad=1)
    b = operand(op_b, k, n, rng, pad=2)
    c = ColMajor(m, n, ld=m + 5, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n, ld=m + 5)
    c_ref.buf[...] = c.buf
    mode = f"fp64_int8_{S}"
    assert _run_gemm(m_, h, op_a, op_b, m, n, k, 1.0, a, b, 0.0, c, mode) == 0
    assert O.gemm(op_a, op_b, m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    got = c.download()
    np.testing.assert_array_equal(got.view(np.uint64), c_ref.view.view(np.uint64))
    # the padding rows of C (ld > m) must be untouched (NaN poison survives)
    assert np.isnan(c.buf[:, m:]).all()


@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (-2.5, 0.0), (1.0, 1.0), (0.75, -1.25)])
def test_gemm_alpha_beta(oz, alpha, beta):
    m_, h = oz
    m, n, k, S = 130, 70, 200, 9
    rng = np.random.default_rng(5)
    a = operand("N", m, k, rng)
    b = operand("T", k, n, rng)
    c = ColMajor(m, n, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n)
    c_ref.buf[...] = c.buf
    assert _run_gemm(m_, h, "N", "T", m, n, k, alpha, a, b, beta, c, "fp64_int8_9") == 0
    assert O.gemm("N", "T", m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


@pytest.mark.parametrize("S", [6, 8, 9, 12])
def test_gemm_vs_reference_pair_order(oz, S):
    """Against the reference's pair-by-pair FP64 accumulation (src/gemm.cu:385-403) the fused kernel
    differs only by regrouping the same exact terms: |diff| <= 4 ulp of the largest partial sum, i.e.
    <= 4 * 2^-52 * max|C| * 2 here, and the residual vs the long-double truth is not worse."""
    m_, h = oz
    m, n, k = 200, 150, 512
    rng = np.random.default_rng(11)
    a = operand("T", m, k, rng)
    b = operand("N", k, n, rng)
    c = ColMajor(m, n)
    c_ref = ColMajor(m, n)
    assert _run_gemm(m_, h, "T", "N", m, n, k, 1.0, a, b, 0.0, c, f"fp64_int8_{S}") == 0
    O.gemm("T", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_REFERENCE)
    got = c.download()
    scale = np.abs(c_ref.view).max()
    assert np.abs(got - c_ref.view).max() <= 8 * 2.0 ** -52 * scale
    r_hip = O.relative_residual("T", "N", m, n, k, a.view, b.view, got)
    r_ref = O.relative_residual("T", "N", m, n, k, a.view, b.view, c_ref.view)
    assert r_hip <= 1.05 * r_ref + 1e-17


def test_gemm_k_chunking_large_k(oz):
    """K above the INT32-safe length per pass (S*K*127^2 < 2^31): two passes through the FP64 acc
    workspace, same fma chain as the oracle with kchunk."""
    m_, h = oz
    m, n, k, S = 64, 64, 20000, 9
    rng = np.random.default_rng(3)
    a = operand("N", m, k, rng)
    b = operand("N", k, n, rng)
    c = ColMajor(m, n)
    c_ref = ColMajor(m, n)
    assert _run_gemm(m_, h, "N", "N", m, n, k, 1.0, a, b, 0.0, c, "fp64_int8_9") == 0
    kchunk = (2147483647 // (S * 127 * 127)) // 64 * 64   # the library keeps a pass at an even number of 32-k blocks
    O.gemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL, kchunk=kchunk)
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


@pytest.mark.parametrize("k", [22145, 22176, 22177])
def test_gemm_k_at_the_pass_boundary(oz, k):
    """S = 6, L = 7: 2^31 / (6 * 127^2) = 22190 k fit one pass, i.e. 693 k-blocks - an odd count, while the planes hold an
    even one (694 for K = 22145..22176).  The library's pass length is a multiple of 64 k (22144), so these K take two
    chained passes AND get an FP64 workspace for them (round 3 sized the workspace for one pass and then ran two: ADVICE r3)"""
    m_, h = oz
    m, n, S = 40, 72, 6
    assert O.bits_per_int8(k) == 7
    rng = np.random.default_rng(k)
    a = operand("T", m, k, rng)
    b = operand("N", k, n, rng)
    c = ColMajor(m, n)
    c_ref = ColMajor(m, n)
    assert _run_gemm(m_, h, "T", "N", m, n, k, 1.0, a, b, 0.0, c, "fp64_int8_6") == 0
    kchunk = (2147483647 // (S * 127 * 127)) // 64 * 64
    assert kchunk == 22144
    O.gemm("T", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL, kchunk=kchunk)
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


@pytest.mark.parametrize("m,n,k,S", [(40, 24, 140000, 9), (40, 24, 140000, 14), (24, 24, 530000, 9)])
def test_gemm_narrow_slices_for_very_long_k(oz, m, n, k, S):
    """k > 2^17 -> 6-bit slices, k > 2^19 -> 5-bit slices (get_bits_per_int8, src/split.cu:520-536); the INT32-safe
    pass length follows (2^L - 1)^2, and S = 14 combines K-chunking with the two diagonal passes"""
    m_, h = oz
    L = O.bits_per_int8(k)
    assert L == (6 if k < (1 << 19) else 5)
    rng = np.random.default_rng(k + S)
    a = operand("T", m, k, rng, fill=wide_exponent(3))
    b = operand("N", k, n, rng)
    c = ColMajor(m, n)
    c_ref = ColMajor(m, n)
    assert _run_gemm(m_, h, "T", "N", m, n, k, 1.0, a, b, 0.0, c, f"fp64_int8_{S}") == 0
    kchunk = (2147483647 // (S * ((1 << L) - 1) ** 2)) // 64 * 64   # an even number of 32-k blocks per pass
    O.gemm("T", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL, kchunk=kchunk)
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))
    assert O.relative_residual(
        "T", "N", m, n, k, a.view, b.view, c.view) < (1e-13 if S * L >= 54 else 1e-9)


def test_gemm_argument_errors(oz):
    """src/gemm.cu:535-556: bad leading dimension / misaligned pointer -> 1, nothing written"""
    import torch
    m_, h = oz
    a = torch.zeros(64 * 64, dtype=torch.float64, device="cuda")
    c = torch.full((64 * 64,), 3.0, dtype=torch.float64, device="cuda")
    assert m_.gemm(h, "N", "N", 64, 64, 64, 1.0, a, 63, a, 64, 0.0, c, 64, "fp64_int8_6") == 1
    assert m_.gemm(h, "T", "N", 64, 64, 32, 1.0, a, 16, a, 64, 0.0, c, 64, "fp64_int8_6") == 1
    assert m_.gemm(h, "N", "N", 64, 64, 64, 1.0, a, 64, a, 64, 0.0, c, 10, "fp64_int8_6") == 1
    assert m_.gemm(h, "N", "N", 64, 64, 64, 1.0, a.data_ptr() + 4, 64, a, 64, 0.0, c, 64, "fp64_int8_6") == 1
    _sync()
    assert (c == 3.0).all()


# ---------------------------------------------------------------- the reference's own gate

@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
@pytest.mark.parametrize("mnk", [(1023, 1024, 1025), (1025, 1023, 1024), (1024, 1025, 1023)])
@pytest.mark.parametrize("S", [8, 12, 16])
def test_reference_ci_gate(oz, op_a, op_b, mnk, S):
    """test/main_test.cu:702-746: urand01 inputs, alpha=1, beta=0, relative residual < 1e-15"""
    m_, h = oz
    m, n, k = mnk
    rng = np.random.default_rng(0)
    a = operand(op_a, m, k, rng, fill=uniform01)
    b = operand(op_b, k, n, rng, fill=uniform01)
    c = ColMajor(m, n)
    assert _run_gemm(m_, h, op_a, op_b, m, n, k, 1.0, a, b, 0.0, c, f"fp64_int8_{S}") == 0
    r = O.relative_residual_sampled(op_a, op_b, m, n, k, a.view, b.view, c.download(), ns=4096)
    assert r < 1e-15


# ---------------------------------------------------------------- auto mode (row A11)

@pytest.mark.parametrize("fill,thr", [(uniform_pm1, 1.5), (wide_exponent(8), 1.5), (uniform_pm1, 0.0),
                                      (wide_exponent(30), 0.0)])
@pytest.mark.parametrize("op_a,op_b", [("N", "T"), ("T", "N")])
def test_auto_mode_counters_and_selection(oz, fill, thr, op_a, op_b):
    m_, h = oz
    m, n, k = 150, 90, 300
    rng = np.random.default_rng(21)
    a = operand(op_a, m, k, rng, fill=fill)
    b = operand(op_b, k, n, rng, fill=fill)
    a.view[3, 4] = 0.0
    s_ref, cnt_ref = O.auto_select(op_a, op_b, m, n, k, a.view, b.view, thr)
    cnt = m_.mantissa_loss(h, op_a, op_b, m, n, k, a.dev, a.ld, b.dev, b.ld)
    assert cnt == cnt_ref.tolist()
    mode = m_.auto_mode_select(h, op_a, op_b, m, n, k, a.dev, a.ld, b.dev, b.ld, m_.real, thr)
    expect = m_.dgemm if s_ref == 0 else m_.fp64_int8_3 + (s_ref - 3)
    assert mode == expect


# ---------------------------------------------------------------- committed golden fixtures (tests/golden)

import glob  # noqa: E402
import os  # noqa: E402

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _dev_colmajor(arr):
    """device copy of a column-major (Fortran) 2-D array; returns (tensor, ld)"""
    import torch
    assert arr.flags.f_contiguous
    return torch.from_numpy(np.ascontiguousarray(arr.T)).cuda(), arr.shape[0]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_path_reproduces_golden(oz, ozh, path):
    import torch
    m_, h = oz
    mt_, ht = ozh   # only the INT32 diagonal sums come from the test flavour; slices and the FP64 result from the product
    g = np.load(path)
    op_a, op_b = str(g["op_a"]), str(g["op_b"])
    m, n, k, S, L = (int(g[x]) for x in ("m", "n", "k", "S", "L"))
    a, lda = _dev_colmajor(np.asfortranarray(g["a"]))
    b, ldb = _dev_colmajor(np.asfortranarray(g["b"]))
    ldo = g["planes_a"].shape[2]
    # slices + max_exp
    for which, rows, ref_p, ref_e, (mm, nn), src, ld, op, mat in (
            ("A", m, g["planes_a"], g["max_exp_a"], (m, k), a, lda, op_a, m_.matrix_A),
            ("B", n, g["planes_b"], g["max_exp_b"], (k, n), b, ldb, op_b, m_.matrix_B)):
        out = torch.zeros((S, rows, ldo), dtype=torch.int8, device="cuda")
        mx = torch.zeros((rows,), dtype=torch.float64, device="cuda")
        assert m_.split_int8(h, out, ldo, mx, mm, nn, src, ld, op, mat, S, L) == 0
        _sync()
        np.testing.assert_array_equal(out.cpu().numpy(), ref_p)
        np.testing.assert_array_equal(mx.cpu().numpy().view(np.uint64), ref_e.view(np.uint64))
    # INT32 diagonal sums
    d = torch.zeros((S, n, m), dtype=torch.int32, device="cuda")
    assert mt_.diagonal_sums(ht, op_a, op_b, m, n, k, a, lda, b, ldb, S, d) == 0
    _sync()
    np.testing.assert_array_equal(d.cpu().numpy().transpose(0, 2, 1).astype(np.int64), g["diag"])
    assert m_.diagonal_sums(h, op_a, op_b, m, n, k, a, lda, b, ldb, S, d) == 2   # the product carries no dump hook
    # FP64 result, bit-exact in the kernel's grouping; a few ulp of the partial-sum scale from the reference's
    c, ldc = _dev_colmajor(np.asfortranarray(g["c0"]))
    assert m_.gemm(h, op_a, op_b, m, n, k, float(g["alpha"]), a, lda, b, ldb, float(g["beta"]), c, ldc,
                   f"fp64_int8_{S}") == 0
    _sync()
    got = c.cpu().numpy().T
    np.testing.assert_array_equal(got.view(np.uint64), np.ascontiguousarray(g["c_diagonal_order"]).view(np.uint64))
    fin = np.isfinite(g["c_reference_order"])
    scale = np.abs(g["c_reference_order"][fin]).max()
    assert np.abs(got[fin] - g["c_reference_order"][fin]).max() <= 16 * 2.0 ** -52 * scale
    assert (np.isnan(got) == np.isnan(g["c_reference_order"])).all()
    # auto mode
    cnt = m_.mantissa_loss(h, op_a, op_b, m, n, k, a, lda, b, ldb)
    assert cnt == g["auto_counters"].tolist()
    sel = int(g["auto_selected"])
    mode = m_.auto_mode_select(h, op_a, op_b, m, n, k, a, lda, b, ldb, m_.real, 1.5)
    assert mode == (m_.dgemm if sel == 0 else m_.fp64_int8_3 + sel - 3)


@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
@pytest.mark.parametrize("cplx", [False, True])
def test_sgemm_mode_matches_an_fp32_product(oz, op_a, op_b, cplx):
    """the `sgemm` compute mode (src/cublas_helper.cu:83-133): C = f64(alpha32 * f32(A) f32(B) + beta32 * f32(C)).
    The vendor SGEMM's summation order is its own, so the tolerance is FP32's (a few k * 2^-24 relative to |A||B|);
    every stored value must be exactly representable in FP32, and the padding rows of C stay untouched."""
    m_, h = oz
    m, n, k = 150, 70, 333
    rng = np.random.default_rng(7)
    dt = np.complex128 if cplx else np.float64

    def mk(op, rows, cols, pad):
        r, c = (rows, cols) if op == "N" else (cols, rows)
        x = ColMajor(r, c, ld=r + pad, dtype=dt)
        v = rng.uniform(-1, 1, (c, r))
        x.buf[:, :r] = v + 1j * rng.uniform(-1, 1, (c, r)) if cplx else v
        x.buf[:, r:] = np.nan
        return x
    a, b, c = mk(op_a, m, k, 1), mk(op_b, k, n, 2), mk("N", m, n, 3)
    c0 = c.view.copy()
    kind = m_.complx if cplx else m_.real
    alpha, beta = (1.25 - 0.5j, -0.75 + 0.25j) if cplx else (1.25, -0.75)
    for be, mode_api in ((beta, "gemm_f32"), (0.0, "gemm")):
        c.buf[:, :m] = c0.T
        c.buf[:, m:] = np.nan
        c._dev = None            # re-upload
        if mode_api == "gemm_f32":
            st = m_.gemm_f32(h, op_a, op_b, m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, be, c.dev, c.ld, kind)
        else:  # the same path through the mode switch
            st = m_.gemm(h, op_a, op_b, m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, be, c.dev, c.ld, "sgemm", kind)
        import torch
        torch.cuda.synchronize()
        assert st == 0
        got = c.download()
        f32 = np.complex64 if cplx else np.float32
        av = (a.view if op_a == "N" else a.view.T).astype(f32)
        bv = (b.view if op_b == "N" else b.view.T).astype(f32)
        want = f32(alpha) * (av.astype(dt) @ bv.astype(dt)) + (f32(be) * c0.astype(f32).astype(dt) if be != 0 else 0)
        scale = np.abs(av).astype(np.float64) @ np.abs(bv).astype(np.float64) * abs(alpha) + abs(be) * np.abs(c0) + 1e-30
        assert np.max(np.abs(got - want) / scale) < 64 * 2.0 ** -24
        np.testing.assert_array_equal(got, got.astype(f32).astype(dt))     # values came out of an FP32 GEMM
        assert np.isnan(c.buf[:, m:]).all()


def test_malloc_async_mode_and_side_stream():
    """OZIMMU_MALLOC_ASYNC / malloc_async (src/handle.cu:63-93): the workspace is (re)allocated in stream order on the
    caller's stream -- here a non-default stream -- and growing it between calls must not disturb results"""
    import torch
    h = ozimmu_amd.create(ozimmu_amd.malloc_async)
    st = torch.cuda.Stream()
    try:
        ozimmu_amd.set_cuda_stream(h, st)
        rng = np.random.default_rng(21)
        for (m, n, k, S) in [(64, 64, 64, 6), (300, 200, 500, 9), (70, 33, 129, 13), (512, 384, 1100, 9)]:
            a = operand("N", m, k, rng, fill=wide_exponent(4), pad=1)
            b = operand("T", k, n, rng, fill=uniform_pm1, pad=2)
            c = ColMajor(m, n)
            c_ref = ColMajor(m, n)
            with torch.cuda.stream(st):
                a.dev, b.dev, c.dev  # uploads ordered on the side stream
            st.synchronize()
            assert ozimmu_amd.gemm(h, "N", "T", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld,
                                   f"fp64_int8_{S}") == 0
            st.synchronize()
            assert O.gemm("N", "T", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL) == 0
            np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))
    finally:
        st.synchronize()
        ozimmu_amd.destroy(h)


@pytest.mark.parametrize("seed", range(6))
def test_gemm_fuzz_bit_exact(oz, seed):
    """seeded fuzz over shapes (ragged, tiny, one tile row/column short of a boundary), all op pairs, S = 3..18,
    alpha/beta, leading-dimension padding and input distributions: every case bit-exact against the oracle"""
    m_, h = oz
    rng = np.random.default_rng(9000 + seed)
    fills = [uniform_pm1, uniform01, exp_rand(2.0), wide_exponent(8)]
    for case in range(10):
        edge = [1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 257]
        m = int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, 300))
        n = int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, 300))
        k = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 255, 256, 257, 640])) if rng.random() < 0.5 \
            else int(rng.integers(1, 700))
        S = int(rng.integers(3, 19))
        op_a, op_b = rng.choice(OPS), rng.choice(OPS)
        alpha = float(rng.choice([1.0, -1.0, 0.5, 3.25]))
        beta = float(rng.choice([0.0, 0.0, 1.0, -0.75]))
        a = operand(op_a, m, k, rng, fill=fills[int(rng.integers(4))], pad=int(rng.integers(0, 4)))
        b = operand(op_b, k, n, rng, fill=fills[int(rng.integers(4))], pad=int(rng.integers(0, 4)))
        c = ColMajor(m, n, ld=m + int(rng.integers(0, 3)), fill=uniform_pm1, rng=rng)
        c_ref = ColMajor(m, n, ld=c.ld)
        c_ref.buf[...] = c.buf
        assert _run_gemm(m_, h, op_a, op_b, m, n, k, alpha, a, b, beta, c, f"fp64_int8_{S}") == 0
        assert O.gemm(op_a, op_b, m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
        got = c.download()
        assert np.array_equal(got.view(np.uint64), c_ref.view.view(np.uint64)), \
            (seed, case, op_a, op_b, m, n, k, S, alpha, beta)


@pytest.mark.parametrize("kind", ["real", "complex"])
def test_auto_mode_gemm_equals_the_selected_mode_bitwise(oz, kind):
    """fp64_int8_auto runs the statistic pass and then the GEMM of the mode it selects; the GEMM takes over the row maxima
    the statistic pass has just computed instead of computing them again -- the result must be the selected mode's, bit for
    bit (and a later plain call on the same, meanwhile MODIFIED operands must not see them)"""
    import torch
    m_, h = oz
    m, n, k = 150, 130, 96
    rng = np.random.default_rng(2024)
    cplx = kind == "complex"
    ek = m_.complx if cplx else m_.real
    dt = torch.complex128 if cplx else torch.float64
    def rnd(r, c):
        x = torch.from_numpy(rng.uniform(-1, 1, (c, r)) * np.exp2(rng.integers(-6, 6, (c, 1)).astype(np.float64)))
        return (torch.complex(x, x.flip(0)) if cplx else x).to(dt).cuda().contiguous()
    a, b = rnd(m, k), rnd(k, n)
    m_.set_auto_mantissa_loss_threashold(h, 1.0)
    sel = m_.auto_mode_select(h, "N", "N", m, n, k, a, m, b, k, ek, 1.0)  # mode id (gemm takes ids or names)
    assert "fp64_int8_" in m_.get_compute_mode_name_str(sel)
    c_auto = torch.zeros(n, m, dtype=dt, device="cuda")
    c_sel = torch.zeros(n, m, dtype=dt, device="cuda")
    assert m_.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c_auto, m, "fp64_int8_auto", ek) == 0
    assert m_.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c_sel, m, sel, ek) == 0
    _sync()
    assert torch.equal(torch.view_as_real(c_auto).view(torch.int64) if cplx else c_auto.view(torch.int64),
                       torch.view_as_real(c_sel).view(torch.int64) if cplx else c_sel.view(torch.int64))
    # the operands change in place (rows get 2^20 times larger): a plain call right after an auto call must compute
    # its own row maxima
    assert m_.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c_auto, m, "fp64_int8_auto", ek) == 0
    a.mul_(2.0 ** 20)
    c1 = torch.zeros(n, m, dtype=dt, device="cuda")
    c2 = torch.zeros(n, m, dtype=dt, device="cuda")
    assert m_.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c1, m, sel, ek) == 0
    h2 = m_.create()
    try:
        assert m_.gemm(h2, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c2, m, sel, ek) == 0
        _sync()
    finally:
        m_.destroy(h2)
    assert torch.equal(torch.view_as_real(c1).view(torch.int64) if cplx else c1.view(torch.int64),
                       torch.view_as_real(c2).view(torch.int64) if cplx else c2.view(torch.int64))


@pytest.mark.parametrize("seed", range(int(os.environ.get("OZIMMU_TEST_FUZZ_SEEDS", "2"))))
def test_gemm_fuzz_forced_kernels_bit_exact(oz, monkeypatch, seed):
    """second fuzz: mid-size shapes (several tiles of every kernel, mixed tile heights, ragged edges) with the slice-GEMM
    kernel forced at random (default policy / wide / classic / K-split), real and complex, batches of 1..3: bit-exact
    against the oracle.  OZIMMU_TEST_FUZZ_SEEDS=n runs a longer campaign."""
    import torch
    m_, h = oz
    rng = np.random.default_rng(77000 + seed)
    for case in range(8):
        m, n = int(rng.integers(1, 620)), int(rng.integers(1, 620))
        k = int(rng.choice([33, 64, 96, 130, 257, 400]))
        S = int(rng.choice([3, 4, 6, 8, 9, 10, 12, 13, 14, 16, 18]))
        kernel = rng.choice(["", "wide", "classic", "k2"])
        if kernel:
            monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", str(kernel))
        else:
            monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
        op_a, op_b = rng.choice(OPS), rng.choice(OPS)
        alpha = float(rng.choice([1.0, -0.5, 2.75]))
        beta = float(rng.choice([0.0, 1.0, -0.25]))
        a = operand(op_a, m, k, rng, fill=exp_rand(2.0), pad=int(rng.integers(0, 3)))
        b = operand(op_b, k, n, rng, fill=uniform_pm1, pad=int(rng.integers(0, 3)))
        c = ColMajor(m, n, ld=m + int(rng.integers(0, 3)), fill=uniform_pm1, rng=rng)
        c_ref = ColMajor(m, n, ld=c.ld)
        c_ref.buf[...] = c.buf
        assert m_.gemm(h, op_a, op_b, m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c.dev, c.ld, f"fp64_int8_{S}") == 0
        _sync()
        assert O.gemm(op_a, op_b, m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
        got = c.download()
        assert np.array_equal(got.view(np.uint64), c_ref.view.view(np.uint64)), \
            (seed, case, kernel, op_a, op_b, m, n, k, S, alpha, beta)


def test_exponent_words_grow_in_stream_order_in_async_mode():
    """malloc_async: the row-exponent words (1 MiB to start with: 262 144 rows) are re-allocated and zero-filled on the
    caller's stream like the workspace (no hipFree / hipMalloc / hipMemset: SURVEY 8(b) "never device-sync on growth").
    A 24 x 280 000 x 40 product outgrows them between two ordinary calls; same bits as a malloc_sync handle."""
    import torch
    m, n, k, S = 24, 280000, 40, 9
    g = torch.Generator(device="cuda").manual_seed(4)
    a = torch.rand(k, m, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    b = torch.rand(n, k, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    a_s = torch.rand(64, 96, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    b_s = torch.rand(80, 64, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    out = {}
    for mode in (ozimmu_amd.malloc_sync, ozimmu_amd.malloc_async):
        h = ozimmu_amd.create(mode)
        st = torch.cuda.Stream()
        try:
            ozimmu_amd.set_cuda_stream(h, st)
            c_s = torch.zeros(80, 96, dtype=torch.float64, device="cuda")
            c = torch.full((n, m), float("nan"), dtype=torch.float64, device="cuda")
            st.wait_stream(torch.cuda.current_stream())   # inputs and the fills above are ordered in front of the calls
            assert ozimmu_amd.gemm(h, "N", "N", 96, 80, 64, 1.0, a_s, 96, b_s, 64, 0.0, c_s, 96, f"fp64_int8_{S}") == 0
            assert ozimmu_amd.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c, m, f"fp64_int8_{S}") == 0   # grows
            assert ozimmu_amd.gemm(h, "N", "N", 96, 80, 64, 1.0, a_s, 96, b_s, 64, 0.0, c_s, 96, f"fp64_int8_{S}") == 0
            st.synchronize()
            out[mode] = (c.clone(), c_s.clone())
        finally:
            st.synchronize()
            ozimmu_amd.destroy(h)
    for x, y in zip(out[ozimmu_amd.malloc_sync], out[ozimmu_amd.malloc_async]):
        assert torch.equal(x.view(torch.int64), y.view(torch.int64))
    ref = b @ a
    assert ((out[ozimmu_amd.malloc_async][0] - ref).norm() / ref.norm()).item() < 1e-14


@pytest.mark.parametrize("m,n,k,S", [
    (1100, 1500, 256, 7),    # k64 tile, 96-row tiles with 64-row reduced ones in the plan (S = 7 discount rule)
    (2100, 1900, 512, 7),
    (1800, 2300, 256, 9),    # 8 k-blocks at S = 9: 32x32x32 tile function
    (1800, 2300, 384, 9),    # 12 k-blocks: k64 tile
    (2500, 1300, 128, 9),    # below the short-loop bar: classic kernel
    (2200, 2200, 256, 6),    # S = 6: wide / k64 from K = 256
    (2050, 1000, 128, 6),
    (1700, 2600, 512, 4),
    (1500, 1500, 384, 4),    # S = 4 below its bar: classic
    (1900, 2100, 256, 10),
    (1300, 3100, 512, 10),
    (2300, 1700, 1056, 8),   # odd number of k-blocks: the k64 tile is not eligible
])
def test_gemm_default_policy_mid_shapes_bit_exact(oz, m, n, k, S):
    """the kernel choice of slice_gemm_launch.h (short-loop bars, k64 eligibility, plan comparison) on mid-size shapes with
    short K, as the policy picks them - no kernel forced: bit-exact against the oracle whatever it picks"""
    m_, h = oz
    rng = np.random.default_rng(m * 7 + n * 3 + k + S)
    a = operand("N", m, k, rng, fill=uniform_pm1, pad=1)
    b = operand("T", k, n, rng, fill=exp_rand(2.0), pad=0)
    c = ColMajor(m, n, ld=m + 1, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n, ld=c.ld)
    c_ref.buf[...] = c.buf
    assert _run_gemm(m_, h, "N", "T", m, n, k, 0.5, a, b, -0.75, c, f"fp64_int8_{S}") == 0
    assert O.gemm("N", "T", m, n, k, 0.5, a.view, b.view, -0.75, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    assert np.array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))
