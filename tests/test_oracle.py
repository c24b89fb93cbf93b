"""CPU tests of the oracle (oracle/ozaki_oracle.c): pins the restatement of the reference algorithm.

What the reference itself pins on this path: only the CI gate `relative_residual < 1e-15`
for fp64_int8_8..16 on uniform(0,1] inputs at m,n,k in {1023,1024,1025}, all four op combinations
(test/main_test.cu:702-746).  Everything else is pinned here by exact identities of the algorithm and by
the committed golden fixtures (tests/golden/, produced by this oracle once it passed these tests).
"""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import ColMajor, exp_rand, operand, uniform01, uniform_pm1, wide_exponent

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


# ------------------------------------------------------------------ host logic (rows A1, A2)

def test_bits_per_int8_table():
    # src/split.cu:520-536: L = min(7, (31 - ceil(log2 k)) / 2)
    assert O.bits_per_int8(0) == 0
    for k, L in [(1, 7), (2, 7), (1024, 7), (8192, 7), (1 << 17, 7), ((1 << 17) + 1, 6), (1 << 19, 6),
                 ((1 << 19) + 1, 5), (1 << 21, 5), ((1 << 21) + 1, 4), (1 << 23, 4), (1 << 25, 3)]:
        assert O.bits_per_int8(k) == L, k
    # the defining property: one pair product can never overflow int32
    for k in [1, 3, 1000, 1 << 17, (1 << 17) + 1, 3_000_000]:
        L = O.bits_per_int8(k)
        assert k * ((1 << L) - 1) ** 2 < 2 ** 31


def test_mode_names():
    # src/handle.cu:146-192, src/cublas.cu:18-48
    assert O.num_split_from_mode("dgemm") == -1
    assert O.num_split_from_mode("sgemm") == -2
    assert O.num_split_from_mode("fp64_int8_auto") == 0
    for s in range(3, 19):
        assert O.num_split_from_mode(f"fp64_int8_{s}") == s
    for bad in ["fp64_int8_2", "fp64_int8_19", "fp64_int8_", "fp64_int8_09", "FP64_INT8_9", "", "int8"]:
        assert O.num_split_from_mode(bad) == -3, bad


@pytest.mark.parametrize("S", range(3, 19))
def test_pair_list(S):
    # src/config.cu:85-93: P = S(S+1)/2 pairs, ordered by i+j ascending then i ascending, all i+j <= S+1
    pairs = O.pair_list(S)
    assert len(pairs) == S * (S + 1) // 2
    assert pairs == [(i, t - i) for t in range(2, S + 2) for i in range(1, t) if i <= S and t - i <= S]
    assert len(set(pairs)) == len(pairs)
    assert pairs[0] == (1, 1)


# ------------------------------------------------------------------ split (rows A3-A5)

def _reconstruct(planes, mx, L):
    """a_trunc = max_exp * 2 * sum_s slice_s * 2^(-(s+1)L)  (SURVEY §8a row A4), exactly in Python ints"""
    S, rows, ldo = planes.shape
    out = np.zeros((rows, ldo), dtype=object)
    for s in range(S):
        out = out * (1 << L) + planes[s].astype(object)
    return out, S * L  # integer mantissa, number of fraction bits: value = mx*2 * out / 2^(S*L)


@pytest.mark.parametrize("op", ["N", "T"])
@pytest.mark.parametrize("S,L", [(3, 7), (9, 7), (18, 7), (9, 5)])
def test_split_reconstructs_truncated_value(op, S, L):
    rng = np.random.default_rng(S * 10 + L)
    rows, k = 17, 23
    a = operand(op, rows, k, rng, fill=wide_exponent(5))
    planes, mx = O.split("A", op, a.view, S, L)
    assert planes.shape == (S, rows, O.pad4(k))
    assert (planes[:, :, k:] == 0).all()                     # zero padding (src/split.cu:222-232)
    vals = a.view if op == "N" else a.view.T
    ints, frac = _reconstruct(planes, mx, L)
    for r in range(rows):
        e = int(np.frexp(mx[r])[1]) - 1                        # mx = 2^e, e = e_max + 1
        assert mx[r] == 2.0 ** e and mx[r] >= 2 * np.abs(vals[r]).max() / 2 and mx[r] > np.abs(vals[r]).max()
        for kk in range(k):
            x = float(vals[r, kk])
            # exact: x*2^(frac-1-e) truncated toward zero == ints   (value = 2*mx*ints/2^frac)
            from fractions import Fraction
            scaled = Fraction(x) * Fraction(2) ** (frac - 1 - e)
            trunc = int(scaled) if scaled >= 0 else -int(-scaled)
            assert trunc == ints[r, kk], (r, kk)
    # slice ranges: first slice < 2^(L-1) in magnitude, the others < 2^L; all carry the sign of the element
    assert np.abs(planes[0].astype(int)).max() < (1 << (L - 1))
    assert np.abs(planes.astype(int)).max() < (1 << L)
    sign = np.sign(vals)
    for s in range(S):
        p = planes[s, :, :k].astype(int)
        assert ((p == 0) | (np.sign(p) == sign)).all()


def test_split_layouts_agree():
    """A with op N and the same matrix handed over transposed with op T give identical slices; same for B."""
    rng = np.random.default_rng(3)
    m, k = 19, 31
    x = rng.uniform(-1, 1, (m, k))
    an = ColMajor(m, k)
    an.view[...] = x
    at = ColMajor(k, m)
    at.view[...] = x.T
    p1, e1 = O.split("A", "N", an.view, 7)
    p2, e2 = O.split("A", "T", at.view, 7)
    np.testing.assert_array_equal(p1, p2)
    np.testing.assert_array_equal(e1, e2)
    # B: op(B) is k x n; slices are per column of op(B)
    bn = ColMajor(k, m)
    bn.view[...] = x.T
    bt = ColMajor(m, k)
    bt.view[...] = x
    p3, e3 = O.split("B", "N", bn.view, 7)
    p4, e4 = O.split("B", "T", bt.view, 7)
    np.testing.assert_array_equal(p3, p4)
    np.testing.assert_array_equal(p3, p1)
    np.testing.assert_array_equal(e3, e1)
    np.testing.assert_array_equal(e4, e1)


def test_split_special_rows_policy():
    """documented deviations: zero/subnormal rows -> max_exp 0; Inf/NaN/2^1023 rows -> NaN"""
    k = 9
    a = ColMajor(k, 6)  # op T storage: column r = row r of op(A)
    v = a.view
    v[:, 0] = 0.0
    v[:, 1] = 5e-324
    v[:, 2] = 1.0
    v[4, 2] = np.inf
    v[:, 3] = 1.0
    v[2, 3] = np.nan
    v[:, 4] = 1e308
    v[0, 4] = 1.7e308
    v[:, 5] = [1e100, 1e-100, -1.0, 0.5, -0.0, 3.0, 1e99, -1e100, 2e-300]
    planes, mx = O.split("A", "T", a.view, 18)
    assert mx[0] == 0 and mx[1] == 0 and np.isnan(mx[2]) and np.isnan(mx[3]) and np.isnan(mx[4])
    assert (planes[:, :5, :] == 0).all()
    assert mx[5] == 2.0 ** 333  # 1e100 = 1.14 * 2^332
    assert (planes[:, 5, 1] == 0).all()   # 1e-100 is > 2^127 below the row maximum: vanishes (shift >= 128)
    assert (planes[:, 5, 8] == 0).all()
    assert planes[0, 5, 7] < 0 and planes[0, 5, 0] > 0
    # reference quirk mode only changes subnormal elements of non-subnormal rows
    p_fix, _ = O.split("A", "T", a.view, 18, quirks=0)
    p_ref, _ = O.split("A", "T", a.view, 18, quirks=O.QUIRK_REF_SUBNORMAL)
    np.testing.assert_array_equal(p_fix, p_ref)  # nothing subnormal survives here


def test_subnormal_element_fix_vs_reference_quirk():
    """a subnormal next to a tiny normal maximum: the reference shifts it one bit too far (src/split.cu:161-173)"""
    a = ColMajor(2, 1)
    a.view[0, 0] = 2.0 ** -1020
    a.view[1, 0] = 3 * 2.0 ** -1030  # subnormal, exactly representable with few bits
    fixed, mx = O.split("A", "T", a.view, 6, 7)
    quirk, _ = O.split("A", "T", a.view, 6, 7, quirks=O.QUIRK_REF_SUBNORMAL)
    ints_f, frac = _reconstruct(fixed, mx, 7)
    ints_q, _ = _reconstruct(quirk, mx, 7)
    val = lambda ints: float(ints) * 2.0 * mx[0] / 2.0 ** frac  # noqa: E731
    assert val(ints_f[0, 1]) == 3 * 2.0 ** -1030       # exact with the fix
    assert val(ints_q[0, 1]) == 1.5 * 2.0 ** -1030     # halved by the reference (truncated: 3/2 -> kept bits)


# ------------------------------------------------------------------ INT8 GEMM + accumulate (rows A6-A8)

def test_int8_gemm_is_exact_integer_matmul():
    rng = np.random.default_rng(0)
    m, n, kp = 37, 29, 64
    a = rng.integers(-127, 128, (m, kp)).astype(np.int8)
    b = rng.integers(-127, 128, (n, kp)).astype(np.int8)
    c = O.int8_gemm(a, b)
    np.testing.assert_array_equal(c.astype(np.int64), a.astype(np.int64) @ b.astype(np.int64).T)
    d = O.diagonal_sums(np.stack([a, a]), np.stack([b, b]))  # S=2 layout: t=2 -> a*b ; t=3 -> 2 pairs
    np.testing.assert_array_equal(d[0], c.astype(np.int64))
    np.testing.assert_array_equal(d[1], 2 * c.astype(np.int64))


@pytest.mark.parametrize("order", [O.ORDER_REFERENCE, O.ORDER_DIAGONAL])
def test_gemm_equals_python_emulation(order):
    """the C oracle against an independent numpy restatement of src/gemm.cu:385-409 on a tiny case"""
    rng = np.random.default_rng(1)
    m, n, k, S = 9, 7, 11, 5
    a = operand("N", m, k, rng)
    b = operand("N", k, n, rng)
    c = ColMajor(m, n)
    assert O.gemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, c.view, S, order) == 0
    L = O.bits_per_int8(k)
    pa, ea = O.split("A", "N", a.view, S, L)
    pb, eb = O.split("B", "N", b.view, S, L)
    acc = np.zeros((m, n))
    if order == O.ORDER_REFERENCE:
        for (i, j) in O.pair_list(S):
            c32 = pa[i - 1].astype(np.int64) @ pb[j - 1].astype(np.int64).T
            rshift = L * (i + j - 2) - (7 - L) * 2
            acc += (c32 * 2.0 ** 32) * 2.0 ** (-rshift)          # one rounded += per pair
    else:
        for t in range(2, S + 2):
            d = sum(pa[i - 1].astype(np.int64) @ pb[t - i - 1].astype(np.int64).T
                    for i in range(1, t) if i <= S and t - i <= S)
            rshift = L * (t - 2) - (7 - L) * 2
            acc = acc + d.astype(np.float64) * (2.0 ** 32 * 2.0 ** (-rshift))  # product exact -> same as fma
    expect = acc / 2.0 ** 44 * ea[:, None] * eb[None, :]
    np.testing.assert_array_equal(c.view, expect)


def test_gemm_shape_errors_and_empty():
    a = np.zeros((4, 4), order="F")
    c = np.zeros((4, 4), order="F")
    assert O.gemm("N", "N", 8, 4, 4, 1.0, a, a, 0.0, c, 6) == 1     # lda < m (src/gemm.cu:535-556)
    assert O.gemm("N", "N", 4, 4, 4, 1.0, a, a, 0.0, c, 2) == 1     # S outside 3..18
    assert O.gemm("N", "N", 0, 4, 4, 1.0, a, a, 0.0, c, 6) == 0


def test_beta_and_alpha():
    rng = np.random.default_rng(5)
    m, n, k, S = 12, 10, 33, 9
    a = operand("T", m, k, rng)
    b = operand("T", k, n, rng)
    c0 = ColMajor(m, n, fill=uniform_pm1, rng=rng)
    base = ColMajor(m, n)
    O.gemm("T", "T", m, n, k, 1.0, a.view, b.view, 0.0, base.view, S)
    out = ColMajor(m, n)
    out.buf[...] = c0.buf
    O.gemm("T", "T", m, n, k, -0.75, a.view, b.view, 2.0, out.view, S)
    np.testing.assert_allclose(out.view, -0.75 * base.view + 2.0 * c0.view, rtol=0, atol=4e-16 * 40)
    # beta == 0 must not read C (src/gemm.cu:143-147): NaN in C does not propagate
    out2 = ColMajor(m, n)
    out2.buf[...] = np.nan
    O.gemm("T", "T", m, n, k, 1.0, a.view, b.view, 0.0, out2.view, S)
    np.testing.assert_array_equal(out2.view, base.view)


# ------------------------------------------------------------------ residuals: the reference's one pinned number

def test_residual_curve_matches_survey():
    """SURVEY §6 / BASELINE.md §2 (numpy emulation of the reference): ~2^-7 per extra slice down to ~3e-16"""
    rng = np.random.default_rng(0)
    n = 96
    a = operand("N", n, n, rng)
    b = operand("N", n, n, rng)
    res = {}
    for S in (3, 6, 7, 8, 9, 10):
        c = ColMajor(n, n)
        O.gemm("N", "N", n, n, n, 1.0, a.view, b.view, 0.0, c.view, S)
        res[S] = O.relative_residual("N", "N", n, n, n, a.view, b.view, c.view)
    assert 1e-6 < res[3] < 2e-5 and 1e-12 < res[6] < 2e-11 and 1e-14 < res[7] < 2e-13
    assert res[8] < 1e-15 and res[9] < 6e-16 and res[10] < 6e-16
    assert 50 < res[6] / res[7] < 250


def test_baseline_config_c1_on_the_cpu_oracle():
    """BASELINE.json configs[0] verbatim: fp64_int8_6, M=N=K=512, U[-1,1), split + INT8 GEMM + accumulate on the CPU.
    Both summation orders; the residual sits where six 7-bit slices put it (~1e-11), and the diagonal
    grouping (what the HIP kernel computes, checked bit for bit in tests/test_gpu_robustness.py) is no worse."""
    m = n = k = 512
    rng = np.random.default_rng(0)
    a = operand("N", m, k, rng)
    b = operand("N", k, n, rng)
    res = {}
    for order in (O.ORDER_REFERENCE, O.ORDER_DIAGONAL):
        c = ColMajor(m, n)
        assert O.gemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, c.view, 6, order) == 0
        res[order] = O.relative_residual_sampled("N", "N", m, n, k, a.view, b.view, c.view, ns=4096)
    assert 1e-12 < res[O.ORDER_REFERENCE] < 1e-10, res
    assert res[O.ORDER_DIAGONAL] <= 1.01 * res[O.ORDER_REFERENCE]


@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
def test_reference_ci_gate_on_cpu(op_a, op_b):
    """test/main_test.cu:702-746 at full size for S=8 (the weakest gated mode), both summation orders"""
    m, n, k = (1023, 1025, 1024) if op_a == "N" else (1025, 1024, 1023)
    rng = np.random.default_rng(0)
    a = operand(op_a, m, k, rng, fill=uniform01)
    b = operand(op_b, k, n, rng, fill=uniform01)
    for order in (O.ORDER_REFERENCE, O.ORDER_DIAGONAL):
        c = ColMajor(m, n)
        assert O.gemm(op_a, op_b, m, n, k, 1.0, a.view, b.view, 0.0, c.view, 8, order) == 0
        r = O.relative_residual_sampled(op_a, op_b, m, n, k, a.view, b.view, c.view, ns=8192)
        assert r < 1e-15, (order, r)


@pytest.mark.parametrize("S", [9, 12, 16])
def test_reference_ci_gate_small_shapes_all_modes(S):
    rng = np.random.default_rng(S)
    m, n, k = 127, 130, 129
    for op_a in "NT":
        for op_b in "NT":
            a = operand(op_a, m, k, rng, fill=uniform01)
            b = operand(op_b, k, n, rng, fill=uniform01)
            c = ColMajor(m, n)
            O.gemm(op_a, op_b, m, n, k, 1.0, a.view, b.view, 0.0, c.view, S)
            assert O.relative_residual(op_a, op_b, m, n, k, a.view, b.view, c.view) < 1e-15


def test_diagonal_order_is_a_regrouping_of_the_reference_sum():
    """same exact terms, different rounding points: results agree to a few ulp of the largest partial sum,
    and K-chunking of the diagonal order changes only the grouping again"""
    rng = np.random.default_rng(9)
    m, n, k, S = 40, 30, 300, 10
    a = operand("N", m, k, rng, fill=exp_rand(2.0))
    b = operand("T", k, n, rng, fill=exp_rand(2.0))
    c = [ColMajor(m, n) for _ in range(3)]
    O.gemm("N", "T", m, n, k, 1.0, a.view, b.view, 0.0, c[0].view, S, O.ORDER_REFERENCE)
    O.gemm("N", "T", m, n, k, 1.0, a.view, b.view, 0.0, c[1].view, S, O.ORDER_DIAGONAL)
    O.gemm("N", "T", m, n, k, 1.0, a.view, b.view, 0.0, c[2].view, S, O.ORDER_DIAGONAL, kchunk=64)
    truth_scale = (np.abs(a.view) @ np.abs(b.view.T))
    for x in c[1:]:
        assert (np.abs(x.view - c[0].view) <= 8 * 2.0 ** -52 * truth_scale).all()
    r = [O.relative_residual("N", "T", m, n, k, a.view, b.view, x.view) for x in c]
    assert max(r) < 1e-15


# ------------------------------------------------------------------ auto mode (row A11)

def test_auto_mode_statistic_and_selection():
    rng = np.random.default_rng(4)
    m, n, k = 64, 48, 200
    L = O.bits_per_int8(k)
    a = operand("N", m, k, rng)
    b = operand("N", k, n, rng)
    a.view[2, 3] = 0.0
    s, cnt = O.auto_select("N", "N", m, n, k, a.view, b.view, 1.5)
    # independent numpy restatement of src/split.cu:317-350
    expect = np.zeros(16, dtype=np.uint64)
    for mat, axis in ((a.view, 1), (b.view.T, 1)):
        e = np.frexp(np.abs(mat).max(axis=axis))[1]          # 2^(e-1) <= max < 2^e ; max_exp = 2^e
        ex = np.frexp(np.abs(mat))[1]
        req = (e[:, None] - ex + 1) + 53                      # (e_max+1 - e_x) + 53; frexp exponents = IEEE + 1
        req = np.where(mat == 0, 0, req)
        for S in range(3, 19):
            expect[S - 3] += np.maximum(0, req - S * L)[mat != 0].sum()
    np.testing.assert_array_equal(cnt, expect)
    assert s == next(S for S in range(3, 19) if cnt[S - 3] / (m * k + k * n) <= 1.5)
    assert s == 8                                             # uniform inputs -> fp64_int8_8
    # threshold 0 (the handle's default, src/handle.hpp:26): first S with no loss at all
    assert O.auto_select("N", "N", m, n, k, a.view, b.view, 0.0)[0] == next(S for S in range(3, 19) if cnt[S - 3] == 0)
    wide = operand("N", m, k, rng, fill=wide_exponent(8))
    assert O.auto_select("N", "N", m, n, k, wide.view, b.view, 1.5)[0] >= 9
    huge = operand("N", m, k, rng, fill=wide_exponent(60))
    assert O.auto_select("N", "N", m, n, k, huge.view, b.view, 0.0)[0] == 0   # -> dgemm


# ------------------------------------------------------------------ committed fixtures

@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    op_a, op_b = str(g["op_a"]), str(g["op_b"])
    m, n, k, S, L = (int(g[x]) for x in ("m", "n", "k", "S", "L"))
    assert O.bits_per_int8(k) == L
    pa, ea = O.split("A", op_a, g["a"], S, L)
    pb, eb = O.split("B", op_b, g["b"], S, L)
    np.testing.assert_array_equal(pa, g["planes_a"])
    np.testing.assert_array_equal(pb, g["planes_b"])
    np.testing.assert_array_equal(ea.view(np.uint64), g["max_exp_a"].view(np.uint64))
    np.testing.assert_array_equal(eb.view(np.uint64), g["max_exp_b"].view(np.uint64))
    np.testing.assert_array_equal(O.diagonal_sums(pa, pb), g["diag"])
    for order, key in ((O.ORDER_REFERENCE, "c_reference_order"), (O.ORDER_DIAGONAL, "c_diagonal_order")):
        c = np.asfortranarray(g["c0"].copy())
        assert O.gemm(op_a, op_b, m, n, k, float(g["alpha"]), g["a"], g["b"], float(g["beta"]), c, S, order) == 0
        np.testing.assert_array_equal(c.view(np.uint64), g[key].view(np.uint64))
    s, cnt = O.auto_select(op_a, op_b, m, n, k, g["a"], g["b"], 1.5)
    assert s == int(g["auto_selected"])
    np.testing.assert_array_equal(cnt, g["auto_counters"])


def test_golden_fixtures_exist():
    assert len(GOLDEN) >= 4


# ------------------------------------------------------------------ complex path (row F1)

def _zrand(rng, r, c, kind="pm1"):
    if kind == "urand01":
        x = (1.0 - rng.uniform(0, 1, (r, c))) + 1j * (1.0 - rng.uniform(0, 1, (r, c)))
    else:
        x = rng.uniform(-1, 1, (r, c)) + 1j * rng.uniform(-1, 1, (r, c))
    return np.asfortranarray(x)


@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
@pytest.mark.parametrize("order", [O.ORDER_REFERENCE, O.ORDER_DIAGONAL])
def test_zgemm_matches_complex_matmul(op_a, op_b, order):
    """gemm_int8<cuDoubleComplex> (src/gemm.cu:412-521): four real products with -alpha, alpha, i*alpha, i*alpha"""
    rng = np.random.default_rng(2)
    m, n, k, S = 45, 38, 77, 10
    a = _zrand(rng, *((m, k) if op_a == "N" else (k, m)))
    b = _zrand(rng, *((k, n) if op_b == "N" else (n, k)))
    c0 = _zrand(rng, m, n)
    alpha, beta = 0.5 + 1.5j, -1.25 + 0.25j
    c = c0.copy(order="F")
    assert O.zgemm(op_a, op_b, m, n, k, alpha, a, b, beta, c, S, order) == 0
    ref = alpha * ((a if op_a == "N" else a.T) @ (b if op_b == "N" else b.T)) + beta * c0
    assert np.abs(c - ref).max() <= 64 * 2.0 ** -52 * np.abs(ref).max()
    # beta == 0: C is not read (init_c_complex_kernel<true>, src/gemm.cu:214-215)
    c2 = np.full((m, n), np.nan + 0j, order="F")
    assert O.zgemm(op_a, op_b, m, n, k, 1.0, a, b, 0.0, c2, S, order) == 0
    assert np.isfinite(c2.real).all() and np.isfinite(c2.imag).all()
    assert O.relative_residual_sampled_z(op_a, op_b, m, n, k, a, b, c2, ns=400) < 1e-15


@pytest.mark.parametrize("S", [8, 11, 16])
def test_reference_ci_gate_complex_on_cpu(S):
    """the complex half of test/main_test.cu:702-746 (sizes reduced for the CPU suite)"""
    rng = np.random.default_rng(S)
    m, n, k = 255, 257, 256
    for op_a in "NT":
        for op_b in "NT":
            a = _zrand(rng, *((m, k) if op_a == "N" else (k, m)), kind="urand01")
            b = _zrand(rng, *((k, n) if op_b == "N" else (n, k)), kind="urand01")
            c = np.zeros((m, n), dtype=np.complex128, order="F")
            assert O.zgemm(op_a, op_b, m, n, k, 1.0, a, b, 0.0, c, S) == 0
            assert O.relative_residual_sampled_z(op_a, op_b, m, n, k, a, b, c, ns=2000) < 1e-15


def test_complex_auto_mode_counts_both_parts():
    rng = np.random.default_rng(8)
    m, n, k = 40, 30, 120
    a, b = _zrand(rng, m, k), _zrand(rng, k, n)
    s, cnt = O.auto_select_z("N", "N", m, n, k, a, b, 1.5)
    _, cre = O.auto_select("N", "N", m, n, k, np.asfortranarray(a.real), np.asfortranarray(b.real), 1.5)
    _, cim = O.auto_select("N", "N", m, n, k, np.asfortranarray(a.imag), np.asfortranarray(b.imag), 1.5)
    np.testing.assert_array_equal(cnt, cre + cim)          # src/split.cu:367-374
    assert s == next(S for S in range(3, 19) if cnt[S - 3] / (m * k + k * n) <= 1.5)


@pytest.mark.parametrize("op_a,op_b", [("C", "N"), ("N", "C"), ("C", "C"), ("C", "T")])
def test_oracle_zgemm_conjugate_transpose(op_a, op_b):
    """OZ_OP_C: the operand's imaginary part negated, then the reference's op_t path (oracle/ozaki_oracle.h; the reference's
    hook runs CUBLAS_OP_C as a plain transpose, src/cublas.cu:50-56).  Against numpy's conjugate transpose, and bit-identical
    to handing the oracle an explicitly conjugated operand with op T."""
    m, n, k, S = 37, 29, 130, 10
    rng = np.random.default_rng(5)

    def z(rows, cols):
        return np.asfortranarray(rng.uniform(-1, 1, (rows, cols)) + 1j * rng.uniform(-1, 1, (rows, cols)))
    a = z(m, k) if op_a == "N" else z(k, m)
    b = z(k, n) if op_b == "N" else z(n, k)
    c0 = z(m, n)
    alpha, beta = 0.75 + 0.5j, -1.5 + 0.25j
    c = np.array(c0, order="F")
    assert O.zgemm(op_a, op_b, m, n, k, alpha, a, b, beta, c, S, O.ORDER_REFERENCE) == 0
    opm = {"N": lambda x: x, "T": lambda x: x.T, "C": lambda x: x.conj().T}
    want = alpha * (opm[op_a](a) @ opm[op_b](b)) + beta * c0
    assert np.abs(c - want).max() / np.abs(want).max() < 1e-13
    c2 = np.array(c0, order="F")
    a2 = np.asfortranarray(a.conj()) if op_a == "C" else a
    b2 = np.asfortranarray(b.conj()) if op_b == "C" else b
    assert O.zgemm(op_a.replace("C", "T"), op_b.replace("C", "T"), m, n, k, alpha, a2, b2, beta, c2, S, O.ORDER_REFERENCE) == 0
    np.testing.assert_array_equal(c.T.copy().view(np.uint64), c2.T.copy().view(np.uint64))
