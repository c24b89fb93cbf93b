"""GPU parity tests of the complex path: gemm_int8<cuDoubleComplex>, src/gemm.cu:412-521 --
Re/Im split separately, C scaled by beta, then four real Ozaki products (Im,Im) (Re,Re) (Im,Re) (Re,Im)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import ozimmu_amd
from oracle import oracle as O
from tests.util import ColMajor

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["resident_split", "resident_split_16", "resident_split_32", "multi_view_split", "banded_split", "one_pass_split"])
def split_kernel(request, monkeypatch):
    """every test runs with every split implementation: the resident one-read kernel (what the library does for K <= 2048;
    strips of 8 rows by policy at these sizes, and forced to 16 and 32), one launch per pass for all operand views (what it does
    for longer K), one launch per view walking row bands (operands larger than the Infinity Cache; the band size is forced
    down so that these shapes have several bands), and the one-pass kernel that re-reads its strip from L2"""
    if request.param in ("resident_split_16", "resident_split_32"):
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_RESIDENT", request.param[-2:])
    elif request.param != "resident_split":
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_RESIDENT", "0")
    if request.param == "banded_split":
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_MULTI_BYTES", "0")
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_BAND_BYTES", "65536")
    elif request.param == "one_pass_split":
        monkeypatch.setenv("OZIMMU_HIP_SPLIT_ONE_PASS_BYTES", str(1 << 40))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def zfill(kind):
    def f(rng, shape):
        if kind == "urand01":  # the reference's complex ci_test input: both parts uniform (0,1]
            return (1.0 - rng.uniform(0, 1, shape)) + 1j * (1.0 - rng.uniform(0, 1, shape))
        if kind == "wide":
            return (rng.uniform(-1, 1, shape) * 10.0 ** (6 * rng.uniform(0, 1, shape))
                    + 1j * rng.uniform(-1, 1, shape) * 10.0 ** (6 * rng.uniform(0, 1, shape)))
        return rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)
    return f


def zoperand(op, rows, cols, rng, kind="pm1", pad=0):
    r, c = (rows, cols) if op == "N" else (cols, rows)
    x = ColMajor(r, c, ld=r + pad, dtype=np.complex128)
    x.buf[:, :r] = zfill(kind)(rng, (c, r))
    if pad:
        x.buf[:, r:] = np.nan
    return x


def _sync():
    import torch
    torch.cuda.synchronize()


def zbits(x):
    """bit patterns of a complex array: (2, ...) uint64"""
    return np.stack([np.ascontiguousarray(x.real).view(np.uint64), np.ascontiguousarray(x.imag).view(np.uint64)])


# OZIMMU_OP_C: computed as the conjugate transpose (the reference's hook maps CUBLAS_OP_C to op_t, src/cublas.cu:50-56; DESIGN.md 2
# deviation 10) - every combination with N / T / C, through the one-launch form of the four real products and through four launches
@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T"), ("C", "N"), ("N", "C"), ("C", "C"),
                                       ("C", "T"), ("T", "C")])
@pytest.mark.parametrize("m,n,k,S", [(64, 64, 64, 6), (1, 1, 1, 9), (70, 33, 129, 9), (130, 65, 257, 13)])
def test_zgemm_bit_exact_vs_oracle(oz, monkeypatch, op_a, op_b, m, n, k, S, fused):
    m_, h = oz
    monkeypatch.setenv("OZIMMU_HIP_FUSED_PRODUCTS", fused)
    rng = np.random.default_rng(m + n + k + S)
    a = zoperand(op_a, m, k, rng, "wide", pad=1)
    b = zoperand(op_b, k, n, rng, "wide", pad=2)
    c = zoperand("N", m, n, rng, pad=3)
    c_ref = ColMajor(m, n, ld=m + 3, dtype=np.complex128)
    c_ref.buf[...] = c.buf
    c0 = np.array(c.view)
    alpha, beta = 1.25 - 0.5j, -0.75 + 2.0j
    st = m_.gemm(h, op_a, op_b, m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c.dev, c.ld, f"fp64_int8_{S}", m_.complx)
    _sync()
    assert st == 0
    a.buf[:, a.rows:] = 0   # (the oracle's conjugated copy walks the padding too)
    b.buf[:, b.rows:] = 0
    assert O.zgemm(op_a, op_b, m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    got = np.array(c.download())
    np.testing.assert_array_equal(zbits(got), zbits(c_ref.view))
    assert np.isnan(c.buf[:, m:]).all()  # ld padding untouched
    # ... and it IS the conjugate transpose: against numpy on the same data
    opm = {"N": lambda x: x, "T": lambda x: x.T, "C": lambda x: x.conj().T}
    want = alpha * (opm[op_a](a.view) @ opm[op_b](b.view)) + beta * c0
    if S >= 9:
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-11


def test_zgemm_k_at_the_pass_boundary(oz):
    """the complex twin of test_gpu_parity.py::test_gemm_k_at_the_pass_boundary: S = 6, K = 22176 holds 694 k-blocks against a
    pass length of 692 - two chained passes per real product, through the FP64 workspace"""
    m_, h = oz
    m, n, k, S = 24, 40, 22176, 6
    rng = np.random.default_rng(k)
    a = zoperand("N", m, k, rng)
    b = zoperand("T", k, n, rng)
    c = zoperand("N", m, n, rng)
    c_ref = ColMajor(m, n, dtype=np.complex128)
    c_ref.buf[...] = c.buf
    alpha, beta = 0.5 + 1.5j, 1.0 - 0.25j
    assert m_.gemm(h, "N", "T", m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c.dev, c.ld, "fp64_int8_6", m_.complx) == 0
    _sync()
    kchunk = (2147483647 // (S * 127 * 127)) // 64 * 64
    assert O.zgemm("N", "T", m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL, kchunk=kchunk) == 0
    np.testing.assert_array_equal(zbits(c.download()), zbits(c_ref.view))


@pytest.mark.parametrize("op_a,op_b", [("C", "N"), ("N", "C"), ("C", "C")])
def test_zgemm_conjugate_transpose_across_k_passes(oz, op_a, op_b):
    """OP_C with two chained K passes per real product (S = 6, K = 22176: 694 k-blocks against a pass length of 692): the sign
    flip of the conjugated operand's imaginary products is applied in every pass"""
    m_, h = oz
    m, n, k, S = 24, 40, 22176, 6
    rng = np.random.default_rng(k + ord(op_a) + 2 * ord(op_b))
    a = zoperand(op_a, m, k, rng)
    b = zoperand(op_b, k, n, rng)
    c = zoperand("N", m, n, rng)
    c_ref = ColMajor(m, n, dtype=np.complex128)
    c_ref.buf[...] = c.buf
    alpha, beta = 0.5 + 1.5j, 1.0 - 0.25j
    assert m_.gemm(h, op_a, op_b, m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c.dev, c.ld, "fp64_int8_6", m_.complx) == 0
    _sync()
    kchunk = (2147483647 // (S * 127 * 127)) // 64 * 64
    assert O.zgemm(op_a, op_b, m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL, kchunk=kchunk) == 0
    np.testing.assert_array_equal(zbits(c.download()), zbits(c_ref.view))


@pytest.mark.parametrize("op_a,op_b", [("C", "N"), ("N", "C"), ("C", "C"), ("T", "C")])
def test_zgemm_strided_batched_conjugate_transpose(oz, op_a, op_b):
    """a strided batch with OP_C: every matrix bit-identical to the oracle's zgemm and to numpy's conjugate transpose"""
    import torch
    m_, h = oz
    m, n, k, S, batch = 70, 66, 100, 9, 3
    rng = np.random.default_rng(17 + ord(op_a) + 2 * ord(op_b))
    ra, ca = (m, k) if op_a == "N" else (k, m)
    rb, cb_ = (k, n) if op_b == "N" else (n, k)
    A = [zoperand("N", ra, ca, rng) for _ in range(batch)]      # storage as given (op applied by the call)
    B = [zoperand("N", rb, cb_, rng) for _ in range(batch)]
    C0 = [zoperand("N", m, n, rng) for _ in range(batch)]
    a = torch.from_numpy(np.stack([x.buf for x in A])).cuda()    # (batch, cols, ld) = column-major matrices back to back
    b = torch.from_numpy(np.stack([x.buf for x in B])).cuda()
    c = torch.from_numpy(np.stack([x.buf for x in C0])).cuda()
    alpha, beta = 0.5 - 1j, 0.25 + 2j
    assert m_.gemm_strided_batched(h, torch.cuda.current_stream(), op_a, op_b, m, n, k, alpha, a, ra, ra * ca, b, rb, rb * cb_,
                                   beta, c, m, n * m, batch, "fp64_int8_9", m_.complx) == 0
    _sync()
    got = c.cpu().numpy()
    opm = {"N": lambda x: x, "T": lambda x: x.T, "C": lambda x: x.conj().T}
    for i in range(batch):
        c_ref = ColMajor(m, n, dtype=np.complex128)
        c_ref.buf[...] = C0[i].buf
        assert O.zgemm(op_a, op_b, m, n, k, alpha, A[i].view, B[i].view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
        np.testing.assert_array_equal(zbits(got[i].T), zbits(c_ref.view))
        want = alpha * (opm[op_a](A[i].view) @ opm[op_b](B[i].view)) + beta * C0[i].view
        assert np.abs(got[i].T - want).max() / np.abs(want).max() < 1e-11


def test_zgemm_beta_zero_does_not_read_c_and_real_alpha(oz):
    m_, h = oz
    m, n, k, S = 96, 80, 200, 9
    rng = np.random.default_rng(1)
    a = zoperand("N", m, k, rng)
    b = zoperand("N", k, n, rng)
    c = ColMajor(m, n, dtype=np.complex128)
    c.buf[...] = np.nan
    c_ref = ColMajor(m, n, dtype=np.complex128)
    assert m_.gemm(h, "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld, "fp64_int8_9", m_.complx) == 0
    _sync()
    O.zgemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL)
    np.testing.assert_array_equal(zbits(c.download()), zbits(c_ref.view))
    r = O.relative_residual_sampled_z("N", "N", m, n, k, a.view, b.view, c.view, ns=500)
    assert r < 1e-15


@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
@pytest.mark.parametrize("S", [8, 12, 16])
def test_reference_ci_gate_complex(oz, op_a, op_b, S):
    """the complex half of test/main_test.cu:702-746: residual < 1e-15 for fp64_int8_8..16"""
    m_, h = oz
    m, n, k = 1023, 1025, 1024
    rng = np.random.default_rng(0)
    a = zoperand(op_a, m, k, rng, "urand01")
    b = zoperand(op_b, k, n, rng, "urand01")
    c = ColMajor(m, n, dtype=np.complex128)
    assert m_.gemm(h, op_a, op_b, m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld, f"fp64_int8_{S}",
                   m_.complx) == 0
    _sync()
    r = O.relative_residual_sampled_z(op_a, op_b, m, n, k, a.view, b.view, c.download(), ns=2048)
    assert r < 1e-15, r


def test_zgemm_auto_mode(oz):
    m_, h = oz
    m, n, k = 100, 90, 260
    rng = np.random.default_rng(3)
    a = zoperand("T", m, k, rng, "wide")
    b = zoperand("N", k, n, rng)
    s_ref, _ = O.auto_select_z("T", "N", m, n, k, a.view, b.view, 1.5)
    mode = m_.auto_mode_select(h, "T", "N", m, n, k, a.dev, a.ld, b.dev, b.ld, m_.complx, 1.5)
    assert mode == (m_.dgemm if s_ref == 0 else m_.fp64_int8_3 + s_ref - 3)
    # `dgemm` mode with complex operands goes to the vendor ZGEMM (src/gemm.cu:639-645)
    c = ColMajor(m, n, dtype=np.complex128)
    assert m_.gemm(h, "T", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld, "dgemm", m_.complx) == 0
    _sync()
    assert O.relative_residual_sampled_z("T", "N", m, n, k, a.view, b.view, c.download(), ns=300) < 1e-14


def test_pytorch_complex128_matmul_is_intercepted():
    """torch.mm(complex128) -> hipblasZgemm -> rocblas_zgemm -> the shim (cublasZgemm_v2, src/cublas.cu:297-313)"""
    code = textwrap.dedent("""
        import torch
        torch.manual_seed(0)
        a = torch.rand(1100, 1024, dtype=torch.float64, device="cuda") + 1j * torch.rand(1100, 1024, dtype=torch.float64, device="cuda")
        b = torch.rand(1024, 1050, dtype=torch.float64, device="cuda") - 1j * torch.rand(1024, 1050, dtype=torch.float64, device="cuda")
        c = a @ b
        torch.cuda.synchronize()
        ref = a.cpu() @ b.cpu()
        print("MAXDIFF %.3e" % ((c.cpu() - ref).abs().max() / ref.abs().max()).item())
    """)
    out = {}
    for mode in ("fp64_int8_3", "fp64_int8_10"):
        e = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
        e.update(LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE=mode, OZIMMU_ENABLE_CULIP_PROFILING="1")
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stdout + p.stderr
        out[mode] = (float([l for l in p.stdout.splitlines() if l.startswith("MAXDIFF")][0].split()[1]), p.stdout)
    assert out["fp64_int8_3"][0] > 1e-7 and out["fp64_int8_10"][0] < 1e-13, (out["fp64_int8_3"][0], out["fp64_int8_10"][0])
    assert "[CULiP Result][Zfp64_int8_10-" in out["fp64_int8_10"][1]


@pytest.mark.parametrize("seed", range(int(os.environ.get("OZIMMU_TEST_FUZZ_SEEDS", "2"))))
def test_zgemm_fuzz_forced_kernels_bit_exact(oz, monkeypatch, seed):
    """complex fuzz: random shapes / ops / slice counts / complex alpha and beta with the slice-GEMM kernel forced at random
    (the four real products run fused in one launch, one by one on the wide or the classic kernel, in one or two diagonal
    passes): bit-exact against the oracle's ZGEMM.  OZIMMU_TEST_FUZZ_SEEDS=n runs a longer campaign."""
    m_, h = oz
    rng = np.random.default_rng(88000 + seed)
    for case in range(6):
        m, n = int(rng.integers(1, 420)), int(rng.integers(1, 420))
        k = int(rng.choice([1, 33, 64, 130, 257]))
        S = int(rng.choice([3, 6, 8, 9, 11, 13, 15, 18]))
        kernel = rng.choice(["", "wide", "classic", "k2"])
        if kernel:
            monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", str(kernel))
        else:
            monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
        op_a, op_b = rng.choice(["N", "T"]), rng.choice(["N", "T"])
        a = zoperand(op_a, m, k, rng, "wide", pad=int(rng.integers(0, 3)))
        b = zoperand(op_b, k, n, rng, "pm1", pad=int(rng.integers(0, 3)))
        pad = int(rng.integers(0, 3))
        c = zoperand("N", m, n, rng, pad=pad)
        c_ref = ColMajor(m, n, ld=m + pad, dtype=np.complex128)
        c_ref.buf[...] = c.buf
        alpha = complex(rng.choice([1.0, -0.5, 1.25 - 0.5j]))
        beta = complex(rng.choice([0.0, 1.0, -0.75 + 2.0j]))
        assert m_.gemm(h, op_a, op_b, m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c.dev, c.ld, f"fp64_int8_{S}",
                       m_.complx) == 0
        _sync()
        assert O.zgemm(op_a, op_b, m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
        got = c.download()
        assert np.array_equal(zbits(got), zbits(c_ref.view)), (seed, case, kernel, op_a, op_b, m, n, k, S, alpha, beta)


@pytest.mark.parametrize("S,kernel", [(9, None), (6, None), (12, None), (9, "wide"), (8, "x16")])
def test_zgemm_one_persistent_launch_equals_four_launches_bitwise(oz, monkeypatch, S, kernel):
    """Problems that fill the chip run the four real products of a ZGEMM as ONE launch of the persistent wide kernel: every
    claimed tile is walked through (Im,Im), (Re,Re), (Im,Re), (Re,Im) in the reference's order (src/gemm.cu:479-518) before
    the next tile is claimed, so each element of C sees the same sequence of updates as with one launch per product
    (OZIMMU_HIP_FUSED_PRODUCTS=0, the form the oracle tests above pin).  2304 x 1536 x 1024: mixed tile heights, more
    tiles than CUs, phase hints and claim counters in use; both tile functions; alpha and beta complex."""
    import torch
    m_, h = oz
    if kernel:
        monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", kernel)
    m, n, k = 2304, 1536, 1024
    g = torch.Generator(device="cuda").manual_seed(7 + S)
    def z(*shape):
        return torch.complex(torch.rand(*shape, dtype=torch.float64, device="cuda", generator=g) * 2 - 1,
                             torch.rand(*shape, dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
    a, b, c0 = z(k, m), z(n, k), z(n, m)
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("OZIMMU_HIP_FUSED_PRODUCTS", fused)
        c = c0.clone()
        assert m_.gemm(h, "N", "N", m, n, k, 0.5 - 1.5j, a, m, b, k, 0.25 + 0.75j, c, m, f"fp64_int8_{S}", m_.complx) == 0
        _sync()
        out[fused] = torch.view_as_real(c).view(torch.int64)
    assert torch.equal(out["1"], out["0"])
    ref = (b @ a) * (0.5 - 1.5j) + (0.25 + 0.75j) * c0          # row-major view of the column-major product
    got = torch.view_as_complex(out["1"].view(torch.float64))
    tol = {6: 1e-8, 8: 1e-12}.get(S, 1e-13)
    assert ((got - ref).norm() / ref.norm()).item() < tol
