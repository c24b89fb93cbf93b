"""GPU parity tests of the K-split slice-GEMM kernel (slice_gemm_k2_kernel.h: 64x64 tiles, eight waves, the two wave
groups accumulate the two halves of the k-blocks and add their INT32 sums through LDS).

The library picks it when a launch has no more 64x64 tiles than the device has CUs (the oracle-sized shapes of the
other parity files already run through it by default); OZIMMU_HIP_GEMM_KERNEL=k2 forces it for every S it is built
for (S <= 9 in one pass, and the first diagonal pass of S >= 13).  Cases: one k-block (group 1 idle), odd and even
numbers of k-blocks (the groups' barrier rounds differ by one), ragged edges, K chunking, batches.  Same bar as
test_gpu_parity.py: INT32 diagonal sums and the FP64 result bit-exact vs the oracle (OZ_ORDER_DIAGONAL)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import ColMajor, exp_rand, operand, uniform_pm1, wide_exponent

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def force_k2(monkeypatch):
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", "k2")


def _sync():
    import torch
    torch.cuda.synchronize()


@pytest.mark.parametrize("S", list(range(3, 19)))
@pytest.mark.parametrize("m,n,k", [(97, 129, 65), (130, 70, 200), (64, 64, 32)])
def test_k2_diagonal_sums_bit_exact(ozh, S, m, n, k):
    import torch
    m_, h = ozh   # the INT32 dump is a test hook: libozimmu_hip_test.so
    rng = np.random.default_rng(m * 7 + n * 3 + k + S)
    a = operand("N", m, k, rng, fill=exp_rand(2.0))
    b = operand("T", k, n, rng, fill=exp_rand(2.0))
    L = O.bits_per_int8(k)
    pa, _ = O.split("A", "N", a.view, S, L)
    pb, _ = O.split("B", "T", b.view, S, L)
    d_ref = O.diagonal_sums(pa, pb)  # [S][m][n] int64
    out = torch.full((S, n, m), 12345, dtype=torch.int32, device="cuda")
    assert m_.diagonal_sums(h, "N", "T", m, n, k, a.dev, a.ld, b.dev, b.ld, S, out) == 0
    _sync()
    np.testing.assert_array_equal(out.cpu().numpy().transpose(0, 2, 1).astype(np.int64), d_ref)


@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (64, 64, 32), (64, 64, 64), (97, 129, 33), (97, 129, 96), (200, 130, 161),
                                   (389, 257, 130)])
@pytest.mark.parametrize("S", [3, 6, 9, 13, 18])
def test_k2_gemm_bit_exact_vs_oracle(oz, op_a, op_b, m, n, k, S):
    m_, h = oz
    rng = np.random.default_rng(m + 2 * n + 3 * k + S)
    a = operand(op_a, m, k, rng, pad=1)
    b = operand(op_b, k, n, rng, pad=2)
    c = ColMajor(m, n, ld=m + 3, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n, ld=m + 3)
    c_ref.buf[...] = c.buf
    st = m_.gemm(h, op_a, op_b, m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld, f"fp64_int8_{S}")
    _sync()
    assert st == 0
    assert O.gemm(op_a, op_b, m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))
    assert np.isnan(c.buf[:, m:]).all()  # ld padding untouched


@pytest.mark.parametrize("alpha,beta", [(-2.5, 0.0), (1.0, 1.0), (0.75, -1.25)])
def test_k2_gemm_alpha_beta(oz, alpha, beta):
    m_, h = oz
    m, n, k, S = 230, 190, 160, 9
    rng = np.random.default_rng(5)
    a = operand("N", m, k, rng, fill=wide_exponent(4))
    b = operand("T", k, n, rng)
    c = ColMajor(m, n, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n)
    c_ref.buf[...] = c.buf
    assert m_.gemm(h, "N", "T", m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c.dev, c.ld, "fp64_int8_9") == 0
    _sync()
    assert O.gemm("N", "T", m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


@pytest.mark.parametrize("S", [9, 14])
def test_k2_gemm_k_chunking(oz, S):
    """K above the INT32-safe pass length: every chunk is split between the two wave groups again"""
    m_, h = oz
    m, n, k = 100, 130, 20000
    rng = np.random.default_rng(3 + S)
    a = operand("N", m, k, rng)
    b = operand("N", k, n, rng)
    c = ColMajor(m, n)
    c_ref = ColMajor(m, n)
    assert m_.gemm(h, "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld, f"fp64_int8_{S}") == 0
    _sync()
    kchunk = (2147483647 // (S * 127 * 127)) // 64 * 64   # the library keeps a pass at an even number of 32-k blocks
    O.gemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL, kchunk=kchunk)
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


def test_k2_equals_classic_bitwise_and_is_the_default_for_1024(oz, monkeypatch):
    """1024 x 1024 x 1024 (256 tiles on 256 CUs): the default choice and both forced kernels give the same bits"""
    import torch
    m_, h = oz
    n, S = 1024, 9
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.rand(n, n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    b = torch.rand(n, n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    outs = []
    for kernel in ("k2", "classic", None):
        if kernel is None:
            monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL")
        else:
            monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", kernel)
        c = torch.zeros(n, n, dtype=torch.float64, device="cuda")
        assert m_.gemm(h, "N", "T", n, n, n, 1.0, a, n, b, n, 0.0, c, n, f"fp64_int8_{S}") == 0
        _sync()
        outs.append(c)
    assert torch.equal(outs[0].view(torch.int64), outs[1].view(torch.int64))
    assert torch.equal(outs[0].view(torch.int64), outs[2].view(torch.int64))


def test_k2_strided_batch(oz):
    """a batch of small matrices in one launch: blockIdx.y = matrix, every matrix K-split inside its workgroups"""
    import torch
    m_, h = oz
    m, n, k, S, batch = 70, 90, 130, 8, 3
    rng = np.random.default_rng(23)
    A = [operand("N", m, k, rng) for _ in range(batch)]
    B = [operand("N", k, n, rng) for _ in range(batch)]
    a = torch.stack([x.dev for x in A]).contiguous()
    b = torch.stack([x.dev for x in B]).contiguous()
    c = torch.zeros(batch, n, m, dtype=torch.float64, device="cuda")
    st = m_.gemm_strided_batched(h, torch.cuda.current_stream(), "N", "N", m, n, k, 1.0, a, A[0].ld, a.stride(0), b,
                                 B[0].ld, b.stride(0), 0.0, c, m, c.stride(0), batch, f"fp64_int8_{S}")
    _sync()
    assert st == 0
    for i in range(batch):
        c_ref = ColMajor(m, n)
        assert O.gemm("N", "N", m, n, k, 1.0, A[i].view, B[i].view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL) == 0
        np.testing.assert_array_equal(c[i].cpu().numpy().view(np.uint64), c_ref.buf.view(np.uint64))
