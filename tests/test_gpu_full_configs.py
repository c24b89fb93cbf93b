"""GPU tests at BASELINE.json's full sizes (configs C2..C5), through size-independent properties:

* sampled relative residual against a long-double product (mateval's metric, test/main_test.cu:101-117);
* exact scaling: op(A) scaled by a power of two per row (B per column) scales C exactly -- the slices of a row
  do not depend on its exponent, so the results must agree BITWISE after rescaling;
* row equivariance: permuting rows of op(A) permutes rows of C bitwise (rows are cut independently and the
  INT8 products are exact);
* accuracy versus slice count: the residual falls ~2^-7 per slice until FP64 rounding saturates it
  (BASELINE.md §2), and the auto mode picks the slice count the oracle picks.
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _dev_rand(shape, seed, lo=-1.0, hi=1.0):
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.rand(shape, dtype=torch.float64, device="cuda", generator=g) * (hi - lo) + lo


def _gemm(oz, h, op_a, op_b, m, n, k, A, B, C, mode, alpha=1.0, beta=0.0):
    import torch
    lda, ldb = A.shape[1], B.shape[1]
    st = oz.gemm(h, op_a, op_b, m, n, k, alpha, A, lda, B, ldb, beta, C, m, mode)
    torch.cuda.synchronize()
    assert st == 0
    return C


def _residual(op_a, op_b, m, n, k, A, B, C, ns=1024):
    return O.relative_residual_sampled(op_a, op_b, m, n, k, A.cpu().numpy().T, B.cpu().numpy().T,
                                       C.cpu().numpy().T, ns=ns)


def test_c2_fp64_int8_9_8192(oz):
    """BASELINE configs[1]: the headline workload"""
    import torch
    m_, h = oz
    n = 8192
    A = _dev_rand((n, n), 1)   # (k, m) row-major == column-major m x k
    B = _dev_rand((n, n), 2)
    C = torch.empty((n, n), dtype=torch.float64, device="cuda")
    _gemm(m_, h, "N", "N", n, n, n, A, B, C, "fp64_int8_9")
    r = _residual("N", "N", n, n, n, A, B, C)
    assert r < 1e-15, r                                   # the reference's gate; measured ~9e-17
    # exact scaling: row i of A times 2^(i%37 - 18), column j of B times 2^(j%29 - 14)
    # exact powers of two (torch.pow on the GPU is not exact): built with ldexp on the host
    ea = torch.from_numpy(np.ldexp(1.0, np.arange(n) % 37 - 18)).cuda()
    eb = torch.from_numpy(np.ldexp(1.0, np.arange(n) % 29 - 14)).cuda()
    A2 = A * ea[None, :]                                  # A stored (k, m): scale along m
    B2 = B * eb[:, None]                                  # B stored (n, k): scale along n
    C2 = torch.empty_like(C)
    _gemm(m_, h, "N", "N", n, n, n, A2, B2, C2, "fp64_int8_9")
    assert torch.equal(C2, C * ea[None, :] * eb[:, None])  # C stored (n, m)
    # row equivariance
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    A3 = A[:, perm].contiguous()
    C3 = torch.empty_like(C)
    _gemm(m_, h, "N", "N", n, n, n, A3, B, C3, "fp64_int8_9")
    assert torch.equal(C3, C[:, perm])
    # and bits, not only properties: two 256 x 256 blocks of the 8192^3 result (an interior one that straddles tile and XCD-patch
    # boundaries, and the last corner) against the oracle on the same 256 rows of A and 256 columns of B over the full K.
    # Rows and columns are cut independently of each other, so the block of the full product IS the product of the sub-matrices.
    for r0, c0 in ((4000, 1936), (n - 256, n - 256)):
        a_sub = A[:, r0:r0 + 256].contiguous().cpu().numpy().T        # 256 x K, column-major
        b_sub = B[c0:c0 + 256, :].contiguous().cpu().numpy().T        # K x 256, column-major
        c_ref = np.zeros((256, 256), order="F")
        assert O.gemm("N", "N", 256, 256, n, 1.0, a_sub, b_sub, 0.0, c_ref, 9, O.ORDER_DIAGONAL) == 0
        got = C[c0:c0 + 256, r0:r0 + 256].cpu().numpy().T
        np.testing.assert_array_equal(got.view(np.uint64), np.ascontiguousarray(c_ref).view(np.uint64))


def test_c3_slice_sweep_4096_wide_exponent(oz):
    """BASELINE configs[2]: fp64_int8_{3..18}, N=4096, entries u*10^(8w)"""
    import torch
    m_, h = oz
    n = 4096
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    def wide():
        u = torch.rand((n, n), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
        w = torch.rand((n, n), dtype=torch.float64, device="cuda", generator=g)
        return u * torch.pow(torch.tensor(10.0, dtype=torch.float64, device="cuda"), 8 * w)
    A, B = wide(), wide()
    C = torch.empty((n, n), dtype=torch.float64, device="cuda")
    a_h, b_h = A.cpu().numpy().T, B.cpu().numpy().T
    res = {}
    for S in range(3, 19):
        _gemm(m_, h, "N", "N", n, n, n, A, B, C, f"fp64_int8_{S}")
        res[S] = O.relative_residual_sampled("N", "N", n, n, n, a_h, b_h, C.cpu().numpy().T, ns=512)
    for S in range(3, 9):                                  # ~2^-7 per slice while truncation dominates
        assert 30 < res[S] / res[S + 1] < 500, (S, res)
    assert res[3] < 1e-3 and res[9] < 1e-15
    for S in range(10, 19):                                # saturated: identical FP64 rounding floor
        assert res[S] < 3e-16
    C_native = torch.empty_like(C)
    assert m_.native_dgemm(h, "N", "N", n, n, n, 1.0, A, n, B, n, 0.0, C_native, n) == 0
    torch.cuda.synchronize()
    r_native = O.relative_residual_sampled("N", "N", n, n, n, a_h, b_h, C_native.cpu().numpy().T, ns=512)
    assert res[10] <= r_native                             # >= 10 slices: at least as accurate as native DGEMM


def test_c4_auto_mode_16384_graded(oz):
    """BASELINE configs[3]: fp64_int8_auto, N=16384, A column-graded over 10 decades (cond ~1e10), threshold 1.5"""
    import torch
    m_, h = oz
    n = 16384
    A = _dev_rand((n, n), 7)                               # stored (k, m): grade along k
    A *= torch.pow(10.0, -10.0 * torch.arange(n, device="cuda").double() / (n - 1))[:, None]
    B = _dev_rand((n, n), 8)
    mode = m_.auto_mode_select(h, "N", "N", n, n, n, A, n, B, n, m_.real, 1.5)
    # oracle on a sub-problem with the same statistics (rows are i.i.d.): leading 2048 rows / columns
    s_ref, _ = O.auto_select("N", "N", 2048, 2048, n, np.asfortranarray(A[:, :2048].cpu().numpy().T),
                             np.asfortranarray(B[:2048, :].cpu().numpy().T), 1.5)
    assert m_.get_num_split(mode) == s_ref == 11           # graded A -> fp64_int8_11
    m_.set_auto_mantissa_loss_threashold(h, 1.5)
    C = torch.empty((n, n), dtype=torch.float64, device="cuda")
    _gemm(m_, h, "N", "N", n, n, n, A, B, C, "fp64_int8_auto")
    r = _residual("N", "N", n, n, n, A, B, C, ns=512)
    assert r < 1e-15, r
    m_.set_auto_mantissa_loss_threashold(h, 0.0)


def test_c5_hpl_panel_32768_k1024_nt(oz):
    """BASELINE configs[4]: fp64_int8_9, M=N=32768, K=1024, transA=N transB=T"""
    import torch
    m_, h = oz
    m = n = 32768
    k = 1024
    A = _dev_rand((k, m), 11)                              # A is m x k, lda = m
    B = _dev_rand((k, n), 12)                              # op T: B stored n x k, ldb = n
    C = torch.empty((n, m), dtype=torch.float64, device="cuda")
    _gemm(m_, h, "N", "T", m, n, k, A, B, C, "fp64_int8_9")
    r = O.relative_residual_sampled("N", "T", m, n, k, A.cpu().numpy().T, B.cpu().numpy().T, C.cpu().numpy().T, ns=2048)
    assert r < 1e-15, r
    # beta path at full size: C <- 0.5*A*B^T + 2*C must equal 2.5x the product up to one rounding per element
    C2 = C.clone()
    _gemm(m_, h, "N", "T", m, n, k, A, B, C2, "fp64_int8_9", alpha=0.5, beta=2.0)
    assert torch.equal(C2, 2.5 * C)                        # powers of two and 2.5x: fma(0.5, x, 2x) == 2.5x exactly


@pytest.mark.parametrize("k", [288, 320])   # 9 k-blocks: 32x32x32 tile; 10: the k64 tile function (even number of k-blocks)
@pytest.mark.parametrize("S", [3, 4, 6, 9, 12])
def test_sub_block_consistency_across_tile_shapes(oz, S, k):
    """An element of C depends only on its row of op(A) and its column of op(B): computing a column block of C in a
    separate, smaller call must give the same bits.  The full call (4096 x 8192 outputs) and the block calls
    (4096 x 448 and 200 x 8192) are scheduled differently -- other workgroup shape (128x64 vs 64x64 for S <= 6),
    other tile order, other k rotation, 32x32x32 or 16x16x64 tile function, full- or reduced-height tiles -- so this pins
    the launch policy (slice_gemm_launch.h: pick_kernel, plan_wide) to the arithmetic."""
    import torch
    m_, h = oz
    m, n = 4096, 8192
    A = _dev_rand((m, k), 11, -4.0, 4.0)            # op T: stored k-contiguous, (rows, ld=k)
    B = _dev_rand((n, k), 12)                       # op N: (cols, ld=k)
    mode = f"fp64_int8_{S}"
    C = torch.empty((n, m), dtype=torch.float64, device="cuda")
    _gemm(m_, h, "T", "N", m, n, k, A, B, C, mode)
    j0, nb = 1000, 448                               # a column block of C
    Cb = torch.empty((nb, m), dtype=torch.float64, device="cuda")
    _gemm(m_, h, "T", "N", m, nb, k, A, B[j0:j0 + nb], Cb, mode)
    assert torch.equal(Cb, C[j0:j0 + nb])
    i0, mb = 3000, 200                               # a row block of C
    Cr = torch.empty((n, mb), dtype=torch.float64, device="cuda")
    _gemm(m_, h, "T", "N", mb, n, k, A[i0:i0 + mb], B, Cr, mode)
    assert torch.equal(Cr, C[:, i0:i0 + mb])


def test_two_streams_share_one_handle_without_racing_on_the_workspace(oz):
    """The reference has one global handle and one workspace (src/cublas.cu:58, src/handle.cu:63-93): calls arriving
    on different streams would overwrite each other's slice planes.  Here a stream switch makes the new stream wait
    for the previous user of the workspace: back-to-back calls on two streams must both be bit-identical to the
    same calls issued alone."""
    import torch
    m_, h = oz
    n, mode = 2048, "fp64_int8_9"
    A1, B1 = _dev_rand((n, n), 31), _dev_rand((n, n), 32)
    A2, B2 = _dev_rand((n, n), 33, -8.0, 8.0), _dev_rand((n, n), 34)
    ref1 = torch.empty((n, n), dtype=torch.float64, device="cuda")
    ref2 = torch.empty_like(ref1)
    _gemm(m_, h, "N", "N", n, n, n, A1, B1, ref1, mode)
    _gemm(m_, h, "N", "T", n, n, n, A2, B2, ref2, mode)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    c1 = [torch.zeros_like(ref1) for _ in range(3)]
    c2 = [torch.zeros_like(ref1) for _ in range(3)]
    torch.cuda.synchronize()
    try:
        for i in range(3):
            m_.set_cuda_stream(h, s1)
            assert m_.gemm(h, "N", "N", n, n, n, 1.0, A1, n, B1, n, 0.0, c1[i], n, mode) == 0
            m_.set_cuda_stream(h, s2)
            assert m_.gemm(h, "N", "T", n, n, n, 1.0, A2, n, B2, n, 0.0, c2[i], n, mode) == 0
        torch.cuda.synchronize()
    finally:
        m_.set_cuda_stream(h, torch.cuda.current_stream())
    for i in range(3):
        assert torch.equal(c1[i], ref1) and torch.equal(c2[i], ref2)
