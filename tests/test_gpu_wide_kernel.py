"""GPU parity tests of the wide slice-GEMM kernel (slice_gemm_w_kernel.h: one 4-wave workgroup per CU, (32*WA) x 128
tiles, accumulators split over the VGPR and AGPR halves, inline-asm MFMAs and LDS-DMA copies).

The library picks this kernel only for problems that fill the chip; OZIMMU_HIP_GEMM_KERNEL=wide forces it so that the
small, ragged shapes the oracle can check in seconds run through it too: mixed tile heights (full + reduced rows of
tiles), rows beyond the padded planes (clamped row-blocks), one k-step, the tail steps of both prefetch distances, K
chunking, the two diagonal passes of S >= 13, and the complex path.  Same bar as test_gpu_parity.py: INT32 diagonal sums
and the FP64 result bit-exact vs the oracle (OZ_ORDER_DIAGONAL)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import ColMajor, exp_rand, operand, uniform_pm1, wide_exponent

pytestmark = pytest.mark.gpu


# "wide": the 32x32x32 tile function (w_tile); "x16": the paired 16x16x64 tile function (slice_gemm_x_tile.h: two slice
# products of a diagonal per MFMA).  Same persistent kernel, same staging; every test below runs through both.
# "k64": the 64-k-step 16x16x64 tile function (slice_gemm_y_tile.h: one slice product over 64 k per MFMA; single-pass modes with
# an even number of k-blocks - every other case falls back to "wide", which keeps these tests meaningful for all shapes)
@pytest.fixture(autouse=True, params=["wide", "x16", "k64"])
def force_wide(monkeypatch, request):
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", request.param)


def _sync():
    import torch
    torch.cuda.synchronize()


# every S has its own instantiation (WA = 4 for S <= 7, 3 for 8..9, 2 for 10..12; S = 13: 4 + 2; S >= 14: first pass)
@pytest.mark.parametrize("S", list(range(3, 19)))
@pytest.mark.parametrize("m,n,k", [(97, 129, 65), (300, 140, 200)])
def test_wide_diagonal_sums_bit_exact(ozh, S, m, n, k):
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
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (64, 64, 32), (96, 128, 64), (97, 129, 33), (200, 130, 96),
                                   (389, 257, 130), (1000, 200, 70)])
@pytest.mark.parametrize("S", [4, 8, 9, 11, 13, 16])
def test_wide_gemm_bit_exact_vs_oracle(oz, op_a, op_b, m, n, k, S):
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


@pytest.mark.parametrize("grid", [1, 3, 7])
@pytest.mark.parametrize("m,n,k,S", [(700, 520, 96, 9), (333, 900, 70, 6), (500, 300, 20000, 9), (260, 390, 64, 14)])
def test_wide_persistent_workgroups_claim_many_tiles(oz, monkeypatch, grid, m, n, k, S):
    """the persistent form of the kernel (per-XCD tile queues + stealing, what large problems run): forced down to a
    handful of workgroups so that each one walks many tiles of both heights, re-using its LDS and registers; K chunks
    and the two diagonal passes use separate claim counters"""
    m_, h = oz
    monkeypatch.setenv("OZIMMU_HIP_WIDE_GRID", str(grid))
    rng = np.random.default_rng(m + n + k + S + grid)
    a = operand("T", m, k, rng, pad=1)
    b = operand("N", k, n, rng)
    c = ColMajor(m, n, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n)
    c_ref.buf[...] = c.buf
    assert m_.gemm(h, "T", "N", m, n, k, 1.5, a.dev, a.ld, b.dev, b.ld, -0.5, c.dev, c.ld, f"fp64_int8_{S}") == 0
    _sync()
    kchunk = (2147483647 // (S * 127 * 127)) // 64 * 64   # the library keeps a pass at an even number of 32-k blocks
    assert O.gemm("T", "N", m, n, k, 1.5, a.view, b.view, -0.5, c_ref.view, S, O.ORDER_DIAGONAL,
                  kchunk=kchunk if k > kchunk else 0) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


@pytest.mark.parametrize("alpha,beta", [(-2.5, 0.0), (1.0, 1.0), (0.75, -1.25)])
def test_wide_gemm_alpha_beta(oz, alpha, beta):
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
def test_wide_gemm_k_chunking(oz, S):
    """K above the INT32-safe pass length: chained passes through the FP64 acc workspace (and, for S = 14, the two
    diagonal passes: the first on the wide kernel, the second on the classic one)"""
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


def test_wide_equals_classic_bitwise_on_a_chip_filling_problem(oz, monkeypatch):
    """2048 x 1536 x 1024 (fills 256 CUs with mixed-height tiles): the two kernels give the same bits"""
    import torch
    m_, h = oz
    m, n, k = 2048, 1536, 1024
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.rand(k, m, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    b = torch.rand(n, k, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    out = {}
    for which in ("wide", "classic", "x16", "k64"):
        monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", which)
        c = torch.full((n, m), float("nan"), dtype=torch.float64, device="cuda")
        assert m_.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c, m, "fp64_int8_9") == 0
        _sync()
        out[which] = c
    assert torch.equal(out["wide"].view(torch.int64), out["classic"].view(torch.int64))
    assert torch.equal(out["wide"].view(torch.int64), out["x16"].view(torch.int64))
    assert torch.equal(out["wide"].view(torch.int64), out["k64"].view(torch.int64))
    ref = (b @ a)  # row-major view of the column-major product
    assert ((out["wide"] - ref).norm() / ref.norm()).item() < 1e-14


@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "T")])
def test_wide_zgemm_bit_exact(oz, op_a, op_b):
    from tests.test_gpu_zgemm import zbits, zoperand
    m_, h = oz
    m, n, k, S = 130, 140, 96, 9
    rng = np.random.default_rng(17)
    a = zoperand(op_a, m, k, rng, "wide", pad=1)
    b = zoperand(op_b, k, n, rng, "wide")
    c = zoperand("N", m, n, rng, pad=3)
    c_ref = ColMajor(m, n, ld=m + 3, dtype=np.complex128)
    c_ref.buf[...] = c.buf
    alpha, beta = 0.5 - 1.5j, 0.25 + 0.75j
    st = m_.gemm(h, op_a, op_b, m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c.dev, c.ld, "fp64_int8_9", m_.complx)
    _sync()
    assert st == 0
    assert O.zgemm(op_a, op_b, m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(zbits(c.download()), zbits(c_ref.view))


@pytest.mark.parametrize("kernel", ["wide", "classic", "x16", "k64"])
@pytest.mark.parametrize("ld_extra,c_offset", [(0, 0), (2, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (-0.5, 2.0)])
@pytest.mark.parametrize("S", [9, 13])
def test_epilogue_store_forms_agree_with_the_oracle(oz, monkeypatch, kernel, ld_extra, c_offset, alpha, beta, S):
    """the epilogue stores interior blocks of a real GEMM as 16-byte row pairs (lane-pair exchange) when ldc is even and
    C is 16-byte aligned, and element by element otherwise or at the edges: an odd ldc, a C that starts 8 bytes off, and
    ragged edges (m, n not multiples of the tile) must all give the oracle's bits, with and without the read of C,
    in one pass (S = 9) and when a second diagonal pass continues the chain from the workspace (S = 13)"""
    import torch
    m_, h = oz
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", kernel)
    m, n, k = 262, 301, 96
    ld = m + ld_extra
    rng = np.random.default_rng(17 + ld_extra + 2 * c_offset + S)
    a = operand("N", m, k, rng)
    b = operand("N", k, n, rng)
    c_ref = ColMajor(m, n, ld=ld, fill=uniform_pm1, rng=rng)
    flat = torch.full((n * ld + c_offset + 1,), float("nan"), dtype=torch.float64, device="cuda")
    flat[c_offset:c_offset + n * ld] = torch.from_numpy(c_ref.buf.reshape(-1)).cuda()
    c_ptr = flat.data_ptr() + 8 * c_offset
    assert m_.gemm(h, "N", "N", m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c_ptr, ld, f"fp64_int8_{S}") == 0
    _sync()
    assert O.gemm("N", "N", m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    got = flat[c_offset:c_offset + n * ld].cpu().numpy().reshape(n, ld)
    np.testing.assert_array_equal(np.ascontiguousarray(got.T[:m]).view(np.uint64),
                                  np.ascontiguousarray(c_ref.view).view(np.uint64))
    if ld > m:
        assert np.isnan(got[:, m:]).all()  # ld padding untouched
    assert np.isnan(flat[:c_offset].cpu().numpy()).all() and np.isnan(flat[c_offset + n * ld:].cpu().numpy()).all()


@pytest.mark.parametrize("xcds", [1, 2, 4, 16])
@pytest.mark.parametrize("kernel,grid", [("wide", 0), ("wide", 5), ("k64", 0), ("k64", 3), ("classic", 0), ("k2", 0)])
def test_tile_partition_follows_the_xcd_count(oz, monkeypatch, xcds, kernel, grid):
    """nothing in the kernels assumes 8 XCDs.  The XCD count is a kernel argument (probed per device,
    csrc/topology.h); OZIMMU_HIP_XCDS emulates parts with 1, 2, 4 (and 16) XCDs: the runs of tiles per XCD, the per-XCD
    phase lines and the claim counters of the persistent workgroups follow it, and every tile is still computed exactly
    once: the oracle's bits."""
    m_, h = oz
    monkeypatch.setenv("OZIMMU_HIP_XCDS", str(xcds))
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", kernel)
    if grid:
        monkeypatch.setenv("OZIMMU_HIP_WIDE_GRID", str(grid))
    m, n, k, S = (700, 520, 96, 9) if kernel != "k2" else (500, 390, 160, 9)
    rng = np.random.default_rng(m + xcds + grid)
    a = operand("T", m, k, rng, pad=1)
    b = operand("N", k, n, rng)
    c = ColMajor(m, n, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n)
    c_ref.buf[...] = c.buf
    assert m_.gemm(h, "T", "N", m, n, k, 1.5, a.dev, a.ld, b.dev, b.ld, -0.5, c.dev, c.ld, f"fp64_int8_{S}") == 0
    _sync()
    assert O.gemm("T", "N", m, n, k, 1.5, a.view, b.view, -0.5, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


@pytest.mark.parametrize("xcds", [1, 2, 4])
def test_chip_filling_problem_on_emulated_xcd_counts(oz, monkeypatch, xcds):
    """3072 x 2560 x 1024 with the production policy (persistent wide kernel, per-XCD queues + stealing, phase hints):
    the emulated XCD counts give the bits of the real topology"""
    import torch
    m_, h = oz
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    m, n, k = 3072, 2560, 1024
    g = torch.Generator(device="cuda").manual_seed(2)
    a = torch.rand(k, m, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    b = torch.rand(n, k, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    out = {}
    for x in (0, xcds):
        if x:
            monkeypatch.setenv("OZIMMU_HIP_XCDS", str(x))
        c = torch.full((n, m), float("nan"), dtype=torch.float64, device="cuda")
        assert m_.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c, m, "fp64_int8_9") == 0
        _sync()
        out[x] = c
    assert torch.equal(out[0].view(torch.int64), out[xcds].view(torch.int64))


# (S >= 13: the FIRST diagonal pass runs on the k64 tile function - it stages the slices 0 .. ND-1 only -, the second on the
# paired / 32x32x32 one)
@pytest.mark.parametrize("S", [3, 4, 5, 6, 7, 8, 9, 10, 13, 14, 17, 18])
@pytest.mark.parametrize("m,n,k", [(97, 129, 64), (300, 140, 192), (64, 128, 448), (70, 129, 1120)])
def test_k64_tile_diagonal_sums_bit_exact(ozh, monkeypatch, S, m, n, k):
    """the k64 tile function with an even number of k-blocks (2, 6, 14: its own path, not the fallback): INT32 diagonal sums
    bit-exact, every single-pass S it is built for, mixed tile heights and clamped row-blocks"""
    import torch
    m_, h = ozh   # the INT32 dump is a test hook: libozimmu_hip_test.so
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", "k64")
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
    np.testing.assert_array_equal(out.cpu().numpy().transpose(0, 2, 1).astype(np.int64), d_ref)


@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "T")])
# (1056 = 33 k-blocks: beyond 32 the planes are padded to an even count - layout.h: k_blocks - so that the k64 tile applies)
@pytest.mark.parametrize("m,n,k", [(64, 64, 64), (200, 130, 128), (389, 257, 256), (1000, 200, 320), (130, 200, 1056)])
@pytest.mark.parametrize("S", [4, 8, 9, 10, 13, 16])
@pytest.mark.parametrize("grid", [0, 3])
def test_k64_tile_gemm_bit_exact_vs_oracle(oz, monkeypatch, op_a, op_b, m, n, k, S, grid):
    m_, h = oz
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", "k64")
    if grid:
        monkeypatch.setenv("OZIMMU_HIP_WIDE_GRID", str(grid))
    rng = np.random.default_rng(m + 2 * n + 3 * k + S + grid)
    a = operand(op_a, m, k, rng, pad=1)
    b = operand(op_b, k, n, rng, pad=2)
    c = ColMajor(m, n, ld=m + 3, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n, ld=m + 3)
    c_ref.buf[...] = c.buf
    st = m_.gemm(h, op_a, op_b, m, n, k, -0.75, a.dev, a.ld, b.dev, b.ld, 1.25, c.dev, c.ld, f"fp64_int8_{S}")
    _sync()
    assert st == 0
    assert O.gemm(op_a, op_b, m, n, k, -0.75, a.view, b.view, 1.25, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))
    assert np.isnan(c.buf[:, m:]).all()  # ld padding untouched
