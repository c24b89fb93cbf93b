"""GPU parity of the one-launch form of small real GEMMs (csrc/slice_gemm_one_launch.hip): the 8-row strips of op(A) and op(B)
are cut by the slice GEMM's own workgroups, which publish one epoch-tagged READY word per strip and wait for the 8 + 8 strips
their 64 x 64 tile reads.  Replaces split_int8 x 2 + the pair loop of /root/reference/src/gemm.cu:344-410 for K <= 2048, S <= 9,
at most one tile per CU.

What can go wrong here that the two-launch form cannot: a consumer that reads planes before their producer's stores are
visible (or stale L2 lines of the previous call: the planes live at the same addresses in every call), a ready word of an
EARLIER call taken for this one's, the self-service path (a workgroup that cuts a strip it did not see in time), several strips
per workgroup, and the words' coexistence with the two-pass split's row-exponent words in the same buffer.  Every case is bit-exact
against the oracle (OZ_ORDER_DIAGONAL) and the identity of the kernel is asserted."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import ColMajor, exp_rand, operand, uniform_pm1

pytestmark = pytest.mark.gpu


def _sync():
    import torch
    torch.cuda.synchronize()


def _run(m_, h, op_a, op_b, m, n, k, S, rng, alpha=1.5, beta=-0.5, fill=None):
    kw = {} if fill is None else {"fill": fill}
    a = operand(op_a, m, k, rng, pad=1, **kw)
    b = operand(op_b, k, n, rng, pad=2, **kw)
    c = ColMajor(m, n, ld=m + 1, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n, ld=m + 1)
    c_ref.buf[...] = c.buf
    assert m_.gemm(h, op_a, op_b, m, n, k, alpha, a.dev, a.ld, b.dev, b.ld, beta, c.dev, c.ld, f"fp64_int8_{S}") == 0
    _sync()
    kernel = m_.last_kernel(h)[0]
    assert O.gemm(op_a, op_b, m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))
    assert np.isnan(c.buf[:, m:]).all()  # ld padding untouched
    return kernel


# one tile .. 256 tiles (1024 x 1024: every CU one workgroup, one strip each); 2 .. 32 strips per workgroup (64 x 64 x K:
# one workgroup cuts all 32 strips of the padded planes); K = 128 .. 2048 (one or two unit blocks per wave, idle waves at K < 1024)
@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
@pytest.mark.parametrize("m,n,k,S", [(64, 64, 128, 9), (1, 1, 129, 9), (200, 130, 300, 9), (1024, 1024, 128, 9), (1000, 1001, 1024, 8),
                                     (130, 1900, 2048, 9), (960, 1024, 1120, 3), (512, 448, 2000, 6)])
def test_one_launch_gemm_bit_exact_vs_oracle(oz, monkeypatch, op_a, op_b, m, n, k, S):
    m_, h = oz
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "1")
    rng = np.random.default_rng(m + 3 * n + 7 * k + S)
    kernel = _run(m_, h, op_a, op_b, m, n, k, S, rng)
    # (the cost model may prefer the classic kernel on a few of these shapes: then the two-launch form ran, also checked above)
    pred, pick = m_.policy_predict(h, S, m, n, k)
    assert kernel == ("k2_one_launch" if pick == "k2" else pick)


@pytest.mark.parametrize("m,n,k", [(512, 512, 512), (1024, 1024, 512), (640, 640, 640)])
def test_one_launch_is_the_default_where_the_policy_picks_the_k_split_tile(oz, monkeypatch, m, n, k):
    """no switch set: problems of at most one 64 x 64 tile per CU whose K loop the cost model gives to the K-split tile run
    split + GEMM as one kernel (1024^3 itself goes to the k64 register kernel, two launches: profiles/r5_ablate/r5h_*)"""
    m_, h = oz
    for kname in ("OZIMMU_HIP_GEMM_KERNEL", "OZIMMU_HIP_ONE_LAUNCH", "OZIMMU_HIP_SPLIT_RESIDENT"):
        monkeypatch.delenv(kname, raising=False)
    pick = m_.policy_predict(h, 9, m, n, k)[1]   # (follows the device's measured MFMA time: 768^3 goes either way from box to box)
    assert _run(m_, h, "N", "N", m, n, k, 9, np.random.default_rng(5)) == ("k2_one_launch" if pick == "k2" else pick)
    assert pick == "k2"


@pytest.mark.parametrize("spin", [1, 3])
@pytest.mark.parametrize("m,n,k,S", [(1024, 1024, 256, 9), (200, 130, 300, 7), (960, 1024, 2048, 4)])
def test_self_service_when_a_strip_is_not_seen_in_time(oz, monkeypatch, m, n, k, S, spin):
    """OZIMMU_HIP_ONE_LAUNCH_SPIN = 1: every workgroup gives up after one poll and cuts the strips it misses itself, racing
    their owners with identical bytes - the path that keeps the launch from hanging when fewer CUs than workgroups are free"""
    m_, h = oz
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", "k2")
    monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "1")
    monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH_SPIN", str(spin))
    rng = np.random.default_rng(m + n + k + S + spin)
    assert _run(m_, h, "T", "N", m, n, k, S, rng, fill=exp_rand(3.0)) == "k2_one_launch"


def test_ready_words_of_earlier_calls_and_of_the_two_pass_split(oz, monkeypatch):
    """One handle, alternating: one-launch calls on the same workspace addresses with different operands (a stale READY word or
    a stale L2 line of the previous call would show as the previous call's slices), and two-pass splits (K > 2048) whose
    row-exponent words share the epoch-tagged buffer with the READY words"""
    m_, h = oz
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", "k2")
    monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "1")
    rng = np.random.default_rng(11)
    for it in range(6):
        assert _run(m_, h, "N", "T", 512, 512, 512, 9, rng, fill=exp_rand(4.0)) == "k2_one_launch"
        if it % 2:
            # K = 4096: two streaming passes, atomicMax on the words the one-launch calls used as READY words
            assert _run(m_, h, "N", "N", 300, 200, 4096, 9, rng, fill=exp_rand(4.0)) == "k2"
    monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "0")
    assert _run(m_, h, "N", "T", 512, 512, 512, 9, rng) == "k2"


def test_back_to_back_calls_without_host_synchronisation(oz, monkeypatch):
    """20 calls enqueued without a host sync in between, every one on fresh operands and its own C: launch n + 1 starts while
    launch n drains; each result is checked against the oracle afterwards"""
    import torch
    m_, h = oz
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", "k2")
    monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "1")
    m, n, k, S = 320, 256, 384, 9
    rng = np.random.default_rng(3)
    cases = []
    for _ in range(20):
        a = operand("N", m, k, rng)
        b = operand("N", k, n, rng)
        c = ColMajor(m, n)
        a.dev, b.dev, c.dev  # upload before the burst
        cases.append((a, b, c))
    torch.cuda.synchronize()
    for a, b, c in cases:
        assert m_.gemm(h, "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld, f"fp64_int8_{S}") == 0
    _sync()
    assert m_.last_kernel(h)[0] == "k2_one_launch"
    for a, b, c in cases:
        c_ref = ColMajor(m, n)
        assert O.gemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL) == 0
        np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


@pytest.mark.parametrize("S", [3, 5, 9])
def test_one_launch_diagonal_sums_bit_exact(ozh, monkeypatch, S):
    """the INT32 diagonal sums (test hook) of the one-launch kernel = the oracle's"""
    import torch
    m_, h = ozh
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", "k2")
    monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "1")
    m, n, k = 389, 257, 640
    rng = np.random.default_rng(S)
    a = operand("N", m, k, rng, fill=exp_rand(2.0))
    b = operand("T", k, n, rng, fill=exp_rand(2.0))
    L = O.bits_per_int8(k)
    pa, _ = O.split("A", "N", a.view, S, L)
    pb, _ = O.split("B", "T", b.view, S, L)
    out = torch.full((S, n, m), 12345, dtype=torch.int32, device="cuda")
    assert m_.diagonal_sums(h, "N", "T", m, n, k, a.dev, a.ld, b.dev, b.ld, S, out) == 0
    _sync()
    assert m_.last_kernel(h)[0] == "k2_one_launch"
    np.testing.assert_array_equal(out.cpu().numpy().transpose(0, 2, 1).astype(np.int64), O.diagonal_sums(pa, pb))


def test_one_launch_equals_two_launches_under_churn(oz, monkeypatch):
    """Stress of the visibility protocol (slice_gemm_one_launch.hip: write-through producers, no cache maintenance): many
    calls in a row on ONE handle - same workspace addresses, fresh random operands of changing shape and layout every time, no
    host synchronisation between the one-launch call and its two-launch twin - every result compared bit for bit with the
    two-launch form of the same product.  A stale cache line of an earlier call, a READY word seen too early or a torn strip
    shows up as a mismatch.  OZIMMU_ONE_LAUNCH_STRESS=n runs n rounds (default 300; profiles/r5_ablate/r5k_*: 20 000)."""
    import os
    import torch
    m_, h = oz
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", "k2")
    rounds = int(os.environ.get("OZIMMU_ONE_LAUNCH_STRESS", "300"))
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    rng = np.random.default_rng(77)
    bad = torch.zeros((), dtype=torch.int64, device="cuda")   # counted on the device: the calls queue back to back
    for it in range(rounds):
        m, n = (int(rng.integers(1, 1025)) for _ in range(2))
        k = int(rng.choice([32, 96, 128, 200, 256, 512, 777, 1024, 1536, 2048]))
        S = int(rng.choice([3, 6, 8, 9]))
        op_a, op_b = rng.choice(["N", "T"]), rng.choice(["N", "T"])
        scale = float(rng.choice([1.0, 1e-3, 1e5]))
        a = (torch.rand((k, m) if op_a == "N" else (m, k), dtype=torch.float64, device="cuda", generator=g) - 0.5) * scale
        b = (torch.rand((n, k) if op_b == "N" else (k, n), dtype=torch.float64, device="cuda", generator=g) - 0.5)
        lda, ldb = a.shape[1], b.shape[1]
        c0 = torch.rand((n, m), dtype=torch.float64, device="cuda", generator=g)
        c1, c2 = c0.clone(), c0.clone()
        monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "1")
        assert m_.gemm(h, op_a, op_b, m, n, k, 1.25, a, lda, b, ldb, -0.5, c1, m, f"fp64_int8_{S}") == 0
        ran = m_.last_kernel(h)[0]
        monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "0")
        assert m_.gemm(h, op_a, op_b, m, n, k, 1.25, a, lda, b, ldb, -0.5, c2, m, f"fp64_int8_{S}") == 0
        assert ran == "k2_one_launch" and m_.last_kernel(h)[0] == "k2"
        bad += (c1.view(torch.int64) != c2.view(torch.int64)).any().to(torch.int64)
        if it % 50 == 49:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    assert int(bad) == 0, f"{int(bad)} of {rounds} one-launch results differ from the two-launch form"


def test_fewer_resident_workgroups_than_the_grid_degrades_within_a_time_bound(oz, monkeypatch):
    """the launch assumes every workgroup resident.  On a CU-masked stream (hipExtStreamCreateWithCUMask: 32
    of the device's CUs) 256 workgroups run 32 at a time: the resident ones wait for strips whose owners cannot start before
    they finish.  With the DEFAULT spin count the launch must degrade to self-service within a few of its own durations - the
    old default (20 000 polls per missing strip, the counter restarting after every self-cut) took hundreds of milliseconds -
    and the result stays bit-exact."""
    import ctypes
    import torch
    m_, h = oz
    monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", "k2")
    monkeypatch.setenv("OZIMMU_HIP_ONE_LAUNCH", "1")
    monkeypatch.delenv("OZIMMU_HIP_ONE_LAUNCH_SPIN", raising=False)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    words = (cus + 31) // 32
    mask = (ctypes.c_uint32 * words)(*([0] * words))
    mask[0] = 0xFFFFFFFF  # the first 32 CUs
    stream = ctypes.c_void_p()
    if hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), words, mask) != 0 or not stream.value:
        pytest.skip("hipExtStreamCreateWithCUMask is not available on this box")
    try:
        m, n, k, S = 1024, 1024, 512, 9
        rng = np.random.default_rng(99)
        a = operand("N", m, k, rng)
        b = operand("T", k, n, rng)
        c = ColMajor(m, n)
        c_ref = ColMajor(m, n)
        assert O.gemm("N", "T", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL) == 0
        a.dev, b.dev, c.dev
        torch.cuda.synchronize()
        times = []
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ext = torch.cuda.ExternalStream(stream.value)
            e0.record(ext)
            assert m_.gemm_on_stream(h, stream.value, "N", "T", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld,
                                     f"fp64_int8_{S}") == 0
            e1.record(ext)
            assert hip.hipStreamSynchronize(stream) == 0
            assert m_.last_kernel(h)[0] == "k2_one_launch"
            times.append(e0.elapsed_time(e1))
            np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))
        # 256 tiles on 32 CUs are eight rounds of the tile function (~40 us each at full occupancy) + each round's bounded wait
        # and self-service cuts: a few milliseconds at most.  The unbounded form: 8 rounds x up to 16 strips x 20-40 ms.
        assert min(times) < 20.0, times
    finally:
        torch.cuda.synchronize()
        m_.set_cuda_stream(h, None)
        hip.hipStreamDestroy(stream)
