"""GPU tests of the measured kernel choice (csrc/kernel_tuner.h): a plain real GEMM whose (mode, op_A, op_B, m, n, k, beta != 0) a handle
is called with for the 8th time (OZIMMU_HIP_AUTOTUNE_AFTER; most tests here set it to 1) runs the kernels the cost model predicts
within 25 % of its best in turn (four rounds), times each whole call with two events on the caller's stream and keeps the fastest.  The reference has no counterpart (every slice product is a cublasGemmEx that cuBLAS plans:
/root/reference/src/gemm.cu:315-329); what makes this legitimate here is that every kernel returns the same bits.

What is asserted: EVERY call of the exploration and after it is bit-exact against the oracle (OZ_ORDER_DIAGONAL), whatever kernel
ran; the tuner visits each candidate, decides on one of them, and from then on only that kernel runs; calls queued without any
synchronisation behave the same; a handle destroyed with samples in flight is harmless; forced kernels, development switches and
OZIMMU_HIP_AUTOTUNE=0 bypass it (the rest of the suite runs with it off - tests/conftest.py - because the tests that assert
WHICH kernel the model picks must not depend on what a timing decides)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import ColMajor, operand, uniform_pm1

pytestmark = pytest.mark.gpu

SLOTS = ["k2", "classic", "wide", "x16", "k64", "k64_breg"]


def _sync():
    import torch
    torch.cuda.synchronize()


def _candidates(m_, h, S, m, n, k):
    pred, pick = m_.policy_predict(h, S, m, n, k)
    base = pred[pick]
    if base < 100.0:  # TUNE_MIN_US: short calls keep the model's pick
        return [pick]
    band = 1.25 if (k + 31) // 32 <= 16 else 1.12  # csrc/kernel_tuner.cpp: TUNE_BAND_SHORT / TUNE_BAND
    others = sorted((v, nm) for nm, v in pred.items() if nm != pick and v <= band * base)
    return [pick] + [nm for _, nm in others][:3]  # TUNE_MAX_CAND = 4


def _shape_with(m_, h, S, want, shapes):
    for (m, n, k) in shapes:
        c = _candidates(m_, h, S, m, n, k)
        if (len(c) >= 2) == want:
            return (m, n, k), c
    pytest.skip("the cost model offers no such shape on this device")


def _ran(name):
    return "k2" if name == "k2_one_launch" else name


class _Case:
    def __init__(self, m, n, k, S, seed, op_a="N", op_b="N", alpha=1.25, beta=0.5):
        rng = np.random.default_rng(seed)
        self.args = (op_a, op_b, m, n, k)
        self.S, self.alpha, self.beta = S, alpha, beta
        self.a = operand(op_a, m, k, rng, pad=1)
        self.b = operand(op_b, k, n, rng, pad=2)
        self.c0 = ColMajor(m, n, ld=m + 3, fill=uniform_pm1, rng=rng)
        self.c = ColMajor(m, n, ld=m + 3)
        ref = ColMajor(m, n, ld=m + 3)
        ref.buf[...] = self.c0.buf
        assert O.gemm(op_a, op_b, m, n, k, alpha, self.a.view, self.b.view, beta, ref.view, S, O.ORDER_DIAGONAL) == 0
        self.want = ref.view.view(np.uint64).copy()
        self.reset()

    def reset(self):
        self.c.dev.copy_(self.c0.dev)

    def call(self, m_, h):
        op_a, op_b, m, n, k = self.args
        return m_.gemm(h, op_a, op_b, m, n, k, self.alpha, self.a.dev, self.a.ld, self.b.dev, self.b.ld, self.beta,
                       self.c.dev, self.c.ld, f"fp64_int8_{self.S}")

    def check(self):
        np.testing.assert_array_equal(self.c.download().view(np.uint64), self.want)


# (calls the model predicts under 100 us are not tuned: outputs of >= 1e7 elements here)
SHAPES = [(4096, 4096, 256), (3072, 4096, 384), (2048, 2048, 2048), (1536, 1536, 1536), (4096, 2048, 512), (4096, 4096, 384)]


@pytest.mark.parametrize("S", [9, 8])
def test_exploration_is_bit_exact_visits_every_candidate_and_settles(oz, monkeypatch, S):
    m_, _ = oz
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE_AFTER", "1")
    for sw in ("OZIMMU_HIP_GEMM_KERNEL", "OZIMMU_HIP_K64_BREG", "OZIMMU_HIP_K64_TILE", "OZIMMU_HIP_PAIRED_TILE", "OZIMMU_HIP_WIDE_GRID"):
        monkeypatch.delenv(sw, raising=False)
    h = m_.create()
    try:
        (m, n, k), cand = _shape_with(m_, h, S, True, SHAPES)
        case = _Case(m, n, k, S, seed=S * 1000 + m)
        assert m_.tuner_state(h, f"fp64_int8_{S}", m, n, k)[0] == -1
        seen = []
        for i in range(4 * len(cand) + 2):  # four rounds (the first one is a warm-up) + the calls that collect them
            case.reset()
            assert case.call(m_, h) == 0
            _sync()
            seen.append(_ran(m_.last_kernel(h)[0]))
            case.check()  # every call of the exploration, whatever kernel ran
        assert seen[0] == cand[0]  # the first call of a shape runs what the model picks
        assert set(cand) <= set(seen), (cand, seen)
        st, slot, nc = m_.tuner_state(h, f"fp64_int8_{S}", m, n, k)
        assert st == 1 and nc == len(cand) and SLOTS[slot] in cand, (st, slot, nc, cand, seen)
        for _ in range(4):  # decided: only the winner runs
            case.reset()
            assert case.call(m_, h) == 0
            _sync()
            assert _ran(m_.last_kernel(h)[0]) == SLOTS[slot]
            case.check()
        # the switch turns it off at once (tests: the environment is followed per call) and the model's pick runs again
        monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "0")
        case.reset()
        assert case.call(m_, h) == 0
        _sync()
        assert _ran(m_.last_kernel(h)[0]) == cand[0]
        case.check()
    finally:
        _sync()
        m_.destroy(h)


def test_queued_calls_without_synchronisation(oz, monkeypatch):
    """40 calls of one shape back to back (the samples are collected by later calls with hipEventQuery, never waited for), two
    shapes interleaved: the last result of each is bit-exact and both shapes end up decided"""
    m_, _ = oz
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE_AFTER", "1")
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    h = m_.create()
    try:
        (m, n, k), cand = _shape_with(m_, h, 9, True, SHAPES)
        (m2, n2, k2), cand2 = _shape_with(m_, h, 8, True, SHAPES[::-1])
        c1 = _Case(m, n, k, 9, seed=5, op_a="T", op_b="N", beta=0.0)
        c2 = _Case(m2, n2, k2, 8, seed=6, op_a="N", op_b="T", beta=0.0)
        for rounds in range(4):
            for _ in range(10):
                assert c1.call(m_, h) == 0
                assert c2.call(m_, h) == 0
            _sync()
            c1.check()
            c2.check()
        assert m_.tuner_state(h, "fp64_int8_9", m, n, k)[0] == 1
        assert m_.tuner_state(h, "fp64_int8_8", m2, n2, k2)[0] == 1
    finally:
        _sync()
        m_.destroy(h)


def test_one_candidate_is_decided_without_a_measurement(oz, monkeypatch):
    m_, _ = oz
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE_AFTER", "1")
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    h = m_.create()
    try:
        (m, n, k), cand = _shape_with(m_, h, 6, False, [(300, 300, 1500), (2048, 2048, 2048), (4096, 4096, 1024), (64, 64, 2048)])
        case = _Case(m, n, k, 6, seed=11, beta=0.0)
        assert case.call(m_, h) == 0
        _sync()
        case.check()
        st, slot, nc = m_.tuner_state(h, "fp64_int8_6", m, n, k)
        assert (st, nc) == (1, 1) and SLOTS[slot] == cand[0] == _ran(m_.last_kernel(h)[0])
    finally:
        _sync()
        m_.destroy(h)


def test_forced_kernels_and_switches_bypass_the_tuner(oz, monkeypatch):
    m_, _ = oz
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE_AFTER", "1")
    h = m_.create()
    try:
        (m, n, k), cand = _shape_with(m_, h, 9, True, SHAPES)
        case = _Case(m, n, k, 9, seed=12, beta=0.0)
        for env in ({"OZIMMU_HIP_GEMM_KERNEL": "classic"}, {"OZIMMU_HIP_K64_BREG": "0"}, {"OZIMMU_HIP_WIDE_GRID": "7"}):
            for kk, vv in env.items():
                monkeypatch.setenv(kk, vv)
            for _ in range(3):
                assert case.call(m_, h) == 0
            _sync()
            case.check()
            if "OZIMMU_HIP_GEMM_KERNEL" in env:
                assert m_.last_kernel(h)[0] == "classic"
            assert m_.tuner_state(h, "fp64_int8_9", m, n, k)[0] == -1  # never entered
            for kk in env:
                monkeypatch.delenv(kk)
    finally:
        _sync()
        m_.destroy(h)


def test_destroying_a_handle_with_samples_in_flight(oz, monkeypatch):
    m_, _ = oz
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE_AFTER", "1")
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    for rep in range(3):
        h = m_.create()
        (m, n, k), cand = _shape_with(m_, h, 9, True, SHAPES)
        case = _Case(m, n, k, 9, seed=13 + rep, beta=0.0)
        for _ in range(2 + rep):
            assert case.call(m_, h) == 0
        m_.destroy(h)  # (the library's destroy does not synchronise: events of unfinished calls are destroyed with it)
        _sync()
        case.check()


def test_two_threads_two_handles_tune_independently(oz, monkeypatch):
    """the override of a call is the calling THREAD's (the handle's lock is held for the enqueue): two host threads, each with
    its own handle and stream, explore different shapes at the same time"""
    import threading
    import torch
    m_, _ = oz
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE_AFTER", "1")
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    hs = [m_.create(), m_.create()]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    errors = []
    try:
        picks = [_shape_with(m_, hs[0], 9, True, SHAPES), _shape_with(m_, hs[1], 8, True, SHAPES[::-1])]
        cases = [_Case(*picks[0][0], 9, seed=21, beta=0.0), _Case(*picks[1][0], 8, seed=22, beta=0.0)]
        for h, s in zip(hs, streams):
            m_.set_cuda_stream(h, s)

        def work(i):
            try:
                for _ in range(5):
                    for _ in range(6):
                        assert cases[i].call(m_, hs[i]) == 0
                    streams[i].synchronize()
                    cases[i].check()
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
        assert m_.tuner_state(hs[0], "fp64_int8_9", *picks[0][0])[0] == 1
        assert m_.tuner_state(hs[1], "fp64_int8_8", *picks[1][0])[0] == 1
    finally:
        _sync()
        for h in hs:
            m_.destroy(h)


# ---------------------------------------------------------------- round 6: safe for real call streams

def _timed(fn, reps):
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def test_the_first_seven_calls_of_a_shape_run_the_model_pick_untimed(oz, monkeypatch):
    """production default (OZIMMU_HIP_AUTOTUNE_AFTER unset = 8): calls 1..7 of a shape run what the model picks and take no
    sample - a shape a factorisation sees a few times never pays for candidates predicted up to 25 % slower; the 8th call is
    the first sample, and 16 calls later the shape is decided"""
    m_, _ = oz
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
    monkeypatch.delenv("OZIMMU_HIP_AUTOTUNE_AFTER", raising=False)
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    h = m_.create()
    try:
        (m, n, k), cand = _shape_with(m_, h, 9, True, SHAPES)
        case = _Case(m, n, k, 9, seed=31)
        for i in range(7):
            case.reset()
            assert case.call(m_, h) == 0
            _sync()
            assert _ran(m_.last_kernel(h)[0]) == cand[0]
            case.check()
            st, slot, nc, seen, meas = m_.tuner_state(h, "fp64_int8_9", m, n, k, full=True)
            assert (st, slot, nc, seen, meas) == (0, -1, len(cand), i + 1, 0)
        seen_kernels = []
        for i in range(4 * len(cand) + 2):
            case.reset()
            assert case.call(m_, h) == 0
            _sync()
            seen_kernels.append(_ran(m_.last_kernel(h)[0]))
            case.check()
        assert set(cand) <= set(seen_kernels), (cand, seen_kernels)
        st, slot, nc, seen, meas = m_.tuner_state(h, "fp64_int8_9", m, n, k, full=True)
        assert st == 1 and meas == 1 and SLOTS[slot] in cand and seen == 7 + 4 * len(cand) + 2
    finally:
        _sync()
        m_.destroy(h)


def test_layout_and_beta_class_are_part_of_the_shape(oz, monkeypatch):
    """the split in front of the GEMM is inside every sample and costs differently per operand layout, beta != 0
    adds a read of C: (N,N) and (T,N), beta = 0 and beta != 0 of one (m, n, k) are separate entries with separate counts"""
    m_, _ = oz
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
    monkeypatch.delenv("OZIMMU_HIP_AUTOTUNE_AFTER", raising=False)
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    h = m_.create()
    try:
        (m, n, k), cand = _shape_with(m_, h, 9, True, SHAPES)
        variants = [("N", "N", 0.0), ("T", "N", 0.0), ("N", "N", 0.5)]
        cases = [_Case(m, n, k, 9, seed=40 + i, op_a=oa, op_b=ob, beta=be) for i, (oa, ob, be) in enumerate(variants)]
        for rep in range(3):
            for c in cases[:rep + 1]:      # 3, 2, 1 calls
                c.reset()
                assert c.call(m_, h) == 0
        _sync()
        for c in cases:
            c.check()
        for (oa, ob, be), want in zip(variants, (3, 2, 1)):
            st, slot, nc, seen, meas = m_.tuner_state(h, "fp64_int8_9", m, n, k, op_a=oa, op_b=ob, beta_nonzero=be != 0.0, full=True)
            assert (st, seen, meas) == (0, want, 0), (oa, ob, be, st, seen, meas)
        assert m_.tuner_state(h, "fp64_int8_9", m, n, k, op_a="T", op_b="T")[0] == -1
    finally:
        _sync()
        m_.destroy(h)


def test_a_call_stream_of_many_shapes_seen_three_times_costs_nothing(oz, monkeypatch):
    """100 distinct shapes x 3 calls each (a factorisation's shrinking trailing matrix), tuner on (production default) and off:
    every call runs the model's pick, no sample is taken, and the stream takes the same time within 1 % (best of 3 passes each,
    alternating, a fresh handle per pass: every shape is new to it)"""
    import torch
    m_, _ = oz
    monkeypatch.delenv("OZIMMU_HIP_AUTOTUNE_AFTER", raising=False)
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    nb, big = 256, 4096
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    a = torch.rand(big, nb, dtype=torch.float64, device="cuda", generator=g) - 0.5   # column-major k x m: op T
    b = torch.rand(big, nb, dtype=torch.float64, device="cuda", generator=g) - 0.5   # column-major k x n: op N
    c = torch.zeros(big, big, dtype=torch.float64, device="cuda")
    shapes = [(big - 24 * i, big - 24 * i, nb) for i in range(100)]

    def one_pass(sw):
        monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", sw)
        h = m_.create()
        try:
            m_.reallocate_working_memory(h, [("T", "N", big, big, nb, m_.real, "fp64_int8_9")])   # growth outside the clock

            def stream_of_calls():
                for (m, n, k) in shapes:
                    for _ in range(3):
                        assert m_.gemm(h, "T", "N", m, n, k, 1.0, a, nb, b, nb, 0.0, c, big, "fp64_int8_9") == 0

            t = _timed(stream_of_calls, 1)
            if sw == "1":
                for (m, n, k) in shapes[::17]:
                    st, slot, nc, seen, meas = m_.tuner_state(h, "fp64_int8_9", m, n, k, full=True)
                    assert seen == 3 and meas == 0 and (st == 0 or nc == 1), (m, st, nc, seen, meas)
            return t
        finally:
            _sync()
            m_.destroy(h)

    one_pass("0")      # (first use of the kernels, clock ramp)
    best = {"0": 1e30, "1": 1e30}
    for rep in range(3):
        for sw in ("0", "1"):
            best[sw] = min(best[sw], one_pass(sw))
    assert best["1"] <= 1.01 * best["0"], best


def test_samples_perturbed_by_another_stream_do_not_stand_for_ever(oz, monkeypatch):
    """a sample is a whole call's time on the caller's stream, and whatever another stream ran meanwhile is
    in it.  A second stream hammers the device with DGEMMs through the first measurement; every call stays bit-exact and the
    shape is decided on one of its candidates; then the neighbour stops, the shape stays in use, and the measurement is
    repeated 64 calls later on a quiet device: from then on the decided kernel is at least as fast as the model's pick."""
    import torch
    m_, _ = oz
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE_AFTER", "1")
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    h = m_.create()
    side = torch.cuda.Stream()
    x = torch.rand(3072, 3072, dtype=torch.float64, device="cuda")
    try:
        (m, n, k), cand = _shape_with(m_, h, 9, True, SHAPES)
        case = _Case(m, n, k, 9, seed=51, beta=0.0)
        for i in range(4 * len(cand) + 2):
            with torch.cuda.stream(side):
                for _ in range(1 + i % 3):      # a neighbour that comes and goes
                    torch.mm(x, x)
            assert case.call(m_, h) == 0
            _sync()
            case.check()
        st, slot, nc, seen, meas = m_.tuner_state(h, "fp64_int8_9", m, n, k, full=True)
        assert st == 1 and meas == 1 and SLOTS[slot] in cand
        # quiet from here: 64 decided calls, then the second measurement
        for i in range(64 + 4 * len(cand) + 2):
            assert case.call(m_, h) == 0
            if i % 8 == 7:
                _sync()
        _sync()
        case.check()
        st, slot2, nc, seen, meas = m_.tuner_state(h, "fp64_int8_9", m, n, k, full=True)
        assert st == 1 and meas == 2 and SLOTS[slot2] in cand
        if SLOTS[slot2] != cand[0]:     # (deciding on the model's own pick needs no timing to be right)
            t_decided, t_model = 1e30, 1e30
            for _ in range(4):          # alternating legs: the part's clock drifts by several percent over seconds
                monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
                t_decided = min(t_decided, _timed(lambda: case.call(m_, h), 20))
                monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "0")
                t_model = min(t_model, _timed(lambda: case.call(m_, h), 20))
            assert t_decided <= 1.03 * t_model, (t_decided, t_model, SLOTS[slot], SLOTS[slot2], cand)
    finally:
        _sync()
        m_.destroy(h)


def test_a_handle_that_has_been_captured_is_never_tuned(oz, monkeypatch):
    """hipEventQuery / hipEventRecord next to a capture that may be in flight on another thread are not something to
    find out in production: once a call of the handle has been captured into a graph, its later eager calls run the model's
    pick without a sample"""
    import torch
    m_, _ = oz
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE", "1")
    monkeypatch.setenv("OZIMMU_HIP_AUTOTUNE_AFTER", "1")
    monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
    h = m_.create()
    s = torch.cuda.Stream()
    try:
        (m, n, k), cand = _shape_with(m_, h, 9, True, SHAPES)
        case = _Case(m, n, k, 9, seed=61, beta=0.0)
        m_.set_cuda_stream(h, s)
        with torch.cuda.stream(s):
            assert case.call(m_, h) == 0          # eager: the workspace exists (and one sample is in flight)
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                assert case.call(m_, h) == 0
            g.replay()
            s.synchronize()
            case.check()
            for _ in range(6):
                assert case.call(m_, h) == 0
                s.synchronize()
                assert _ran(m_.last_kernel(h)[0]) == cand[0]
            case.check()
        st, slot, nc, seen, meas = m_.tuner_state(h, "fp64_int8_9", m, n, k, full=True)
        assert meas == 0 and seen == 1
    finally:
        _sync()
        m_.destroy(h)
