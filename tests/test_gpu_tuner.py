"""GPU tests of the measured kernel choice (csrc/kernel_tuner.h): a plain real GEMM whose (mode, m, n, k) a handle sees again runs
the kernels the cost model predicts within 25 % of its best in turn (four rounds), times each whole call with two events on the caller's
stream and keeps the fastest.  The reference has no counterpart (every slice product is a cublasGemmEx that cuBLAS plans:
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
