"""GPU tests of the host pipeline's robustness: strided batches through one set of
launches, host threads sharing a handle on different streams, the error paths (failure before / after C was modified),
BLAS quick returns, the stage profiler's report, and BASELINE config C1 verbatim."""
import ctypes
import os
import subprocess
import sys
import textwrap
import threading

import numpy as np
import pytest

import ozimmu_amd
from oracle import oracle as O
from tests.util import ColMajor, operand, uniform_pm1, wide_exponent

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sync():
    import torch
    torch.cuda.synchronize()


# ---------------------------------------------------------------- strided batches (src/cublas.cu:315-512)

@pytest.mark.parametrize("op_a,op_b", [("N", "N"), ("T", "N"), ("N", "T")])
@pytest.mark.parametrize("m,n,k,S,batch", [(70, 50, 90, 9, 3), (130, 200, 64, 6, 5), (64, 64, 40, 13, 2)])
def test_strided_batched_equals_per_matrix_gemm(oz, op_a, op_b, m, n, k, S, batch):
    """one set of launches for the whole batch == the reference's per-matrix loop, bit for bit, and == the oracle"""
    import torch
    m_, h = oz
    rng = np.random.default_rng(m + n + k + S)
    ar, ac = (m, k) if op_a == "N" else (k, m)
    br, bc = (k, n) if op_b == "N" else (n, k)
    lda, ldb, ldc = ar + 1, br + 2, m + 3
    sa, sb, sc = lda * ac + 5, ldb * bc + 7, ldc * n + 11          # padded strides
    A = rng.uniform(-1, 1, batch * sa)
    B = rng.uniform(-1, 1, batch * sb)
    C0 = rng.uniform(-1, 1, batch * sc)
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    dC = torch.from_numpy(C0.copy()).cuda()
    st = m_.gemm_strided_batched(h, torch.cuda.current_stream(), op_a, op_b, m, n, k, 1.25, dA, lda, sa, dB, ldb, sb, -0.5,
                                 dC, ldc, sc, batch, f"fp64_int8_{S}")
    _sync()
    assert st == 0
    got = dC.cpu().numpy()
    for i in range(batch):
        a = np.lib.stride_tricks.as_strided(A[i * sa:], shape=(ar, ac), strides=(8, 8 * lda))
        b = np.lib.stride_tricks.as_strided(B[i * sb:], shape=(br, bc), strides=(8, 8 * ldb))
        c_ref = C0[i * sc:i * sc + ldc * n].copy()
        cv = np.lib.stride_tricks.as_strided(c_ref, shape=(m, n), strides=(8, 8 * ldc))
        assert O.gemm(op_a, op_b, m, n, k, 1.25, a, b, -0.5, cv, S, O.ORDER_DIAGONAL) == 0
        np.testing.assert_array_equal(got[i * sc:i * sc + ldc * n].view(np.uint64), c_ref.view(np.uint64))
    # the gaps between the matrices of C are untouched
    for i in range(batch - 1):
        np.testing.assert_array_equal(got[i * sc + ldc * n:(i + 1) * sc], C0[i * sc + ldc * n:(i + 1) * sc])


def test_strided_batched_chunks_and_broadcast_operands(oz, monkeypatch):
    """a workspace budget of one slot forces one chunk per matrix; stride 0 broadcasts A"""
    import torch
    m_, h = oz
    m, n, k, S, batch = 96, 64, 128, 9, 4
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.rand(k, m, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    b = torch.rand(batch, n, k, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    out = {}
    for budget in ("1", str(1 << 34)):
        monkeypatch.setenv("OZIMMU_HIP_BATCH_WORKSPACE_BYTES", budget)
        c = torch.zeros(batch, n, m, dtype=torch.float64, device="cuda")
        assert m_.gemm_strided_batched(h, torch.cuda.current_stream(), "N", "N", m, n, k, 1.0, a, m, 0, b, k, n * k, 0.0, c,
                                       m, n * m, batch, f"fp64_int8_{S}") == 0
        _sync()
        out[budget] = c
    assert torch.equal(out["1"].view(torch.int64), out[str(1 << 34)].view(torch.int64))
    ref = torch.matmul(b, a)          # row-major view of the column-major products
    assert ((out["1"] - ref).norm() / ref.norm()).item() < 1e-15


def test_zgemm_strided_batched_equals_per_matrix(oz):
    import torch
    m_, h = oz
    m, n, k, batch = 70, 66, 100, 3
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.view_as_complex(torch.rand(batch, k, m, 2, dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
    b = torch.view_as_complex(torch.rand(batch, n, k, 2, dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
    c0 = torch.view_as_complex(torch.rand(batch, n, m, 2, dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
    cb = c0.clone()
    assert m_.gemm_strided_batched(h, torch.cuda.current_stream(), "N", "N", m, n, k, 0.5 - 1j, a, m, k * m, b, k, n * k,
                                   0.25 + 2j, cb, m, n * m, batch, "fp64_int8_9", m_.complx) == 0
    cl = c0.clone()
    for i in range(batch):
        assert m_.gemm(h, "N", "N", m, n, k, 0.5 - 1j, a[i], m, b[i], k, 0.25 + 2j, cl[i], m, "fp64_int8_9", m_.complx) == 0
    _sync()
    assert torch.equal(torch.view_as_real(cb).view(torch.int64), torch.view_as_real(cl).view(torch.int64))


# ---------------------------------------------------------------- host threads on one handle

def test_two_host_threads_two_streams_share_one_handle(oz):
    """Each thread enqueues GEMMs with ozimmu_hip_gemm_on_stream on its own stream, reading inputs produced on that
    stream right before the call and consuming C right after it.  Were a GEMM enqueued on the other thread's stream
    (the set_stream race of round 1) it would not be ordered with its producer / consumer.  Results must equal the
    single-threaded run bit for bit."""
    import torch
    m_, h = oz
    n, S, reps = 512, 6, 12
    base = [torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4)]
    _sync()

    def work(tid, stream, out):
        torch.cuda.set_device(0)
        with torch.cuda.stream(stream):
            for r in range(reps):
                a = base[2 * tid] * (1.0 + r)          # produced on this stream
                b = base[2 * tid + 1] + float(r)
                c = torch.empty(n, n, dtype=torch.float64, device="cuda")
                st = m_.gemm_on_stream(h, stream, "N", "T", n, n, n, 1.0, a, n, b, n, 0.0, c, n, f"fp64_int8_{S}")
                assert st == 0
                out.append(c.sum(dim=0))               # consumed on this stream
        stream.synchronize()

    solo = [[], []]
    for tid in range(2):
        work(tid, torch.cuda.Stream(), solo[tid])
    _sync()
    par = [[], []]
    threads = [threading.Thread(target=work, args=(tid, torch.cuda.Stream(), par[tid])) for tid in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    _sync()
    for tid in range(2):
        for x, y in zip(solo[tid], par[tid]):
            assert torch.equal(x.view(torch.int64), y.view(torch.int64))


# ---------------------------------------------------------------- error paths

def test_injected_launch_failure_real_leaves_c_untouched(ozh, monkeypatch):
    """status 3 = failed before C was written: the caller (the interposer) may fall back to the vendor GEMM"""
    import torch
    m_, h = ozh   # launch-failure injection is a test hook: libozimmu_hip_test.so
    n = 128
    a = torch.rand(n, n, dtype=torch.float64, device="cuda")
    c = torch.full((n, n), 7.0, dtype=torch.float64, device="cuda")
    monkeypatch.setenv("OZIMMU_HIP_TEST_FAIL_LAUNCH", "1")
    monkeypatch.setenv("OZIMMU_ERROR", "0")
    assert m_.gemm(h, "N", "N", n, n, n, 1.0, a, n, a, n, 0.5, c, n, "fp64_int8_9") == 3
    assert m_.gemm(h, "N", "N", n, n, n, 1.0, a, n, a, n, 0.5, c, n, "fp64_int8_14") == 3   # two diagonal passes
    monkeypatch.setenv("OZIMMU_HIP_TEST_FAIL_LAUNCH", "2")           # second K chunk (k > INT32-safe pass length)
    ak = torch.rand(20000, n, dtype=torch.float64, device="cuda")
    assert m_.gemm(h, "T", "N", n, n, 20000, 1.0, ak, 20000, ak, 20000, 0.5, c, n, "fp64_int8_9") == 3
    _sync()
    assert (c == 7.0).all()
    monkeypatch.delenv("OZIMMU_HIP_TEST_FAIL_LAUNCH")
    assert m_.gemm(h, "N", "N", n, n, n, 1.0, a, n, a, n, 0.5, c, n, "fp64_int8_9") == 0


def test_injected_launch_failure_complex_reports_modified_c(ozh, monkeypatch):
    """the complex path scales C by beta before its four products: a later failure is status 4, never 3"""
    import torch
    m_, h = ozh
    n = 96
    a = torch.view_as_complex(torch.rand(n, n, 2, dtype=torch.float64, device="cuda"))
    c = torch.view_as_complex(torch.rand(n, n, 2, dtype=torch.float64, device="cuda"))
    monkeypatch.setenv("OZIMMU_ERROR", "0")
    monkeypatch.setenv("OZIMMU_HIP_TEST_FAIL_LAUNCH", "3")
    assert m_.gemm(h, "N", "N", n, n, n, 1.0, a, n, a, n, 0.5 + 0.5j, c, n, "fp64_int8_9", m_.complx) == 4


PRELOAD_FAIL = r"""
#include <hip/hip_runtime_api.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
int main() {
  const int n = 256;
  std::vector<double> A(n * n, 0.5), C(n * n, 2.0), out(n * n);
  std::vector<rocblas_double_complex> ZA(n * n, rocblas_double_complex{0.5, 0.25}), ZC(n * n, rocblas_double_complex{2.0, 1.0}), zout(n * n);
  double *dA, *dC; rocblas_double_complex *zA, *zC;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dC, n * n * 8); hipMalloc(&zA, n * n * 16); hipMalloc(&zC, n * n * 16);
  hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), n * n * 8, hipMemcpyHostToDevice);
  hipMemcpy(zA, ZA.data(), n * n * 16, hipMemcpyHostToDevice); hipMemcpy(zC, ZC.data(), n * n * 16, hipMemcpyHostToDevice);
  rocblas_handle h; rocblas_create_handle(&h);
  const double alpha = 1.0, beta = 0.5;
  rocblas_status st = rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, n, n, n, &alpha, dA, n, dA, n, &beta, dC, n);
  hipDeviceSynchronize();
  hipMemcpy(out.data(), dC, n * n * 8, hipMemcpyDeviceToHost);
  printf("DGEMM status=%d c00=%.6f\n", (int)st, out[0]);          // 0.25*256 + 0.5*2 = 65
  const rocblas_double_complex za{1.0, 0.0}, zb{0.5, 0.0};
  st = rocblas_zgemm(h, rocblas_operation_none, rocblas_operation_none, n, n, n, &za, zA, n, zA, n, &zb, zC, n);
  hipDeviceSynchronize();
  printf("ZGEMM status=%d\n", (int)st);
  rocblas_destroy_handle(h);
  return 0;
}
"""


def test_preload_fallback_only_when_c_is_untouched(tmp_path):
    """through LD_PRELOAD: a real-path failure (C untouched) falls back to the vendor DGEMM with the right result;
    a complex-path failure after the beta scaling returns an error status instead of a double-scaled C"""
    src = tmp_path / "d.cpp"
    src.write_text(PRELOAD_FAIL)
    exe = tmp_path / "d"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src), "-o",
                           str(exe), "-L/opt/rocm/lib", "-lrocblas", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    e = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
    e.update(LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_9", OZIMMU_INTERCEPT_THRESHOLD_M="64",
             OZIMMU_INTERCEPT_THRESHOLD_N="64", OZIMMU_INTERCEPT_THRESHOLD_K="64")
    e1 = dict(e, OZIMMU_HIP_TEST_FAIL_LAUNCH="1", LD_PRELOAD=ozimmu_amd.TEST_LIB_PATH)  # the injection is a test hook
    p = subprocess.run([str(exe)], env=e1, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "DGEMM status=0 c00=65.000000" in p.stdout, p.stdout           # vendor fallback, beta applied once
    assert "falling back to the vendor GEMM" in p.stdout
    assert "ZGEMM status=0" not in p.stdout and "ZGEMM status=" in p.stdout  # launch 1 of 4 fails after the scaling
    assert "after C had been modified" in p.stdout
    p = subprocess.run([str(exe)], env=e, capture_output=True, text=True, timeout=300)
    assert "DGEMM status=0 c00=65.000000" in p.stdout and "ZGEMM status=0" in p.stdout, p.stdout


def test_workspace_allocation_failure_is_status_3_with_c_untouched(oz, monkeypatch):
    """VERDICT r5 weak 8 (/root/reference/src/handle.cu:63-93 throws through the C ABI here): a workspace the device cannot
    hold - 9 slices x (m + n) x K = 1.2 TB for K = 2^24 under a 4096 x 4096 output - fails in hipMalloc BEFORE anything is
    launched: status 3, C untouched, and the handle is left usable (the old block was released: the next call allocates
    again).  A and B are never dereferenced on this path (nothing is enqueued), so small tensors stand in for them."""
    import torch
    m_, _ = oz
    h = m_.create()
    try:
        monkeypatch.setenv("OZIMMU_ERROR", "0")
        n, k = 4096, 1 << 24
        stub = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
        c = torch.full((n, n), 7.0, dtype=torch.float64, device="cuda")
        need = ozimmu_amd.lib().ozimmu_hip_working_memory_size(0, 0, n, n, k, 0, m_._mode("fp64_int8_9"))
        assert need > torch.cuda.get_device_properties(0).total_memory
        # a small call first: there IS a block to lose
        a = torch.rand(256, 256, dtype=torch.float64, device="cuda")
        c_small = torch.zeros(256, 256, dtype=torch.float64, device="cuda")
        assert m_.gemm(h, "N", "N", 256, 256, 256, 1.0, a, 256, a, 256, 0.0, c_small, 256, "fp64_int8_9") == 0
        _sync()
        want = c_small.clone()
        assert m_.gemm(h, "N", "N", n, n, k, 1.0, stub, n, stub, k, 0.5, c, n, "fp64_int8_9") == 3
        _sync()
        assert (c == 7.0).all()
        c_small.zero_()
        assert m_.gemm(h, "N", "N", 256, 256, 256, 1.0, a, 256, a, 256, 0.0, c_small, 256, "fp64_int8_9") == 0
        _sync()
        assert torch.equal(c_small.view(torch.int64), want.view(torch.int64))
    finally:
        _sync()
        m_.destroy(h)


def test_preload_falls_back_to_the_vendor_gemm_when_the_workspace_does_not_fit():
    """the same through LD_PRELOAD with real operands: all but 3 GB of the free HBM is held by the application when a
    16384^3 float64 matmul arrives (workspace 4.8 GB): the shim reports the failed allocation, the vendor DGEMM runs, the
    result is the vendor's bit for bit; once the application releases memory the next call runs on the Ozaki path"""
    code = textwrap.dedent("""
        import ctypes, os, torch
        n = 16384
        torch.manual_seed(0)
        a = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
        b = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
        out = torch.empty(n, n, dtype=torch.float64, device="cuda")
        ref = torch.empty(n, n, dtype=torch.float64, device="cuda")
        os.environ["OZIMMU_COMPUTE_MODE"] = "dgemm"
        torch.mm(a, b, out=ref)                       # the vendor's own result (and its first-use allocations)
        torch.cuda.synchronize()
        os.environ["OZIMMU_COMPUTE_MODE"] = "fp64_int8_9"
        free, total = torch.cuda.mem_get_info()
        hog = torch.empty(free - (3 << 30), dtype=torch.uint8, device="cuda")
        torch.mm(a, b, out=out)
        torch.cuda.synchronize()
        print("SAME_AS_VENDOR", torch.equal(out.view(torch.int64), ref.view(torch.int64)))
        lib = ctypes.CDLL(os.environ["LD_PRELOAD"])
        st = (ctypes.c_ulonglong * 4)()
        lib.ozimmu_hip_intercept_stats(st, 4)
        print("STATS1", list(st))
        del hog
        torch.cuda.empty_cache()
        torch.mm(a, b, out=out)
        torch.cuda.synchronize()
        lib.ozimmu_hip_intercept_stats(st, 4)
        print("STATS2", list(st))
        print("MAXDIFF %.3e" % (out - ref).abs().max().item())
    """)
    e = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
    e.update(LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_9")
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "SAME_AS_VENDOR True" in p.stdout, p.stdout
    assert "falling back to the vendor GEMM" in p.stdout
    # (a hipBLAS application reaches the shim twice when the Ozaki path declines: at the hipBLAS entry point and, through the
    # vendor hipBLAS routine it is forwarded to, at the rocBLAS one - two attempts seen and declined for the one torch.mm)
    import json
    st1 = json.loads([l for l in p.stdout.splitlines() if l.startswith("STATS1")][0][len("STATS1 "):])
    st2 = json.loads([l for l in p.stdout.splitlines() if l.startswith("STATS2")][0][len("STATS2 "):])
    assert st1[0] in (1, 2) and st1 == [st1[0], 0, st1[0], 0], st1
    assert st2 == [st1[0] + 1, 1, st1[0], 0], (st1, st2)
    diff = float([l for l in p.stdout.splitlines() if l.startswith("MAXDIFF")][0].split()[1])
    assert 0 < diff < 1e-9, diff       # the Ozaki result: not the vendor's bits, FP64-accurate (|C| ~ 1e2)


# ---------------------------------------------------------------- BLAS quick returns / out-of-range sizes

def test_quick_returns_do_not_split_the_operands(oz):
    """alpha == 0 or k == 0: C = beta*C on the vendor path; A and B are not read (a NaN in A must not reach C)"""
    import torch
    m_, h = oz
    n = 64
    a = torch.full((n, n), float("nan"), dtype=torch.float64, device="cuda")
    c = torch.full((n, n), 3.0, dtype=torch.float64, device="cuda")
    assert m_.gemm(h, "N", "N", n, n, n, 0.0, a, n, a, n, 2.0, c, n, "fp64_int8_9") == 0
    _sync()
    assert (c == 6.0).all()
    assert m_.gemm(h, "N", "N", n, n, 0, 1.0, a, n, a, n, 0.5, c, n, "fp64_int8_9") == 0
    _sync()
    assert (c == 3.0).all()
    z = torch.view_as_complex(torch.full((n, n, 2), 1.0, dtype=torch.float64, device="cuda"))
    za = torch.view_as_complex(torch.full((n, n, 2), float("nan"), dtype=torch.float64, device="cuda"))
    assert m_.gemm(h, "N", "N", n, n, 0, 1.0, za, n, za, n, 2.0, z, n, "fp64_int8_9", m_.complx) == 0   # was SIGFPE
    _sync()
    assert (torch.view_as_real(z) == 2.0).all()


def test_working_memory_size_of_degenerate_k():
    lib = ozimmu_amd.lib()
    f = lib.ozimmu_hip_working_memory_size
    assert f(0, 0, 1024, 1024, 0, ozimmu_amd.real, ozimmu_amd.fp64_int8_9) == 0          # was SIGFPE (division by q*q = 0)
    assert f(0, 0, 1024, 1024, 0, ozimmu_amd.complx, ozimmu_amd.fp64_int8_9) == 0
    assert f(0, 0, 8, 8, (1 << 30) + 1, ozimmu_amd.real, ozimmu_amd.fp64_int8_9) == 0


# ---------------------------------------------------------------- stage profiler (F3: src/handle.cu:246-265)

def test_profiler_report_labels_calls_and_shares(oz, capfd):
    import torch
    m_, h = oz
    n = 512
    a = torch.rand(n, n, dtype=torch.float64, device="cuda")
    c = torch.empty(n, n, dtype=torch.float64, device="cuda")
    m_.clear_profiler_result(h)
    m_.enable_profiling(h)
    for _ in range(3):
        assert m_.gemm(h, "N", "N", n, n, n, 1.0, a, n, a, n, 0.0, c, n, "fp64_int8_9") == 0
    m_.disable_profiling(h)
    assert m_.gemm(h, "N", "N", n, n, n, 1.0, a, n, a, n, 0.0, c, n, "fp64_int8_9") == 0     # not counted
    _sync()
    capfd.readouterr()
    m_.print_profiler_result(h, "unit", csv=True)
    out = capfd.readouterr().out
    rows = [l.split(",") for l in out.strip().splitlines()]
    assert rows[0] == ["tag", "label", "calls", "total_ms", "share"]
    labels = [r[1] for r in rows[1:]]
    assert labels == ["split_A", "split_B", "int8tc", "accumulate_in_f64", "copy_result"]   # src/gemm.cu:38-48, :393-407
    assert all(r[0] == "unit" and r[2] == "3" for r in rows[1:])
    t = {r[1]: float(r[3]) for r in rows[1:]}
    sh = {r[1]: float(r[4]) for r in rows[1:]}
    assert t["split_A"] > 0 and t["split_B"] > 0 and t["int8tc"] > 0
    assert t["accumulate_in_f64"] == 0 and t["copy_result"] == 0      # fused into the int8tc kernel
    assert abs(sum(sh.values()) - 1.0) < 1e-3
    m_.print_profiler_result(h, "unit")
    txt = capfd.readouterr().out
    assert "[unit] (3 calls)" in txt and all(l in txt for l in labels)
    m_.clear_profiler_result(h)
    m_.print_profiler_result(h, "cleared", csv=True)
    out = capfd.readouterr().out
    assert all(l.split(",")[2] == "0" and float(l.split(",")[3]) == 0 for l in out.strip().splitlines()[1:])


# ---------------------------------------------------------------- BASELINE config C1 verbatim

def test_baseline_config_c1_fp64_int8_6_512_cubed(oz):
    """C1: fp64_int8_6, M=N=K=512, U[-1,1): HIP result bit-exact vs the oracle (the CPU leg is tests/test_oracle.py)"""
    m_, h = oz
    m = n = k = 512
    rng = np.random.default_rng(0)
    a = operand("N", m, k, rng, fill=uniform_pm1)
    b = operand("N", k, n, rng, fill=uniform_pm1)
    c, c_ref = ColMajor(m, n), ColMajor(m, n)
    assert m_.gemm(h, "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld, "fp64_int8_6") == 0
    _sync()
    assert O.gemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, 6, O.ORDER_DIAGONAL) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))
    r = O.relative_residual_sampled("N", "N", m, n, k, a.view, b.view, c.view, ns=2048)
    assert 1e-12 < r < 1e-10         # six 7-bit slices: the accuracy level of BASELINE.md's curve at S=6


def test_gemm_is_capturable_into_a_graph_once_the_workspace_exists(oz):
    """PyTorch-style graph capture of the caller's stream: with a workspace that is already large enough every launch of
    a call is an ordinary kernel, so the captured graph replays the same bits; a call that would have to GROW the
    workspace during capture (hipMalloc / hipFree are illegal there) reports "failed, C untouched" (status 3: the
    interposer would capture the vendor GEMM instead) and leaves the capture valid; fp64_int8_auto reads a statistic
    back on the host and is refused the same way"""
    import torch
    import ozimmu_amd as m_
    m, n, k, S = 300, 260, 200, 9
    rng = np.random.default_rng(77)
    a = operand("N", m, k, rng)
    b = operand("N", k, n, rng)
    c_ref = ColMajor(m, n)
    assert O.gemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    h = m_.create()
    try:
        s = torch.cuda.Stream()
        c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
        a.dev, b.dev
        torch.cuda.synchronize()
        # 1) fresh handle, no workspace yet: the captured call is refused, the capture survives
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0, stream=s):
            st_grow = m_.gemm_on_stream(h, torch.cuda.current_stream(), "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld,
                                        0.0, c, m, f"fp64_int8_{S}")
            st_auto = m_.gemm_on_stream(h, torch.cuda.current_stream(), "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld,
                                        0.0, c, m, "fp64_int8_auto")
            c.add_(1.0)  # something capturable so that the graph is not empty
        assert st_grow == 3 and st_auto == 3
        g0.replay()
        torch.cuda.synchronize()
        assert float(c.min()) == 1.0 and float(c.max()) == 1.0  # C untouched by the refused calls
        # 2) eager call grows the workspace; the same call is then captured and replayed
        with torch.cuda.stream(s):
            assert m_.gemm_on_stream(h, s, "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c, m, f"fp64_int8_{S}") == 0
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=s):
            st = m_.gemm_on_stream(h, torch.cuda.current_stream(), "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0,
                                   c, m, f"fp64_int8_{S}")
        assert st == 0
        for _ in range(2):
            c.fill_(float("nan"))
            g1.replay()
            torch.cuda.synchronize()
            np.testing.assert_array_equal(c.cpu().numpy().view(np.uint64), c_ref.buf.view(np.uint64))
        # 3) the graph is replayed on NEW data after other calls have used the handle: the split's row-exponent words
        #    carry a per-call epoch, and the replay runs with the epoch of capture time -- it must neither see the later
        #    calls' words nor its own earlier maxima (the new A is 2^-20 times smaller: a stale maximum would shift every
        #    slice)
        other = operand("T", 500, 300, rng)
        other_b = operand("N", 300, 410, rng)
        c_other = torch.zeros(410, 500, dtype=torch.float64, device="cuda")
        with torch.cuda.stream(s):
            assert m_.gemm_on_stream(h, s, "T", "N", 500, 410, 300, 1.0, other.dev, other.ld, other_b.dev, other_b.ld, 0.0,
                                     c_other, 500, "fp64_int8_11") == 0
        torch.cuda.synchronize()
        a.buf[...] *= 2.0 ** -20
        a.buf[:, 5] = 0.0
        a.dev.copy_(torch.from_numpy(a.buf))
        c_ref2 = ColMajor(m, n)
        assert O.gemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref2.view, S, O.ORDER_DIAGONAL) == 0
        c.fill_(float("nan"))
        g1.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(c.cpu().numpy().view(np.uint64), c_ref2.buf.view(np.uint64))
        # 4) a much larger eager call outgrows the workspace the graph points into: the old block must stay alive
        big_a = torch.rand(1500, 1400, dtype=torch.float64, device="cuda")
        big_b = torch.rand(1300, 1500, dtype=torch.float64, device="cuda")
        big_c = torch.zeros(1300, 1400, dtype=torch.float64, device="cuda")
        with torch.cuda.stream(s):
            assert m_.gemm_on_stream(h, s, "N", "N", 1400, 1300, 1500, 1.0, big_a, 1400, big_b, 1500, 0.0, big_c, 1400,
                                     "fp64_int8_12") == 0
        torch.cuda.synchronize()
        junk = [torch.full((1 << 22,), 3.0, dtype=torch.float64, device="cuda") for _ in range(4)]  # reuse freed memory, if any
        c.fill_(float("nan"))
        g1.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(c.cpu().numpy().view(np.uint64), c_ref2.buf.view(np.uint64))
        assert all(float(j.min()) == 3.0 and float(j.max()) == 3.0 for j in junk)  # the replay wrote into its own memory
    finally:
        torch.cuda.synchronize()
        m_.destroy(h)


def test_eager_calls_after_a_capture_on_another_stream_do_not_synchronise_the_device(oz):
    """ADVICE r4 / VERDICT r5 weak 8: a handle that had been captured AND had seen several streams synchronised the DEVICE in front
    of every later eager call, for ever.  Now the first eager call after a capture leaves the workspace and the exponent words
    the capture used to the graph and continues on fresh blocks (api.cpp: leave_blocks_to_graphs).  Checked: (1) eager calls
    on stream A while a long replay of the graph is in flight on stream B are bit-exact AND the replay is bit-exact (they share no
    memory); (2) a long kernel queued on a THIRD stream does not delay the eager calls' enqueue (a device synchronisation per
    call would wait for it every time)."""
    import time
    import torch
    import ozimmu_amd as m_
    m, n, k, S = 384, 320, 256, 9
    rng = np.random.default_rng(123)
    a = operand("N", m, k, rng)
    b = operand("N", k, n, rng)
    a2 = operand("T", 200, 300, rng)
    b2 = operand("N", 300, 260, rng)
    ref = ColMajor(m, n)
    ref2 = ColMajor(200, 260)
    assert O.gemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, ref.view, S, O.ORDER_DIAGONAL) == 0
    assert O.gemm("T", "N", 200, 260, 300, 1.0, a2.view, b2.view, 0.0, ref2.view, 8, O.ORDER_DIAGONAL) == 0
    h = m_.create()
    try:
        sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
        c2 = torch.zeros(260, 200, dtype=torch.float64, device="cuda")
        a.dev, b.dev, a2.dev, b2.dev
        torch.cuda.synchronize()
        # eager warm-up on the capture stream (workspace), capture there, eager calls on ANOTHER stream afterwards
        assert m_.gemm_on_stream(h, sb, "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c, m, f"fp64_int8_{S}") == 0
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=sb):
            for _ in range(20):      # a replay that lasts: twenty calls
                assert m_.gemm_on_stream(h, torch.cuda.current_stream(), "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c, m,
                                         f"fp64_int8_{S}") == 0
        c.fill_(float("nan"))
        with torch.cuda.stream(sb):
            g.replay()
        for _ in range(10):          # ... while eager calls of another shape run on stream A
            assert m_.gemm_on_stream(h, sa, "T", "N", 200, 260, 300, 1.0, a2.dev, a2.ld, b2.dev, b2.ld, 0.0, c2, 200, "fp64_int8_8") == 0
        torch.cuda.synchronize()
        np.testing.assert_array_equal(c.cpu().numpy().view(np.uint64), ref.buf.view(np.uint64))
        np.testing.assert_array_equal(c2.cpu().numpy().view(np.uint64), ref2.buf.view(np.uint64))
        # (2) a long-running neighbour on a third stream: enqueueing eager calls must not wait for it
        big = torch.rand(8192, 8192, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(sc):
            for _ in range(6):
                torch.mm(big, big)       # ~6 x 15 ms of device time
        t0 = time.perf_counter()
        for _ in range(10):
            assert m_.gemm_on_stream(h, sa, "T", "N", 200, 260, 300, 1.0, a2.dev, a2.ld, b2.dev, b2.ld, 0.0, c2, 200, "fp64_int8_8") == 0
        host_ms = (time.perf_counter() - t0) * 1e3
        busy = not sc.query()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(c2.cpu().numpy().view(np.uint64), ref2.buf.view(np.uint64))
        assert busy, "the neighbour finished before the eager calls were enqueued: the check says nothing"
        assert host_ms < 30.0, f"10 eager calls took {host_ms:.1f} ms of host time next to a 90 ms neighbour: a device sync per call?"
        # the graph still replays its own bits afterwards
        c.fill_(float("nan"))
        with torch.cuda.stream(sb):
            g.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(c.cpu().numpy().view(np.uint64), ref.buf.view(np.uint64))
    finally:
        torch.cuda.synchronize()
        m_.destroy(h)


@pytest.mark.parametrize("m,n,k,op_a,op_b", [
    (60000, 40, 50, "N", "N"),    # tall and skinny: ~1900 row tiles, one column tile
    (33, 50000, 70, "T", "N"),    # short and wide
    (70, 90, 40000, "N", "T"),    # deep: three chained K chunks at S = 9
    (5000, 3, 1000, "T", "T"),    # n below every tile width
    (1, 4097, 129, "N", "N"),     # a single row
])
def test_extreme_aspect_ratios_bit_exact(oz, m, n, k, op_a, op_b):
    """shapes far from square: tile grids with one row or one column of tiles, row counts that leave most of a tile
    empty, K deep enough for chunked passes; whatever kernel the library picks, the bits are the oracle's"""
    import torch
    _, h = oz
    S = 9
    rng = np.random.default_rng(m + 3 * n + 7 * k)
    a = operand(op_a, m, k, rng)
    b = operand(op_b, k, n, rng)
    c = ColMajor(m, n, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n)
    c_ref.buf[...] = c.buf
    assert ozimmu_amd.gemm(h, op_a, op_b, m, n, k, -1.25, a.dev, a.ld, b.dev, b.ld, 0.5, c.dev, c.ld, f"fp64_int8_{S}") == 0
    torch.cuda.synchronize()
    L = O.bits_per_int8(k)
    kchunk = (2147483647 // (S * ((1 << L) - 1) ** 2)) // 64 * 64   # an even number of 32-k blocks per pass
    assert O.gemm(op_a, op_b, m, n, k, -1.25, a.view, b.view, 0.5, c_ref.view, S, O.ORDER_DIAGONAL,
                  kchunk=kchunk if k > kchunk else 0) == 0
    np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64))


def test_call_after_the_previous_calls_stream_was_destroyed(oz):
    """cross-stream ordering records its event lazily, on the PREVIOUS call's stream at the moment a call arrives on a
    different one; if the application has destroyed that stream in the meantime, the record is rejected and the library
    synchronises the device instead of touching a dead handle"""
    import torch
    m_, h = oz
    hip = ctypes.CDLL("libamdhip64.so.7")  # the runtime torch and the library already share (same soname)
    m, n, k, S = 200, 150, 96, 9
    rng = np.random.default_rng(123)
    a = operand("N", m, k, rng)
    b = operand("N", k, n, rng)
    c_ref = ColMajor(m, n)
    assert O.gemm("N", "N", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    a.dev, b.dev
    torch.cuda.synchronize()
    raw = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(raw)) == 0
    c1 = torch.zeros(n, m, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert m_.gemm_on_stream(h, raw.value, "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c1, m, f"fp64_int8_{S}") == 0
    assert hip.hipStreamSynchronize(raw) == 0
    assert hip.hipStreamDestroy(raw) == 0
    c2 = torch.zeros(n, m, dtype=torch.float64, device="cuda")
    st = m_.gemm_on_stream(h, torch.cuda.current_stream(), "N", "N", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c2, m,
                           f"fp64_int8_{S}")
    torch.cuda.synchronize()
    assert st == 0
    for c in (c1, c2):
        np.testing.assert_array_equal(c.cpu().numpy().view(np.uint64), c_ref.buf.view(np.uint64))


def test_exponent_word_epoch_wraps_around(monkeypatch):
    """the split's row-exponent words are never zeroed per call: they carry the epoch of the call that wrote them and a
    reader ignores words of other epochs.  The epoch has 21 bits; OZIMMU_HIP_TEST_EXP_EPOCH jumps a fresh handle to just
    below the wrap-around, and calls of alternating shapes (stale words of the larger problem under the smaller one, zero
    rows that write nothing) must stay bit-exact across it"""
    import torch
    monkeypatch.setenv("OZIMMU_HIP_TEST_EXP_EPOCH", str((1 << 21) - 4))
    oz_t = ozimmu_amd.test_flavour()   # the epoch jump is a test hook: libozimmu_hip_test.so
    h = oz_t.create()
    try:
        rng = np.random.default_rng(99)
        for it, (m, n, k) in enumerate([(300, 200, 64), (90, 70, 33), (300, 200, 64), (64, 260, 100), (90, 70, 33),
                                        (300, 200, 64), (33, 35, 37), (300, 200, 64)]):
            if it == 2:
                monkeypatch.delenv("OZIMMU_HIP_TEST_EXP_EPOCH")  # from here on the epoch runs freely through the wrap
            a = operand("N", m, k, rng)
            b = operand("T", k, n, rng)
            a.buf[:, 3] = 0.0  # row 3 of op(A) is zero: nothing is written to its exponent word
            a._dev = None
            c = ColMajor(m, n)
            c_ref = ColMajor(m, n)
            assert oz_t.gemm(h, "N", "T", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld, 0.0, c.dev, c.ld, "fp64_int8_8") == 0
            torch.cuda.synchronize()
            assert O.gemm("N", "T", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, 8, O.ORDER_DIAGONAL) == 0
            np.testing.assert_array_equal(c.download().view(np.uint64), c_ref.view.view(np.uint64), err_msg=f"call {it}")
    finally:
        torch.cuda.synchronize()
        oz_t.destroy(h)


def test_epoch_wrap_around_with_a_captured_graph_in_flight(monkeypatch):
    """a graph captured shortly before the 21-bit epoch of the exponent words wraps replays with its old, LARGE
    tag; an eager call of the new, small epochs would lose its atomicMax against the words such a replay leaves behind and
    cut its rows as all-zero.  On the wrap a handle that has seen a capture retires the buffer (the graphs keep it) and
    continues on a fresh zeroed one: replays and eager calls stay bit-exact on both sides of the wrap, interleaved.
    Capturing a call that would itself cross the wrap is refused (allocation is illegal inside a capture)."""
    import torch
    m, n, k, S = 200, 136, 96, 8
    rng = np.random.default_rng(123)
    a = operand("N", m, k, rng)
    b = operand("T", k, n, rng)
    a2 = operand("N", m, k, rng, fill=wide_exponent(3))   # a second problem for the eager calls: other row maxima
    c_ref, c2_ref = ColMajor(m, n), ColMajor(m, n)
    assert O.gemm("N", "T", m, n, k, 1.0, a.view, b.view, 0.0, c_ref.view, S, O.ORDER_DIAGONAL) == 0
    assert O.gemm("N", "T", m, n, k, 1.0, a2.view, b.view, 0.0, c2_ref.view, S, O.ORDER_DIAGONAL) == 0
    monkeypatch.setenv("OZIMMU_HIP_TEST_EXP_EPOCH", str((1 << 21) - 6))
    oz_t = ozimmu_amd.test_flavour()   # the epoch jump is a test hook: libozimmu_hip_test.so
    h = oz_t.create()
    try:
        s = torch.cuda.Stream()
        c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
        c2 = torch.zeros(n, m, dtype=torch.float64, device="cuda")
        a.dev, a2.dev, b.dev
        torch.cuda.synchronize()

        def eager(src, dst, ref, tag):
            with torch.cuda.stream(s):
                assert oz_t.gemm_on_stream(h, s, "N", "T", m, n, k, 1.0, src.dev, src.ld, b.dev, b.ld, 0.0, dst, m,
                                                 f"fp64_int8_{S}") == 0
            torch.cuda.synchronize()
            np.testing.assert_array_equal(dst.cpu().numpy().view(np.uint64), ref.buf.view(np.uint64), err_msg=tag)

        eager(a, c, c_ref, "warm-up")                      # epoch 2^21 - 5: workspace and exponent words exist
        monkeypatch.delenv("OZIMMU_HIP_TEST_EXP_EPOCH")     # from here on the epoch runs freely
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):                 # captured with tag 2^21 - 4
            st = oz_t.gemm_on_stream(h, torch.cuda.current_stream(), "N", "T", m, n, k, 1.0, a.dev, a.ld, b.dev, b.ld,
                                           0.0, c, m, f"fp64_int8_{S}")
        assert st == 0
        for it in range(8):                                 # the eager epochs run through the wrap at it = 3
            c.fill_(float("nan"))
            g.replay()
            torch.cuda.synchronize()
            np.testing.assert_array_equal(c.cpu().numpy().view(np.uint64), c_ref.buf.view(np.uint64), err_msg=f"replay {it}")
            eager(a2, c2, c2_ref, f"eager {it}")
    finally:
        torch.cuda.synchronize()
        oz_t.destroy(h)


def test_the_policy_runs_what_it_predicts_and_reports_it(oz, monkeypatch):
    """ozimmu_hip_last_kernel / ozimmu_hip_policy_predict: the kernel a call ran is the one the cost model picked for its
    shape on this device; a forced kernel is reported as such; every choice gives the oracle's bits (the forced-kernel fuzz
    of test_gpu_parity.py), here: the report."""
    import torch
    m_, h = oz
    info = m_.device_info(h)
    assert info["cus"] >= 1 and info["xcds"] >= 1 and 0.75 * 0.0194 <= info["mfma32_us"] <= 1.25 * 0.0194
    for (m, n, k, S) in [(1024, 1024, 1024, 9), (512, 384, 2048, 6), (2048, 1536, 256, 4), (700, 900, 1024, 13)]:
        a = torch.rand(k, m, dtype=torch.float64, device="cuda")
        b = torch.rand(n, k, dtype=torch.float64, device="cuda")
        c = torch.zeros(n, m, dtype=torch.float64, device="cuda")
        monkeypatch.delenv("OZIMMU_HIP_GEMM_KERNEL", raising=False)
        assert m_.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c, m, f"fp64_int8_{S}") == 0
        torch.cuda.synchronize()
        ran = m_.last_kernel(h)
        pred, pick = m_.policy_predict(h, S, m, n, k)
        # (a small real GEMM on the K-split tile cuts its strips in the same launch: "k2_one_launch", the cost model's "k2")
        assert ran[0].replace("_one_launch", "") == pick and pick == min(pred, key=pred.get), (ran, pick, pred)
        assert (ran[1] is None) == (S <= 12)
        monkeypatch.setenv("OZIMMU_HIP_GEMM_KERNEL", "classic")
        assert m_.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c, m, f"fp64_int8_{S}") == 0
        torch.cuda.synchronize()
        assert m_.last_kernel(h)[0] == "classic"


@pytest.mark.parametrize("m,n,k,S", [(2048, 2048, 1024, 9), (1024, 4096, 2048, 9), (3072, 2048, 1024, 7), (2048, 4096, 1024, 6)])
def test_static_grid_for_exact_rounds_equals_the_queue_run(oz, monkeypatch, m, n, k, S):
    """slice_gemm_launch.h: wide_grid - a few EXACT rounds of tiles run as a static grid (no claims), everything else as
    persistent workgroups with claim counters.  Same tiles, same arithmetic: the bits of the static run equal those of the
    queue run (forced by OZIMMU_HIP_WIDE_GRID = the CU count), and the residual is the mode's."""
    import torch
    m_, h = oz
    torch.manual_seed(m + n + k + S)
    a = torch.rand(k, m, dtype=torch.float64, device="cuda") * 2 - 1
    b = torch.rand(n, k, dtype=torch.float64, device="cuda") * 2 - 1
    out = []
    for grid in (None, str(m_.device_info(h)["cus"])):
        if grid is None:
            monkeypatch.delenv("OZIMMU_HIP_WIDE_GRID", raising=False)
        else:
            monkeypatch.setenv("OZIMMU_HIP_WIDE_GRID", grid)
        c = torch.full((n, m), float("nan"), dtype=torch.float64, device="cuda")
        for _ in range(2):
            assert m_.gemm(h, "N", "N", m, n, k, 1.0, a, m, b, k, 0.0, c, m, f"fp64_int8_{S}") == 0
        torch.cuda.synchronize()
        out.append(c)
    assert torch.equal(out[0].view(torch.int64), out[1].view(torch.int64))
    ref = b @ a
    assert ((out[0] - ref).norm() / ref.norm()).item() < (1e-14 if S >= 9 else 1e-8)
