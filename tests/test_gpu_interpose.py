"""GPU tests of the LD_PRELOAD boundary with the REAL rocBLAS / hipBLAS underneath (reference:
src/cublas.cu:103-513; the reference's own tests never exercise its interposer)."""
import os
import subprocess
import sys
import textwrap

import pytest

import ozimmu_amd

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
#include <hip/hip_runtime_api.h>
#include <rocblas/rocblas.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
// C = alpha*op(A)*op(B) + beta*C through rocblas_dgemm / rocblas_gemm_ex / strided batched, checked against a
// long double host reference on sampled entries; prints "RESIDUAL <which> <value>".
static double residual(const std::vector<double>& A, const std::vector<double>& B, const std::vector<double>& C0,
                       const std::vector<double>& C, int m, int n, int k, double alpha, double beta, bool ta, bool tb) {
  long double num = 0, den = 0;
  unsigned s = 12345;
  for (int t = 0; t < 600; t++) {
    s = s * 1664525u + 1013904223u; int i = (s >> 8) % m;
    s = s * 1664525u + 1013904223u; int j = (s >> 8) % n;
    long double acc = 0;
    for (int kk = 0; kk < k; kk++) {
      const double a = ta ? A[(size_t)i * k + kk] : A[(size_t)kk * m + i];
      const double b = tb ? B[(size_t)kk * n + j] : B[(size_t)j * k + kk];
      acc += (long double)a * b;
    }
    const long double truth = alpha * acc + beta * C0[(size_t)j * m + i];
    const long double d = C[(size_t)j * m + i] - truth;
    num += d * d; den += truth * truth;
  }
  return (double)sqrtl(num / den);
}
int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 512, n = m + 64, k = m + 32;
  const bool ta = argc > 2 && atoi(argv[2]), tb = argc > 3 && atoi(argv[3]);
  const double alpha = 1.25, beta = -0.5;
  std::vector<double> A((size_t)m * k), B((size_t)k * n), C0((size_t)m * n), C((size_t)m * n);
  unsigned s = 1;
  auto rnd = [&]() { s = s * 1103515245u + 12345u; return ((s >> 8) & 0xffffff) / 8388608.0 - 1.0; };
  for (auto& x : A) x = rnd();
  for (auto& x : B) x = rnd();
  for (auto& x : C0) x = rnd();
  double *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, 3 * C.size() * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  rocblas_handle h;
  if (rocblas_create_handle(&h) != rocblas_status_success) return 2;
  hipStream_t st; hipStreamCreate(&st); rocblas_set_stream(h, st);
  const rocblas_operation oa = ta ? rocblas_operation_transpose : rocblas_operation_none;
  const rocblas_operation ob = tb ? rocblas_operation_transpose : rocblas_operation_none;
  const int lda = ta ? k : m, ldb = tb ? n : k;
  // 1. rocblas_dgemm
  hipMemcpy(dC, C0.data(), C.size() * 8, hipMemcpyHostToDevice);
  if (rocblas_dgemm(h, oa, ob, m, n, k, &alpha, dA, lda, dB, ldb, &beta, dC, m) != rocblas_status_success) return 3;
  hipStreamSynchronize(st);
  hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
  printf("RESIDUAL dgemm %.3e\n", residual(A, B, C0, C, m, n, k, alpha, beta, ta, tb));
  // 2. rocblas_gemm_ex, all f64, in place
  hipMemcpy(dC, C0.data(), C.size() * 8, hipMemcpyHostToDevice);
  if (rocblas_gemm_ex(h, oa, ob, m, n, k, &alpha, dA, rocblas_datatype_f64_r, lda, dB, rocblas_datatype_f64_r, ldb, &beta,
                      dC, rocblas_datatype_f64_r, m, dC, rocblas_datatype_f64_r, m, rocblas_datatype_f64_r,
                      rocblas_gemm_algo_standard, 0, 0) != rocblas_status_success) return 4;
  hipStreamSynchronize(st);
  hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
  printf("RESIDUAL gemm_ex %.3e\n", residual(A, B, C0, C, m, n, k, alpha, beta, ta, tb));
  // 3. strided batched: 3 x the same product into 3 C's
  for (int b = 0; b < 3; b++) hipMemcpy(dC + (size_t)b * m * n, C0.data(), C.size() * 8, hipMemcpyHostToDevice);
  if (rocblas_dgemm_strided_batched(h, oa, ob, m, n, k, &alpha, dA, lda, 0, dB, ldb, 0, &beta, dC, m, (rocblas_stride)m * n, 3)
      != rocblas_status_success) return 5;
  hipStreamSynchronize(st);
  hipMemcpy(C.data(), dC + (size_t)2 * m * n, C.size() * 8, hipMemcpyDeviceToHost);
  printf("RESIDUAL strided_batched %.3e\n", residual(A, B, C0, C, m, n, k, alpha, beta, ta, tb));
  // 4. rocblas_gemm_strided_batched_ex, all f64, in place, 2 matrices
  for (int b = 0; b < 2; b++) hipMemcpy(dC + (size_t)b * m * n, C0.data(), C.size() * 8, hipMemcpyHostToDevice);
  if (rocblas_gemm_strided_batched_ex(h, oa, ob, m, n, k, &alpha, dA, rocblas_datatype_f64_r, lda, 0, dB,
                                      rocblas_datatype_f64_r, ldb, 0, &beta, dC, rocblas_datatype_f64_r, m,
                                      (rocblas_stride)m * n, dC, rocblas_datatype_f64_r, m, (rocblas_stride)m * n, 2,
                                      rocblas_datatype_f64_r, rocblas_gemm_algo_standard, 0, 0) != rocblas_status_success)
    return 6;
  hipStreamSynchronize(st);
  hipMemcpy(C.data(), dC + (size_t)1 * m * n, C.size() * 8, hipMemcpyDeviceToHost);
  printf("RESIDUAL strided_batched_ex %.3e\n", residual(A, B, C0, C, m, n, k, alpha, beta, ta, tb));
  // 5. the ILP64 twins: rocblas_dgemm_64, and rocblas_zgemm_64 on (A + 0i)(B + 0i)
  hipMemcpy(dC, C0.data(), C.size() * 8, hipMemcpyHostToDevice);
  if (rocblas_dgemm_64(h, oa, ob, m, n, k, &alpha, dA, lda, dB, ldb, &beta, dC, m) != rocblas_status_success) return 7;
  hipStreamSynchronize(st);
  hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
  printf("RESIDUAL dgemm_64 %.3e\n", residual(A, B, C0, C, m, n, k, alpha, beta, ta, tb));
  {
    std::vector<rocblas_double_complex> zA(A.size()), zB(B.size()), zC(C0.size());
    for (size_t i = 0; i < A.size(); i++) zA[i] = rocblas_double_complex{A[i], 0.0};
    for (size_t i = 0; i < B.size(); i++) zB[i] = rocblas_double_complex{B[i], 0.0};
    for (size_t i = 0; i < C0.size(); i++) zC[i] = rocblas_double_complex{C0[i], 0.0};
    rocblas_double_complex *dzA, *dzB, *dzC;
    hipMalloc(&dzA, zA.size() * 16); hipMalloc(&dzB, zB.size() * 16); hipMalloc(&dzC, zC.size() * 16);
    hipMemcpy(dzA, zA.data(), zA.size() * 16, hipMemcpyHostToDevice);
    hipMemcpy(dzB, zB.data(), zB.size() * 16, hipMemcpyHostToDevice);
    hipMemcpy(dzC, zC.data(), zC.size() * 16, hipMemcpyHostToDevice);
    const rocblas_double_complex za{alpha, 0.0}, zb{beta, 0.0};
    if (rocblas_zgemm_64(h, oa, ob, m, n, k, &za, dzA, lda, dzB, ldb, &zb, dzC, m) != rocblas_status_success) return 8;
    hipStreamSynchronize(st);
    hipMemcpy(zC.data(), dzC, zC.size() * 16, hipMemcpyDeviceToHost);
    double imag = 0;
    const double* zd = reinterpret_cast<const double*>(zC.data());   // interleaved re / im
    for (size_t i = 0; i < zC.size(); i++) { C[i] = zd[2 * i]; imag = fmax(imag, fabs(zd[2 * i + 1])); }
    printf("RESIDUAL zgemm_64 %.3e\n", residual(A, B, C0, C, m, n, k, alpha, beta, ta, tb) + imag);
  }
  // 6. the application switches to a new stream and destroys the old one (the shim must not touch the dead stream),
  //    then destroys its BLAS handle (the shim's per-device state goes with the last one) and starts over
  hipStream_t st2; hipStreamCreate(&st2); rocblas_set_stream(h, st2);
  hipStreamSynchronize(st); hipStreamDestroy(st);
  hipMemcpy(dC, C0.data(), C.size() * 8, hipMemcpyHostToDevice);
  if (rocblas_dgemm(h, oa, ob, m, n, k, &alpha, dA, lda, dB, ldb, &beta, dC, m) != rocblas_status_success) return 9;
  hipStreamSynchronize(st2);
  hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
  printf("RESIDUAL dgemm_after_stream_destroy %.3e\n", residual(A, B, C0, C, m, n, k, alpha, beta, ta, tb));
  rocblas_destroy_handle(h);
  hipStreamDestroy(st2);
  rocblas_handle h2;
  if (rocblas_create_handle(&h2) != rocblas_status_success) return 10;
  hipStream_t st3; hipStreamCreate(&st3); rocblas_set_stream(h2, st3);
  hipMemcpy(dC, C0.data(), C.size() * 8, hipMemcpyHostToDevice);
  if (rocblas_dgemm(h2, oa, ob, m, n, k, &alpha, dA, lda, dB, ldb, &beta, dC, m) != rocblas_status_success) return 11;
  hipStreamSynchronize(st3);
  hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
  printf("RESIDUAL dgemm_after_handle_recreate %.3e\n", residual(A, B, C0, C, m, n, k, alpha, beta, ta, tb));
  // 7. two host threads, each with its own BLAS handle and stream, hammer the shim's shared per-device state
  {
    double *dC2[2];
    std::vector<double> Cs[2];
    int bad = 0;
    auto worker = [&](int t) {
      rocblas_handle ht; hipStream_t stt;
      if (rocblas_create_handle(&ht) != rocblas_status_success) { bad = 1; return; }
      hipStreamCreate(&stt); rocblas_set_stream(ht, stt);
      for (int it = 0; it < 6; it++) {
        hipMemcpyAsync(dC2[t], C0.data(), C0.size() * 8, hipMemcpyHostToDevice, stt);
        if (rocblas_dgemm(ht, oa, ob, m, n, k, &alpha, dA, lda, dB, ldb, &beta, dC2[t], m) != rocblas_status_success) bad = 1;
      }
      hipStreamSynchronize(stt);
      Cs[t].resize(C0.size());
      hipMemcpy(Cs[t].data(), dC2[t], C0.size() * 8, hipMemcpyDeviceToHost);
      rocblas_destroy_handle(ht); hipStreamDestroy(stt);
    };
    for (int t = 0; t < 2; t++) hipMalloc(&dC2[t], C0.size() * 8);
    rocblas_handle keep; rocblas_create_handle(&keep);   // keeps the shim's state alive while the workers come and go
    std::thread t0(worker, 0), t1(worker, 1);
    t0.join(); t1.join();
    rocblas_destroy_handle(keep);
    if (bad) return 12;
    printf("RESIDUAL dgemm_two_threads %.3e\n",
           fmax(residual(A, B, C0, Cs[0], m, n, k, alpha, beta, ta, tb), residual(A, B, C0, Cs[1], m, n, k, alpha, beta, ta, tb)));
  }
  rocblas_destroy_handle(h2);
  return 0;
}
"""


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    d = tmp_path_factory.mktemp("gpu_interpose")
    src = d / "driver.cpp"
    src.write_text(DRIVER)
    exe = d / "driver"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src), "-o",
                           str(exe), "-L/opt/rocm/lib", "-lrocblas", "-lamdhip64", "-pthread", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def run(exe, args, **env):
    e = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([str(exe)] + [str(a) for a in args], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    res = {l.split()[1]: float(l.split()[2]) for l in p.stdout.splitlines() if l.startswith("RESIDUAL")}
    return res, p.stdout


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_preloaded_rocblas_dgemm_runs_the_ozaki_path(driver, ta, tb):
    thr = dict(OZIMMU_INTERCEPT_THRESHOLD_M=256, OZIMMU_INTERCEPT_THRESHOLD_N=256, OZIMMU_INTERCEPT_THRESHOLD_K=256)
    native, _ = run(driver, [512, ta, tb])
    assert all(v < 1e-14 for v in native.values())
    oz, out = run(driver, [512, ta, tb], LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_9",
                  OZIMMU_INFO=1, OZIMMU_ENABLE_CULIP_PROFILING=1, **thr)
    assert set(oz) == {"dgemm", "gemm_ex", "strided_batched", "strided_batched_ex", "dgemm_64", "zgemm_64",
                       "dgemm_after_stream_destroy", "dgemm_after_handle_recreate", "dgemm_two_threads"}
    assert all(v < 1e-15 for v in oz.values()), oz           # the reference's gate, through the preload
    assert "[ozIMMU LOG] Reallocated memory" in out           # src/handle.cu:69
    # CULiP line format of src/cublas.cu:157-162 / src/culip.cu:19-39
    ops = ("T" if ta else "N") + ("T" if tb else "N")
    assert f"[CULiP Result][Dfp64_int8_9-{ops}-m512-n576-k544]" in out
    assert f"[CULiP Result][Zfp64_int8_9-{ops}-m512-n576-k544]" in out          # rocblas_zgemm_64
    assert f"[CULiP Result][Dfp64_int8_9_stridedBatched-{ops}-m512-n576-k544-batch_count3]" in out
    # fewer slices -> visibly larger error: proves the environment knob reaches the kernel
    coarse, _ = run(driver, [512, ta, tb], LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_4", **thr)
    assert all(1e-10 < v < 1e-6 for v in coarse.values()), coarse


def test_thresholds_and_mode_gate_the_intercept(driver):
    # default thresholds (1024) -> 512^3 passes through even with a mode set: no ozIMMU workspace traffic in the log
    _, out = run(driver, [512, 0, 0], LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_4", OZIMMU_INFO=1)
    res, _ = run(driver, [512, 0, 0], LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_4")
    assert all(v < 1e-14 for v in res.values())               # native accuracy, not the 4-slice error
    res, _ = run(driver, [512, 0, 0], LD_PRELOAD=ozimmu_amd.LIB_PATH)
    assert all(v < 1e-14 for v in res.values())


def test_sgemm_mode_through_the_preload(driver):
    """OZIMMU_COMPUTE_MODE=sgemm (README.md:34; src/cublas.cu:169-186): FP32 accuracy, CULiP tag `Dsgemm`"""
    thr = dict(OZIMMU_INTERCEPT_THRESHOLD_M=256, OZIMMU_INTERCEPT_THRESHOLD_N=256, OZIMMU_INTERCEPT_THRESHOLD_K=256)
    res, out = run(driver, [512, 1, 0], LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="sgemm",
                   OZIMMU_ENABLE_CULIP_PROFILING=1, **thr)
    assert all(1e-9 < v < 1e-5 for v in res.values()), res
    assert "[CULiP Result][Dsgemm-TN-m512-n576-k544]" in out


def test_auto_mode_through_the_preload(driver):
    thr = dict(OZIMMU_INTERCEPT_THRESHOLD_M=256, OZIMMU_INTERCEPT_THRESHOLD_N=256, OZIMMU_INTERCEPT_THRESHOLD_K=256)
    res, out = run(driver, [512, 0, 0], LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_auto",
                   OZIMMU_AUTO_AVG_MANTISSA_LOSS_THRESHOLD=1.5, OZIMMU_INFO=1, **thr)
    assert "AUTO selected mode = fp64_int8_8, threshold average mantissa loss = 1.5" in out   # src/gemm.cu:632-635
    assert all(v < 1e-15 for v in res.values())


def test_pytorch_float64_matmul_is_intercepted():
    """torch.mm(float64) -> hipblasDgemm -> rocblas_dgemm (undefined dynamic symbol in libhipblas) -> the shim"""
    code = textwrap.dedent("""
        import torch
        torch.manual_seed(0)
        a = torch.rand(1536, 1024, dtype=torch.float64, device="cuda") * 2 - 1
        b = torch.rand(1024, 1280, dtype=torch.float64, device="cuda") * 2 - 1
        c = a @ b
        torch.cuda.synchronize()
        ref = (a.cpu().to(torch.float64) @ b.cpu().to(torch.float64))
        print("MAXDIFF %.3e" % (c.cpu() - ref).abs().max().item())
    """)
    e = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
    e.update(LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_3", OZIMMU_INFO="1")
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr
    diff3 = float([l for l in p.stdout.splitlines() if l.startswith("MAXDIFF")][0].split()[1])
    e["OZIMMU_COMPUTE_MODE"] = "fp64_int8_10"
    p10 = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p10.returncode == 0, p10.stdout + p10.stderr
    diff10 = float([l for l in p10.stdout.splitlines() if l.startswith("MAXDIFF")][0].split()[1])
    assert "[ozIMMU LOG]" in p.stdout
    assert diff3 > 1e-6 and diff10 < 1e-11, (diff3, diff10)    # 3 slices are visibly coarse, 10 are FP64-accurate


def test_pytorch_batched_matmul_is_intercepted():
    """torch.bmm(float64 / complex128) -> hipblas{D,Z}gemmStridedBatched -> rocblas_{d,z}gemm_strided_batched -> shim
    (cublasDgemmStridedBatched / cublasZgemmStridedBatched, src/cublas.cu:474-512)"""
    code = textwrap.dedent("""
        import torch
        torch.manual_seed(0)
        a = torch.rand(3, 1100, 1024, dtype=torch.float64, device="cuda") * 2 - 1
        b = torch.rand(3, 1024, 1050, dtype=torch.float64, device="cuda") * 2 - 1
        c = torch.bmm(a, b)
        z = torch.bmm(torch.complex(a, a.flip(0)), torch.complex(b, -b.flip(0)))
        torch.cuda.synchronize()
        ref = torch.bmm(a.cpu(), b.cpu())
        zref = torch.bmm(torch.complex(a, a.flip(0)).cpu(), torch.complex(b, -b.flip(0)).cpu())
        print("MAXDIFF %.3e %.3e" % ((c.cpu() - ref).abs().max().item(), (z.cpu() - zref).abs().max().item()))
    """)
    diffs = {}
    for mode in ("fp64_int8_3", "fp64_int8_10"):
        e = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
        e.update(LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE=mode, OZIMMU_ENABLE_CULIP_PROFILING="1")
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stdout + p.stderr
        line = [l for l in p.stdout.splitlines() if l.startswith("MAXDIFF")][0].split()
        diffs[mode] = (float(line[1]), float(line[2]), p.stdout)
    assert diffs["fp64_int8_3"][0] > 1e-6 and diffs["fp64_int8_3"][1] > 1e-6, diffs["fp64_int8_3"][:2]
    assert diffs["fp64_int8_10"][0] < 1e-11 and diffs["fp64_int8_10"][1] < 1e-11, diffs["fp64_int8_10"][:2]
    out = diffs["fp64_int8_10"][2]
    # one line per strided-batched call, in the reference's format (src/cublas.cu:342-350)
    assert out.count("[CULiP Result][Dfp64_int8_10_stridedBatched-") == 1, out
    assert out.count("[CULiP Result][Zfp64_int8_10_stridedBatched-") == 1, out
    assert "-m1100-n1050-k1024-batch_count3]" in out or "-m1050-n1100-k1024-batch_count3]" in out, out


def test_pytorch_conjugate_transposed_complex_matmul_is_intercepted():
    """torch: A.mH @ B (complex128) -> hipblasZgemm with HIPBLAS_OP_C -> rocblas_zgemm(conjugate_transpose) -> the shim runs it
    on the Ozaki path as OZIMMU_OP_C.  The reference's hook maps CUBLAS_OP_C to op_t (src/cublas.cu:50-56) and would return
    A^T B; round 3 passed such calls through to the vendor.  fp64_int8_3 is visibly coarse: proof that the call was intercepted."""
    code = textwrap.dedent("""
        import torch
        torch.manual_seed(0)
        a = torch.complex(torch.rand(1024, 1100, dtype=torch.float64, device="cuda") * 2 - 1,
                          torch.rand(1024, 1100, dtype=torch.float64, device="cuda") * 2 - 1)
        b = torch.complex(torch.rand(1024, 1050, dtype=torch.float64, device="cuda") * 2 - 1,
                          torch.rand(1024, 1050, dtype=torch.float64, device="cuda") * 2 - 1)
        c = a.mH @ b
        d = a.mH @ b.mH.mH.clone()
        e = (b.mH @ a).mH            # = a^H b again, through the other operand
        torch.cuda.synchronize()
        ref = a.cpu().mH @ b.cpu()
        wrong = a.cpu().mT @ b.cpu()
        print("MAXDIFF %.3e %.3e %.3e" % ((c.cpu() - ref).abs().max().item(), (e.cpu() - ref).abs().max().item(),
                                          (c.cpu() - wrong).abs().max().item()))
    """)
    diffs = {}
    for mode in ("fp64_int8_3", "fp64_int8_10"):
        e = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
        e.update(LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE=mode, OZIMMU_INFO="1")
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stdout + p.stderr
        line = [l for l in p.stdout.splitlines() if l.startswith("MAXDIFF")][0].split()
        diffs[mode] = tuple(float(x) for x in line[1:4])
    assert diffs["fp64_int8_3"][0] > 1e-6 and diffs["fp64_int8_3"][1] > 1e-6, diffs      # intercepted (3 slices are coarse)
    assert diffs["fp64_int8_10"][0] < 1e-10 and diffs["fp64_int8_10"][1] < 1e-10, diffs    # and conjugated
    assert diffs["fp64_int8_10"][2] > 1.0, diffs                                            # not the plain transpose


def test_pytorch_cuda_graph_capture_through_the_preload():
    """torch.cuda.graph around float64 matmuls under LD_PRELOAD: after an eager warm-up (workspace allocated) the captured
    call is the Ozaki path's kernels (fp64_int8_3 is visibly coarse, so a replay that ran the vendor DGEMM would show);
    a capture WITHOUT warm-up must not break (the shim captures the vendor GEMM for that call instead)"""
    code = textwrap.dedent("""
        import os, torch
        torch.manual_seed(0)
        a = torch.rand(1536, 1024, dtype=torch.float64, device="cuda") * 2 - 1
        b = torch.rand(1024, 1280, dtype=torch.float64, device="cuda") * 2 - 1
        ref = a.cpu() @ b.cpu()
        out = torch.empty(1536, 1280, dtype=torch.float64, device="cuda")
        s = torch.cuda.Stream()
        # the vendor library itself cannot be captured cold (it allocates on first use): warm IT up in pass-through mode
        os.environ["OZIMMU_COMPUTE_MODE"] = "dgemm"   # read on every call
        with torch.cuda.stream(s):
            torch.mm(a, b, out=out)
        torch.cuda.synchronize()
        os.environ["OZIMMU_COMPUTE_MODE"] = "fp64_int8_3"
        # 1) capture before the shim has a workspace -> it hands the call to the vendor GEMM inside the graph
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0, stream=s):
            torch.mm(a, b, out=out)
        out.zero_(); g0.replay(); torch.cuda.synchronize()
        cold = (out.cpu() - ref).abs().max().item()
        # 2) eager warm-up on the capture stream, then capture: the Ozaki kernels are in the graph
        with torch.cuda.stream(s):
            torch.mm(a, b, out=out)
        torch.cuda.synchronize()
        eager = out.clone()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=s):
            torch.mm(a, b, out=out)
        out.zero_(); g1.replay(); torch.cuda.synchronize()
        warm = (out.cpu() - ref).abs().max().item()
        same = torch.equal(out.view(torch.int64), eager.view(torch.int64))
        print("RESULT %.3e %.3e %d" % (cold, warm, int(same)))
    """)
    e = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
    e.update(LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_3")
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr
    cold, warm, same = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][0].split()[1:]
    assert float(cold) < 1e-11, cold          # the vendor DGEMM was captured
    assert float(warm) > 1e-6 and same == "1"  # the 3-slice Ozaki product was captured and replays bit for bit


GETENV_SHIM = r"""
// LD_PRELOAD getenv counter: counts the lookups of OZIMMU* names (the library's per-call environment traffic)
#define _GNU_SOURCE
#include <dlfcn.h>
#include <string.h>
static unsigned long long count;
char *getenv(const char *name) {
  static char *(*real)(const char *);
  if (!real) real = (char *(*)(const char *))dlsym(RTLD_NEXT, "getenv");
  if (name && !strncmp(name, "OZIMMU", 6)) __atomic_add_fetch(&count, 1, __ATOMIC_RELAXED);
  return real(name);
}
unsigned long long oz_getenv_count(void) { return __atomic_load_n(&count, __ATOMIC_RELAXED); }
"""

DRIVER2 = r"""
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  const int n = 512;
  std::vector<double> A((size_t)n * n, 0.5), C((size_t)n * n, 0.0);
  double *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, A.size() * 8); hipMalloc(&dC, A.size() * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
  rocblas_handle h;
  if (rocblas_create_handle(&h) != rocblas_status_success) return 2;
  const double alpha = 1.0, beta = 0.0;
  // argument errors are the vendor's to report: the same status with and without the shim
  printf("STATUS lda_negative %d\n", (int)rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, n, n, n, &alpha,
                                                         dA, -1, dB, n, &beta, dC, n));
  printf("STATUS ldc_zero %d\n", (int)rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, n, n, n, &alpha, dA,
                                                     n, dB, n, &beta, dC, 0));
  printf("STATUS null_a %d\n", (int)rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, n, n, n, &alpha,
                                                   nullptr, n, dB, n, &beta, dC, n));
  printf("STATUS negative_stride %d\n",
         (int)rocblas_dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, n, n, n, &alpha, dA, n, -1, dB,
                                            n, 0, &beta, dC, n, 0, 1));
  hipDeviceSynchronize();
  // environment traffic of an intercepted call, measured by the preloaded getenv counter (if it is there)
  auto counter = (unsigned long long (*)())dlsym(RTLD_DEFAULT, "oz_getenv_count");
  for (int i = 0; i < 3; i++) rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, n, n, n, &alpha, dA, n, dB, n, &beta, dC, n);
  hipDeviceSynchronize();
  const unsigned long long c0 = counter ? counter() : 0;
  const int calls = 20;
  for (int i = 0; i < calls; i++) rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, n, n, n, &alpha, dA, n, dB, n, &beta, dC, n);
  hipDeviceSynchronize();
  const unsigned long long c1 = counter ? counter() : 0;
  hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
  printf("GETENV_PER_CALL %.2f\n", counter ? (double)(c1 - c0) / calls : -1.0);
  printf("C00 %.17g\n", C[0]);
  rocblas_destroy_handle(h);
  return 0;
}
"""


@pytest.fixture(scope="module")
def driver2(tmp_path_factory):
    d = tmp_path_factory.mktemp("gpu_interpose2")
    (d / "driver2.cpp").write_text(DRIVER2)
    (d / "shim.c").write_text(GETENV_SHIM)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(d / "driver2.cpp"),
                           "-o", str(d / "driver2"), "-L/opt/rocm/lib", "-lrocblas", "-lamdhip64", "-ldl", "-pthread",
                           "-Wl,-rpath,/opt/rocm/lib"])
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", str(d / "shim.c"), "-o", str(d / "getenv_shim.so"), "-ldl"])
    return d / "driver2", d / "getenv_shim.so"


def _lines(out, key):
    return {l.split()[1]: l.split()[2] for l in out.splitlines() if l.startswith(key + " ")}


def test_argument_errors_are_reported_by_the_vendor_routine(driver2):
    """lda < 1, ldc = 0, a null matrix, a negative batch stride: the shim declines the call (it would cast them to huge
    size_t values and fault in the kernels) and the application sees exactly the vendor's status"""
    exe, _ = driver2
    thr = dict(OZIMMU_INTERCEPT_THRESHOLD_M=256, OZIMMU_INTERCEPT_THRESHOLD_N=256, OZIMMU_INTERCEPT_THRESHOLD_K=256)
    _, native = run(exe, [])
    _, shim = run(exe, [], LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_9", **thr)
    assert _lines(native, "STATUS") == _lines(shim, "STATUS")
    assert all(int(v) != 0 for k, v in _lines(native, "STATUS").items() if k != "negative_stride")
    assert abs(float(shim.split("C00")[1]) - 128.0) < 1e-9      # the valid calls ran (0.5 * 0.5 * 512)


def test_an_intercepted_call_reads_the_environment_at_most_three_times(driver2):
    """the reference reads OZIMMU_COMPUTE_MODE and the auto threshold per call (src/cublas.cu:18-48, :72-83), CULiP its
    switch (src/culip.cu:41-50); every OZIMMU_HIP_* development switch is read once per process (config.h).  Counted by
    a preloaded getenv that forwards to libc."""
    exe, shim = driver2
    thr = dict(OZIMMU_INTERCEPT_THRESHOLD_M=256, OZIMMU_INTERCEPT_THRESHOLD_N=256, OZIMMU_INTERCEPT_THRESHOLD_K=256)
    _, out = run(exe, [], LD_PRELOAD=f"{shim}:{ozimmu_amd.LIB_PATH}", OZIMMU_COMPUTE_MODE="fp64_int8_9", **thr)
    per_call = float(out.split("GETENV_PER_CALL")[1].split()[0])
    assert 1.0 <= per_call <= 3.0, out


def test_capture_on_a_side_stream_after_eager_calls_on_another_stream():
    """eager matmuls on the default stream, then torch.cuda.graph on a side stream the shim has never seen.
    The captured call cannot be ordered behind the eager work (no synchronisation is legal inside a capture): the shim
    must decline it - the capture stays valid and holds the vendor GEMM - and eager calls afterwards still run the Ozaki
    path with their cross-stream ordering intact."""
    code = textwrap.dedent("""
        import os, torch
        torch.manual_seed(0)
        a = torch.rand(1536, 1024, dtype=torch.float64, device="cuda") * 2 - 1
        b = torch.rand(1024, 1280, dtype=torch.float64, device="cuda") * 2 - 1
        ref = a.cpu() @ b.cpu()
        out = torch.empty(1536, 1280, dtype=torch.float64, device="cuda")
        s = torch.cuda.Stream()
        os.environ["OZIMMU_COMPUTE_MODE"] = "dgemm"
        with torch.cuda.stream(s):
            torch.mm(a, b, out=out)               # vendor warm-up on the capture stream (pass-through)
        torch.cuda.synchronize()
        os.environ["OZIMMU_COMPUTE_MODE"] = "fp64_int8_3"
        for _ in range(3):
            eager = a @ b                          # eager Ozaki calls on the DEFAULT stream: workspace allocated, tail = default
        torch.cuda.synchronize()
        e_err = (eager.cpu() - ref).abs().max().item()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):        # first call the shim sees on s, and s is capturing
            torch.mm(a, b, out=out)
        out.zero_(); g.replay(); torch.cuda.synchronize()
        cap = (out.cpu() - ref).abs().max().item()
        again = a @ b                              # eager again: still the Ozaki path
        torch.cuda.synchronize()
        same = torch.equal(again.view(torch.int64), eager.view(torch.int64))
        print("RESULT %.3e %.3e %d" % (e_err, cap, int(same)))
    """)
    e = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
    e.update(LD_PRELOAD=ozimmu_amd.LIB_PATH, OZIMMU_COMPUTE_MODE="fp64_int8_3")
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr
    e_err, cap, same = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][0].split()[1:]
    assert float(e_err) > 1e-6          # the eager calls ran the 3-slice Ozaki product
    assert float(cap) < 1e-11, cap      # the capture holds the vendor DGEMM (the shim declined), and it is valid
    assert same == "1"
