"""CPU tests of the LD_PRELOAD boundary (ozimmu_amd/csrc/interpose.cpp) without a GPU.

A stub `librocblas.so.5` that records calls stands in for the vendor library; a small C driver linked against
it calls rocblas_dgemm / rocblas_gemm_ex / rocblas_{d,z}gemm_strided_batched / rocblas_gemm_strided_batched_ex and their
ILP64 `_64` twins under LD_PRELOAD=libozimmu_hip.so.
Checked: the shim's definitions win symbol resolution, the pass-through reaches the vendor routine with the
arguments untouched, and the intercept predicate (src/cublas.cu:142-148: mode, thresholds, types) decides as
documented.  With an Ozaki mode selected and sizes above the thresholds the shim tries its own path, finds
no GPU (ozimmu_hip_create fails), logs, and falls back to the vendor routine -- never crashing the caller.
"""
import os
import subprocess
import textwrap

import pytest

import ozimmu_amd

STUB_C = r"""
#include <stdio.h>
#include <stdint.h>
typedef struct _h { int pm; void* stream; } *rocblas_handle;
static struct _h the_handle = {0, 0};
int rocblas_create_handle(rocblas_handle* h) { *h = &the_handle; printf("STUB create\n"); return 0; }
int rocblas_destroy_handle(rocblas_handle h) { (void)h; printf("STUB destroy\n"); return 0; }
int rocblas_get_stream(rocblas_handle h, void** s) { *s = h->stream; return 0; }
int rocblas_set_stream(rocblas_handle h, void* s) { h->stream = s; return 0; }
int rocblas_get_pointer_mode(rocblas_handle h, int* pm) { *pm = h->pm; return 0; }
int rocblas_set_pointer_mode(rocblas_handle h, int pm) { h->pm = pm; return 0; }
int rocblas_dgemm(rocblas_handle h, int ta, int tb, int m, int n, int k, const double* al, const double* A, int lda,
                  const double* B, int ldb, const double* be, double* C, int ldc) {
  (void)h; (void)A; (void)B; (void)C;
  printf("STUB dgemm ta=%d tb=%d m=%d n=%d k=%d alpha=%g lda=%d ldb=%d beta=%g ldc=%d\n", ta, tb, m, n, k, *al, lda, ldb,
         *be, ldc);
  return 0;
}
int rocblas_dgemm_64(rocblas_handle h, int ta, int tb, int64_t m, int64_t n, int64_t k, const double* al, const double* A,
                     int64_t lda, const double* B, int64_t ldb, const double* be, double* C, int64_t ldc) {
  (void)h; (void)A; (void)B; (void)C; (void)al; (void)be; (void)lda; (void)ldb; (void)ldc; (void)ta; (void)tb;
  printf("STUB dgemm_64 m=%lld n=%lld k=%lld\n", (long long)m, (long long)n, (long long)k);
  return 0;
}
int rocblas_gemm_ex(rocblas_handle h, int ta, int tb, int m, int n, int k, const void* al, const void* a, int at, int lda,
                    const void* b, int bt, int ldb, const void* be, const void* c, int ct, int ldc, void* d, int dt,
                    int ldd, int compute, int algo, int32_t sol, uint32_t flags) {
  (void)h; (void)al; (void)a; (void)b; (void)be; (void)c; (void)d; (void)lda; (void)ldb; (void)ldc; (void)ldd;
  (void)algo; (void)sol; (void)flags; (void)ta; (void)tb;
  printf("STUB gemm_ex m=%d n=%d k=%d types=%d,%d,%d,%d compute=%d\n", m, n, k, at, bt, ct, dt, compute);
  return 0;
}
int rocblas_dgemm_strided_batched(rocblas_handle h, int ta, int tb, int m, int n, int k, const double* al, const double* A,
                                  int lda, long long sa, const double* B, int ldb, long long sb, const double* be, double* C,
                                  int ldc, long long sc, int batch) {
  (void)h; (void)A; (void)B; (void)C; (void)al; (void)be; (void)lda; (void)ldb; (void)ldc; (void)sa; (void)sb; (void)sc;
  (void)ta; (void)tb;
  printf("STUB dgemm_strided_batched m=%d n=%d k=%d batch=%d\n", m, n, k, batch);
  return 0;
}
int rocblas_zgemm_strided_batched(rocblas_handle h, int ta, int tb, int m, int n, int k, const void* al, const void* A,
                                  int lda, long long sa, const void* B, int ldb, long long sb, const void* be, void* C,
                                  int ldc, long long sc, int batch) {
  (void)h; (void)A; (void)B; (void)C; (void)al; (void)be; (void)lda; (void)ldb; (void)ldc; (void)ta; (void)tb;
  printf("STUB zgemm_strided_batched m=%d n=%d k=%d strides=%lld,%lld,%lld batch=%d\n", m, n, k, sa, sb, sc, batch);
  return 0;
}
int rocblas_zgemm_64(rocblas_handle h, int ta, int tb, int64_t m, int64_t n, int64_t k, const void* al, const void* A,
                     int64_t lda, const void* B, int64_t ldb, const void* be, void* C, int64_t ldc) {
  (void)h; (void)A; (void)B; (void)C; (void)al; (void)be; (void)lda; (void)ldb; (void)ldc; (void)ta; (void)tb;
  printf("STUB zgemm_64 m=%lld n=%lld k=%lld\n", (long long)m, (long long)n, (long long)k);
  return 0;
}
int rocblas_gemm_ex_64(rocblas_handle h, int ta, int tb, int64_t m, int64_t n, int64_t k, const void* al, const void* a,
                       int at, int64_t lda, const void* b, int bt, int64_t ldb, const void* be, const void* c, int ct,
                       int64_t ldc, void* d, int dt, int64_t ldd, int compute, int algo, int32_t sol, uint32_t flags) {
  (void)h; (void)al; (void)a; (void)b; (void)be; (void)c; (void)d; (void)lda; (void)ldb; (void)ldc; (void)ldd;
  (void)algo; (void)sol; (void)flags; (void)ta; (void)tb;
  printf("STUB gemm_ex_64 m=%lld n=%lld k=%lld types=%d,%d,%d,%d compute=%d\n", (long long)m, (long long)n, (long long)k,
         at, bt, ct, dt, compute);
  return 0;
}
int rocblas_dgemm_strided_batched_64(rocblas_handle h, int ta, int tb, int64_t m, int64_t n, int64_t k, const double* al,
                                     const double* A, int64_t lda, long long sa, const double* B, int64_t ldb,
                                     long long sb, const double* be, double* C, int64_t ldc, long long sc, int64_t batch) {
  (void)h; (void)A; (void)B; (void)C; (void)al; (void)be; (void)lda; (void)ldb; (void)ldc; (void)ta; (void)tb;
  printf("STUB dgemm_strided_batched_64 m=%lld n=%lld k=%lld strides=%lld,%lld,%lld batch=%lld\n", (long long)m,
         (long long)n, (long long)k, sa, sb, sc, (long long)batch);
  return 0;
}
int rocblas_zgemm_strided_batched_64(rocblas_handle h, int ta, int tb, int64_t m, int64_t n, int64_t k, const void* al,
                                     const void* A, int64_t lda, long long sa, const void* B, int64_t ldb, long long sb,
                                     const void* be, void* C, int64_t ldc, long long sc, int64_t batch) {
  (void)h; (void)A; (void)B; (void)C; (void)al; (void)be; (void)lda; (void)ldb; (void)ldc; (void)ta; (void)tb;
  printf("STUB zgemm_strided_batched_64 m=%lld n=%lld k=%lld strides=%lld,%lld,%lld batch=%lld\n", (long long)m,
         (long long)n, (long long)k, sa, sb, sc, (long long)batch);
  return 0;
}
int rocblas_gemm_strided_batched_ex_64(rocblas_handle h, int ta, int tb, int64_t m, int64_t n, int64_t k, const void* al,
                                       const void* a, int at, int64_t lda, long long sa, const void* b, int bt,
                                       int64_t ldb, long long sb, const void* be, const void* c, int ct, int64_t ldc,
                                       long long sc, void* d, int dt, int64_t ldd, long long sd, int64_t batch,
                                       int compute, int algo, int32_t sol, uint32_t flags) {
  (void)h; (void)al; (void)a; (void)b; (void)be; (void)c; (void)d; (void)lda; (void)ldb; (void)ldc; (void)ldd;
  (void)algo; (void)sol; (void)flags; (void)ta; (void)tb;
  printf("STUB gemm_strided_batched_ex_64 m=%lld n=%lld k=%lld types=%d,%d,%d,%d strides=%lld,%lld,%lld,%lld batch=%lld "
         "compute=%d\n", (long long)m, (long long)n, (long long)k, at, bt, ct, dt, sa, sb, sc, sd, (long long)batch, compute);
  return 0;
}
int rocblas_gemm_strided_batched_ex(rocblas_handle h, int ta, int tb, int m, int n, int k, const void* al, const void* a,
                                    int at, int lda, long long sa, const void* b, int bt, int ldb, long long sb,
                                    const void* be, const void* c, int ct, int ldc, long long sc, void* d, int dt, int ldd,
                                    long long sd, int batch, int compute, int algo, int32_t sol, uint32_t flags) {
  (void)h; (void)al; (void)a; (void)b; (void)be; (void)c; (void)d; (void)lda; (void)ldb; (void)ldc; (void)ldd;
  (void)algo; (void)sol; (void)flags; (void)ta; (void)tb;
  printf("STUB gemm_strided_batched_ex m=%d n=%d k=%d types=%d,%d,%d,%d strides=%lld,%lld,%lld,%lld batch=%d compute=%d\n",
         m, n, k, at, bt, ct, dt, sa, sb, sc, sd, batch, compute);
  return 0;
}
"""

DRIVER_C = r"""
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef void* rocblas_handle;
int rocblas_create_handle(rocblas_handle*);
int rocblas_destroy_handle(rocblas_handle);
int rocblas_set_pointer_mode(rocblas_handle, int);
int rocblas_dgemm(rocblas_handle, int, int, int, int, int, const double*, const double*, int, const double*, int,
                  const double*, double*, int);
int rocblas_dgemm_64(rocblas_handle, int, int, int64_t, int64_t, int64_t, const double*, const double*, int64_t,
                     const double*, int64_t, const double*, double*, int64_t);
int rocblas_gemm_ex(rocblas_handle, int, int, int, int, int, const void*, const void*, int, int, const void*, int, int,
                    const void*, const void*, int, int, void*, int, int, int, int, int32_t, uint32_t);
int rocblas_dgemm_strided_batched(rocblas_handle, int, int, int, int, int, const double*, const double*, int, long long,
                                  const double*, int, long long, const double*, double*, int, long long, int);
int rocblas_zgemm_strided_batched(rocblas_handle, int, int, int, int, int, const void*, const void*, int, long long,
                                  const void*, int, long long, const void*, void*, int, long long, int);
int rocblas_gemm_strided_batched_ex(rocblas_handle, int, int, int, int, int, const void*, const void*, int, int, long long,
                                    const void*, int, int, long long, const void*, const void*, int, int, long long, void*,
                                    int, int, long long, int, int, int, int32_t, uint32_t);
int rocblas_zgemm_64(rocblas_handle, int, int, int64_t, int64_t, int64_t, const void*, const void*, int64_t, const void*,
                     int64_t, const void*, void*, int64_t);
int rocblas_gemm_ex_64(rocblas_handle, int, int, int64_t, int64_t, int64_t, const void*, const void*, int, int64_t,
                       const void*, int, int64_t, const void*, const void*, int, int64_t, void*, int, int64_t, int, int,
                       int32_t, uint32_t);
int rocblas_dgemm_strided_batched_64(rocblas_handle, int, int, int64_t, int64_t, int64_t, const double*, const double*,
                                     int64_t, long long, const double*, int64_t, long long, const double*, double*,
                                     int64_t, long long, int64_t);
int rocblas_zgemm_strided_batched_64(rocblas_handle, int, int, int64_t, int64_t, int64_t, const void*, const void*,
                                     int64_t, long long, const void*, int64_t, long long, const void*, void*, int64_t,
                                     long long, int64_t);
int rocblas_gemm_strided_batched_ex_64(rocblas_handle, int, int, int64_t, int64_t, int64_t, const void*, const void*, int,
                                       int64_t, long long, const void*, int, int64_t, long long, const void*, const void*,
                                       int, int64_t, long long, void*, int, int64_t, long long, int64_t, int, int,
                                       int32_t, uint32_t);
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 64;
  int device_mode = argc > 2 ? atoi(argv[2]) : 0;
  rocblas_handle h;
  rocblas_create_handle(&h);
  if (device_mode) rocblas_set_pointer_mode(h, 1);
  double alpha = 1.5, beta = 0.25;
  double* fake = (double*)0x1000;  /* never dereferenced by the stub; 8-byte aligned */
  int st = rocblas_dgemm(h, 111, 112, n, n + 1, n + 2, &alpha, fake, n, fake, n + 1, &beta, fake, n);
  printf("APP dgemm status=%d\n", st);
  st = rocblas_dgemm_64(h, 111, 111, n, n, n, &alpha, fake, n, fake, n, &beta, fake, n);
  printf("APP dgemm_64 status=%d\n", st);
  st = rocblas_gemm_ex(h, 111, 111, n, n, n, &alpha, fake, 152, n, fake, 152, n, &beta, fake, 152, n, fake, 152, n, 152, 0, 0, 0);
  printf("APP gemm_ex f64 status=%d\n", st);
  st = rocblas_gemm_ex(h, 111, 111, n, n, n, &alpha, fake, 151, n, fake, 151, n, &beta, fake, 151, n, fake, 151, n, 151, 0, 0, 0);
  printf("APP gemm_ex f32 status=%d\n", st);
  st = rocblas_dgemm_strided_batched(h, 111, 111, n, n, n, &alpha, fake, n, n * n, fake, n, n * n, &beta, fake, n, n * n, 3);
  printf("APP strided status=%d\n", st);
  double zalpha[2] = {1.0, 0.5}, zbeta[2] = {0.0, 0.0};
  st = rocblas_zgemm_strided_batched(h, 111, 112, n, n, n, zalpha, fake, n, n * n, fake, n, 2 * n * n, zbeta, fake, n,
                                     3 * n * n, 2);
  printf("APP zstrided status=%d\n", st);
  st = rocblas_gemm_strided_batched_ex(h, 111, 111, n, n, n, &alpha, fake, 152, n, n * n, fake, 152, n, n * n, &beta, fake,
                                       152, n, n * n, fake, 152, n, n * n, 4, 152, 0, 0, 0);
  printf("APP strided_ex status=%d\n", st);
  /* the ILP64 twins (libhipblas and ILP64 applications bind these) */
  st = rocblas_zgemm_64(h, 111, 111, n, n, n, zalpha, fake, n, fake, n, zbeta, fake, n);
  printf("APP zgemm_64 status=%d\n", st);
  st = rocblas_gemm_ex_64(h, 111, 111, n, n, n, &alpha, fake, 152, n, fake, 152, n, &beta, fake, 152, n, fake, 152, n, 152,
                          0, 0, 0);
  printf("APP gemm_ex_64 status=%d\n", st);
  st = rocblas_dgemm_strided_batched_64(h, 111, 111, n, n, n, &alpha, fake, n, n * n, fake, n, n * n, &beta, fake, n,
                                        n * n, 3);
  printf("APP strided_64 status=%d\n", st);
  st = rocblas_zgemm_strided_batched_64(h, 111, 111, n, n, n, zalpha, fake, n, n * n, fake, n, n * n, zbeta, fake, n,
                                        n * n, 2);
  printf("APP zstrided_64 status=%d\n", st);
  st = rocblas_gemm_strided_batched_ex_64(h, 111, 111, n, n, n, zalpha, fake, 153, n, n * n, fake, 153, n, n * n, zbeta,
                                          fake, 153, n, n * n, fake, 153, n, n * n, 2, 153, 0, 0, 0);
  printf("APP strided_ex_64 status=%d\n", st);
  rocblas_destroy_handle(h);
  return 0;
}
"""


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    from ozimmu_amd import build
    build.build()
    d = tmp_path_factory.mktemp("interpose")
    (d / "stub.c").write_text(STUB_C)
    (d / "driver.c").write_text(DRIVER_C)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-Wl,-soname,librocblas.so.5", "-o",
                           str(d / "librocblas.so.5"), str(d / "stub.c")])
    os.symlink(str(d / "librocblas.so.5"), str(d / "librocblas.so"))
    subprocess.check_call(["gcc", "-O1", "-o", str(d / "driver"), str(d / "driver.c"), "-L" + str(d), "-lrocblas"])
    return d


def run(d, n=64, device_mode=0, preload=True, **env):
    e = dict(os.environ)
    for k in list(e):
        if k.startswith("OZIMMU_"):
            del e[k]
    e["LD_LIBRARY_PATH"] = str(d) + ":" + e.get("LD_LIBRARY_PATH", "")
    if preload:
        e["LD_PRELOAD"] = ozimmu_amd.LIB_PATH
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([str(d / "driver"), str(n), str(device_mode)], env=e, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


EXPECT_PASSTHROUGH = textwrap.dedent("""\
    STUB create
    STUB dgemm ta=111 tb=112 m={n} n={n1} k={n2} alpha=1.5 lda={n} ldb={n1} beta=0.25 ldc={n}
    APP dgemm status=0
    STUB dgemm_64 m={n} n={n} k={n}
    APP dgemm_64 status=0
    STUB gemm_ex m={n} n={n} k={n} types=152,152,152,152 compute=152
    APP gemm_ex f64 status=0
    STUB gemm_ex m={n} n={n} k={n} types=151,151,151,151 compute=151
    APP gemm_ex f32 status=0
    STUB dgemm_strided_batched m={n} n={n} k={n} batch=3
    APP strided status=0
    STUB zgemm_strided_batched m={n} n={n} k={n} strides={nn},{nn2},{nn3} batch=2
    APP zstrided status=0
    STUB gemm_strided_batched_ex m={n} n={n} k={n} types=152,152,152,152 strides={nn},{nn},{nn},{nn} batch=4 compute=152
    APP strided_ex status=0
    STUB zgemm_64 m={n} n={n} k={n}
    APP zgemm_64 status=0
    STUB gemm_ex_64 m={n} n={n} k={n} types=152,152,152,152 compute=152
    APP gemm_ex_64 status=0
    STUB dgemm_strided_batched_64 m={n} n={n} k={n} strides={nn},{nn},{nn} batch=3
    APP strided_64 status=0
    STUB zgemm_strided_batched_64 m={n} n={n} k={n} strides={nn},{nn},{nn} batch=2
    APP zstrided_64 status=0
    STUB gemm_strided_batched_ex_64 m={n} n={n} k={n} types=153,153,153,153 strides={nn},{nn},{nn},{nn} batch=2 compute=153
    APP strided_ex_64 status=0
    STUB destroy
    """)


def expect(n):
    return EXPECT_PASSTHROUGH.format(n=n, n1=n + 1, n2=n + 2, nn=n * n, nn2=2 * n * n, nn3=3 * n * n)


def test_driver_without_preload(harness):
    assert run(harness, preload=False) == expect(64)


def test_preload_is_transparent_when_mode_unset_or_dgemm(harness):
    """src/cublas.cu:18-48: unset / unknown / dgemm -> bit-transparent pass-through"""
    assert run(harness) == expect(64)
    assert run(harness, OZIMMU_COMPUTE_MODE="dgemm") == expect(64)
    assert run(harness, OZIMMU_COMPUTE_MODE="no_such_mode") == expect(64)


def test_sgemm_mode_is_an_intercepting_mode(harness):
    """src/cublas.cu:169-186: `sgemm` goes through the same predicate as the int8 modes (here: no GPU -> the handle
    cannot be created -> logged -> forwarded to the vendor routine)"""
    out = run(harness, OZIMMU_COMPUTE_MODE="sgemm")
    assert "[ozIMMU ERROR]" in out
    assert [l for l in out.splitlines() if not l.startswith("[ozIMMU")] == expect(64).splitlines()


def test_below_threshold_passes_through_without_touching_the_gpu(harness):
    """src/cublas.cu:143-148 + src/handle.cu:25-30: default thresholds 1024.  The handle is created lazily
    (rocblas_create_handle hook), fails without a GPU, is logged, and the call is forwarded."""
    out = run(harness, n=64, OZIMMU_COMPUTE_MODE="fp64_int8_9", OZIMMU_ERROR="0")
    assert out == expect(64)
    out = run(harness, n=64, OZIMMU_COMPUTE_MODE="fp64_int8_9")
    assert "[ozIMMU ERROR]" in out                      # ozIMMU_error is on by default (src/utils.hpp:106-115)
    assert [l for l in out.splitlines() if not l.startswith("[ozIMMU")] == expect(64).splitlines()


def test_above_threshold_without_gpu_falls_back_to_vendor(harness):
    """intercept predicate true (thresholds lowered with the env vars) but no device: the shim must not crash
    and must not report success without computing -- it forwards to the vendor routine"""
    out = run(harness, n=64, OZIMMU_COMPUTE_MODE="fp64_int8_6", OZIMMU_INTERCEPT_THRESHOLD_M=16,
              OZIMMU_INTERCEPT_THRESHOLD_N=16, OZIMMU_INTERCEPT_THRESHOLD_K=16, OZIMMU_ERROR=0, OZIMMU_INFO=1)
    lines = [l for l in out.splitlines() if not l.startswith("[ozIMMU")]
    assert lines == expect(64).splitlines()
    assert "[ozIMMU LOG] Initializing ozIMMU handle" in out   # src/handle.cu:8, src/cublas.cu:66


def test_device_pointer_mode_is_never_intercepted(harness):
    out = run(harness, n=64, device_mode=1, OZIMMU_COMPUTE_MODE="fp64_int8_6", OZIMMU_INTERCEPT_THRESHOLD_M=1,
              OZIMMU_INTERCEPT_THRESHOLD_N=1, OZIMMU_INTERCEPT_THRESHOLD_K=1, OZIMMU_ERROR=0)
    assert [l for l in out.splitlines() if not l.startswith("[ozIMMU")] == expect(64).splitlines()


def test_info_logging_off_by_default(harness):
    out = run(harness, OZIMMU_COMPUTE_MODE="fp64_int8_9", OZIMMU_ERROR=0)
    assert "[ozIMMU LOG]" not in out                       # src/utils.hpp:88-104
    out = run(harness, OZIMMU_COMPUTE_MODE="fp64_int8_9", OZIMMU_ERROR=0, OZIMMU_INFO=0)
    assert "[ozIMMU LOG]" not in out
