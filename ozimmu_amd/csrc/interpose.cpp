// interpose.cpp — the LD_PRELOAD drop-in boundary: rocBLAS / hipBLAS DGEMM entry points.
//
// Replaces the reference's cuBLAS hijack layer /root/reference/src/cublas.cu:103-513:
//   cublasCreate_v2 / cublasDestroy_v2   (:104-131)  -> rocblas_create_handle / rocblas_destroy_handle
//   cublasGemmEx                         (:133-278)  -> rocblas_gemm_ex, hipblasGemmEx (all-FP64 real case)
//   cublasDgemm_v2                       (:280-295)  -> rocblas_dgemm, rocblas_dgemm_64, hipblasDgemm
//   cublasDgemmStridedBatched            (:474-492)  -> rocblas_dgemm_strided_batched, hipblasDgemmStridedBatched (sequential loop, :380-406)
//   cublasZgemm_v2                       (:297-313)  -> rocblas_zgemm, hipblasZgemm (and the f64_c case of rocblas_gemm_ex)
//   cublasZgemmStridedBatched            (:494-512)  -> rocblas_zgemm_strided_batched, hipblasZgemmStridedBatched
//   cublasGemmStridedBatchedEx           (:315-472)  -> rocblas_gemm_strided_batched_ex, hipblasGemmStridedBatchedEx
//   sgemm compute mode                   (:169-186, :355-376) -> ozimmu_hip_gemm_f32 (FP32 vendor GEMM on converted copies)
// Originals are found with dlsym(RTLD_NEXT) (src/utils.hpp:117-141).
//
// Documented deviations from the reference (SURVEY.md §8a "quirks"):
//   * `n` is compared with OZIMMU_INTERCEPT_THRESHOLD_N (the reference compares it with _K, src/cublas.cu:145);
//   * the global handle is created lazily on first use and never dereferenced when absent (:144);
//   * one handle PER DEVICE (workspace on the device that runs the GEMM); all are released when the LAST vendor
//     handle created through this shim is destroyed, not on any destroy (:117-126);
//   * an internal failure falls back to the vendor routine instead of reporting SUCCESS (:215-219);
//   * device pointer mode, out-of-place gemm_ex (C != D) and complex types are passed through untouched;
//   * a thread-local guard keeps the shim from re-intercepting calls made underneath itself
//     (hipBLAS -> rocBLAS, and this library's own native-DGEMM fallback).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <hipblas/hipblas.h>
#include <rocblas/rocblas.h>
#include <time.h>

#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "handle.h"

using namespace ozhip;

namespace {

std::mutex g_mtx;
// src/cublas.cu:58 keeps ONE global handle; its workspace lives on whatever device was current at creation, so a
// process that drives several GPUs would hand device-0 memory to device-1 kernels.  One handle per device here.
std::map<int, ozimmu_hip_handle_t> g_handles;
std::atomic<int> g_live_vendor_handles{0};
thread_local int t_depth = 0;

struct DepthGuard {
  DepthGuard() { ++t_depth; }
  ~DepthGuard() { --t_depth; }
};

template <class F> F original(const char *name) {
  return reinterpret_cast<F>(vendor_symbol(name));
}

// src/cublas.cu:18-48
ozimmu_compute_mode_t get_compute_mode() { return ozimmu_hip_compute_mode_from_str(getenv("OZIMMU_COMPUTE_MODE")); }

// src/cublas.cu:60-86
ozimmu_hip_handle_t get_global_handle() {
  std::lock_guard<std::mutex> lock(g_mtx);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0; // no device: ozimmu_hip_create below fails and is reported
  ozimmu_hip_handle_t &g_handle = g_handles[dev];
  if (!g_handle) {
    const ozimmu_malloc_mode_t mm = env_enabled("OZIMMU_MALLOC_ASYNC", false) ? OZIMMU_MALLOC_ASYNC : OZIMMU_MALLOC_SYNC;
    log_info("Initializing ozIMMU handle...");
    if (ozimmu_hip_create(&g_handle, mm) != 0) {
      g_handles.erase(dev);
      return nullptr;
    }
    log_info("Successfully initialized");
  }
  if (const char *thr = getenv("OZIMMU_AUTO_AVG_MANTISSA_LOSS_THRESHOLD")) {
    char *end = nullptr;
    const double v = std::strtod(thr, &end);
    if (end == thr) // the reference throws std::runtime_error here (src/cublas.cu:75-82)
      log_error(std::string("ERROR: invalid value [OZIMMU_AUTO_AVG_MANTISSA_LOSS_THRESHOLD = ") + thr + "]");
    else
      ozimmu_hip_set_auto_mantissa_loss_threashold(g_handle, v);
  }
  return g_handle;
}

bool culip_enabled() { // src/culip.cu:41-50
  const char *v = getenv("OZIMMU_ENABLE_CULIP_PROFILING");
  return v && std::string(v) != "0";
}

ozimmu_operation_t to_oz(rocblas_operation op) { // src/cublas.cu:50-56: every non-N is T
  return op == rocblas_operation_none ? OZIMMU_OP_N : OZIMMU_OP_T;
}
const char *op_str(ozimmu_operation_t op) { return op == OZIMMU_OP_N ? "N" : "T"; }

// The intercept predicate + the Ozaki path.  true = handled (C holds the result).
bool try_ozaki(hipStream_t stream, bool host_pointer_mode, ozimmu_operation_t op_a, ozimmu_operation_t op_b,
               long long m, long long n, long long k, const void *alpha, const void *A, long long lda,
               const void *B, long long ldb, const void *beta, void *C, long long ldc, bool cplx = false) {
  if (t_depth > 0) return false;
  const ozimmu_compute_mode_t mode = get_compute_mode();
  if (mode == OZIMMU_DGEMM) return false;
  if (!host_pointer_mode || m < 0 || n < 0 || k < 0 || !alpha || !beta) return false;
  ozimmu_hip_handle_t h = get_global_handle();
  if (!h) return false;
  // src/cublas.cu:143-148 (with the threshold_n fix)
  if (!((unsigned long long)m >= h->intercept_threshold_m && (unsigned long long)n >= h->intercept_threshold_n &&
        (unsigned long long)k >= h->intercept_threshold_k))
    return false;
  ozimmu_hip_set_stream(h, stream); // src/cublas.cu:149-151

  const bool prof = culip_enabled();
  timespec t0{}, t1{};
  if (prof) { // src/culip.cu:19-39: stream sync, CLOCK_MONOTONIC
    hipStreamSynchronize(stream);
    clock_gettime(CLOCK_MONOTONIC, &t0);
  }
  int err;
  {
    DepthGuard guard; // auto mode may fall back to the vendor DGEMM underneath
    err = ozimmu_hip_gemm(h, op_a, op_b, (size_t)m, (size_t)n, (size_t)k, alpha, A, (size_t)lda, B, (size_t)ldb,
                          beta, C, (size_t)ldc, mode, cplx ? OZIMMU_COMPLX : OZIMMU_REAL);
  }
  if (prof) {
    hipStreamSynchronize(stream);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const unsigned long ns =
        ((long)t1.tv_sec - (long)t0.tv_sec) * 1000000000l + ((long)t1.tv_nsec - (long)t0.tv_nsec);
    // src/cublas.cu:157-162 name format
    std::printf("[CULiP Result][%s%s-%s%s-m%lld-n%lld-k%lld] %luns\n", cplx ? "Z" : "D",
                ozimmu_hip_get_compute_mode_name_str(mode),
                op_str(op_a), op_str(op_b), m, n, k, ns);
    std::fflush(stdout);
  }
  if (err) log_error("Ozaki path failed (status " + std::to_string(err) + "); falling back to the vendor DGEMM");
  return err == 0;
}

bool rocblas_ctx(rocblas_handle handle, hipStream_t *stream, bool *host_mode) {
  typedef rocblas_status (*get_stream_t)(rocblas_handle, hipStream_t *);
  typedef rocblas_status (*get_pm_t)(rocblas_handle, rocblas_pointer_mode *);
  static get_stream_t get_stream = original<get_stream_t>("rocblas_get_stream");
  static get_pm_t get_pm = original<get_pm_t>("rocblas_get_pointer_mode");
  if (!handle || !get_stream || !get_pm) return false;
  rocblas_pointer_mode pm = rocblas_pointer_mode_host;
  if (get_stream(handle, stream) != rocblas_status_success) return false;
  if (get_pm(handle, &pm) != rocblas_status_success) return false;
  *host_mode = pm == rocblas_pointer_mode_host;
  return true;
}

bool hipblas_ctx(hipblasHandle_t handle, hipStream_t *stream, bool *host_mode) {
  typedef hipblasStatus_t (*get_stream_t)(hipblasHandle_t, hipStream_t *);
  typedef hipblasStatus_t (*get_pm_t)(hipblasHandle_t, hipblasPointerMode_t *);
  static get_stream_t get_stream = original<get_stream_t>("hipblasGetStream");
  static get_pm_t get_pm = original<get_pm_t>("hipblasGetPointerMode");
  if (!handle || !get_stream || !get_pm) return false;
  hipblasPointerMode_t pm = HIPBLAS_POINTER_MODE_HOST;
  if (get_stream(handle, stream) != HIPBLAS_STATUS_SUCCESS) return false;
  if (get_pm(handle, &pm) != HIPBLAS_STATUS_SUCCESS) return false;
  *host_mode = pm == HIPBLAS_POINTER_MODE_HOST;
  return true;
}

ozimmu_operation_t hb_to_oz(hipblasOperation_t op) { return op == HIPBLAS_OP_N ? OZIMMU_OP_N : OZIMMU_OP_T; }

// Strided-batched entry points: a sequential loop over the batch like src/cublas.cu:380-406.  `ozaki(i)` runs
// matrix i through try_ozaki; if the predicate rejects the first matrix nothing has been touched and the caller
// forwards the whole batch to the vendor routine (returns -1); if a later matrix fails, the remainder is finished
// with the vendor's non-batched routine `native(i)`.
template <class Ozaki, class Native> int run_batch(long long batch_count, Ozaki ozaki, Native native) {
  long long done = 0;
  for (; done < batch_count; done++)
    if (!ozaki(done)) break;
  if (done == batch_count) return 0;
  if (done == 0) return -1;
  DepthGuard guard;
  for (; done < batch_count; done++)
    if (const int st = native(done)) return st;
  return 0;
}

} // namespace

extern "C" {

// ---- lifecycle (src/cublas.cu:104-131) -----------------------------------------------------------------

rocblas_status rocblas_create_handle(rocblas_handle *handle) {
  typedef rocblas_status (*fn_t)(rocblas_handle *);
  static fn_t fn = original<fn_t>("rocblas_create_handle");
  if (!fn) return rocblas_status_internal_error;
  const rocblas_status st = fn(handle);
  if (st == rocblas_status_success && t_depth == 0) {
    g_live_vendor_handles++;
    // the reference pre-sizes the workspace for a 1024^3 fp64_int8_9 GEMM here (src/cublas.cu:12-16, :109-110)
    const ozimmu_compute_mode_t mode = get_compute_mode();
    if (is_int8_mode(mode) || mode == OZIMMU_FP64_INT8_AUTO)
      if (ozimmu_hip_handle_t h = get_global_handle())
        ozimmu_hip_reallocate_working_memory(
            h, ozimmu_hip_working_memory_size(OZIMMU_OP_N, OZIMMU_OP_N, 1024, 1024, 1024, OZIMMU_REAL,
                                              OZIMMU_FP64_INT8_9));
  }
  return st;
}

rocblas_status rocblas_destroy_handle(rocblas_handle handle) {
  typedef rocblas_status (*fn_t)(rocblas_handle);
  static fn_t fn = original<fn_t>("rocblas_destroy_handle");
  if (!fn) return rocblas_status_internal_error;
  if (t_depth == 0 && g_live_vendor_handles.fetch_sub(1) == 1) {
    std::lock_guard<std::mutex> lock(g_mtx);
    if (!g_handles.empty()) {
      log_info("Destroying ozIMMU handle...");
      int cur = 0;
      hipGetDevice(&cur);
      DepthGuard guard; // ozimmu_hip_destroy releases its private vendor handle through this shim
      for (auto &kv : g_handles) {
        hipSetDevice(kv.first);
        hipDeviceSynchronize(); // the workspace may still be in use by enqueued work
        ozimmu_hip_destroy(kv.second);
      }
      g_handles.clear();
      hipSetDevice(cur);
    }
  }
  return fn(handle);
}

// ---- rocBLAS -------------------------------------------------------------------------------------------

rocblas_status rocblas_dgemm(rocblas_handle handle, rocblas_operation transA, rocblas_operation transB,
                             rocblas_int m, rocblas_int n, rocblas_int k, const double *alpha, const double *A,
                             rocblas_int lda, const double *B, rocblas_int ldb, const double *beta, double *C,
                             rocblas_int ldc) {
  typedef rocblas_status (*fn_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                 rocblas_int, const double *, const double *, rocblas_int, const double *,
                                 rocblas_int, const double *, double *, rocblas_int);
  static fn_t fn = original<fn_t>("rocblas_dgemm");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  if (t_depth == 0 && get_compute_mode() != OZIMMU_DGEMM && rocblas_ctx(handle, &stream, &host_mode) &&
      try_ozaki(stream, host_mode, to_oz(transA), to_oz(transB), m, n, k, alpha, A, lda, B, ldb, beta, C, ldc))
    return rocblas_status_success;
  if (!fn) return rocblas_status_internal_error;
  DepthGuard guard;
  return fn(handle, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

rocblas_status rocblas_dgemm_64(rocblas_handle handle, rocblas_operation transA, rocblas_operation transB,
                                int64_t m, int64_t n, int64_t k, const double *alpha, const double *A, int64_t lda,
                                const double *B, int64_t ldb, const double *beta, double *C, int64_t ldc) {
  typedef rocblas_status (*fn_t)(rocblas_handle, rocblas_operation, rocblas_operation, int64_t, int64_t, int64_t,
                                 const double *, const double *, int64_t, const double *, int64_t, const double *,
                                 double *, int64_t);
  static fn_t fn = original<fn_t>("rocblas_dgemm_64");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  // the fused kernel indexes rows/columns with 32 bits
  const bool fits = m < (1ll << 31) && n < (1ll << 31) && k < (1ll << 30);
  if (t_depth == 0 && fits && get_compute_mode() != OZIMMU_DGEMM && rocblas_ctx(handle, &stream, &host_mode) &&
      try_ozaki(stream, host_mode, to_oz(transA), to_oz(transB), m, n, k, alpha, A, lda, B, ldb, beta, C, ldc))
    return rocblas_status_success;
  if (!fn) return rocblas_status_internal_error;
  DepthGuard guard;
  return fn(handle, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

rocblas_status rocblas_gemm_ex(rocblas_handle handle, rocblas_operation transA, rocblas_operation transB,
                               rocblas_int m, rocblas_int n, rocblas_int k, const void *alpha, const void *a,
                               rocblas_datatype a_type, rocblas_int lda, const void *b, rocblas_datatype b_type,
                               rocblas_int ldb, const void *beta, const void *c, rocblas_datatype c_type,
                               rocblas_int ldc, void *d, rocblas_datatype d_type, rocblas_int ldd,
                               rocblas_datatype compute_type, rocblas_gemm_algo algo, int32_t solution_index,
                               uint32_t flags) {
  typedef rocblas_status (*fn_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                 rocblas_int, const void *, const void *, rocblas_datatype, rocblas_int, const void *,
                                 rocblas_datatype, rocblas_int, const void *, const void *, rocblas_datatype,
                                 rocblas_int, void *, rocblas_datatype, rocblas_int, rocblas_datatype,
                                 rocblas_gemm_algo, int32_t, uint32_t);
  static fn_t fn = original<fn_t>("rocblas_gemm_ex");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  // src/cublas.cu:146-148: all operands FP64 real; in place only (C == D)
  const bool f64 = a_type == rocblas_datatype_f64_r && b_type == rocblas_datatype_f64_r &&
                   c_type == rocblas_datatype_f64_r && d_type == rocblas_datatype_f64_r &&
                   compute_type == rocblas_datatype_f64_r && c == d && ldc == ldd;
  // ... or all FP64 complex (CUDA_C_64F, src/cublas.cu:147-148), without conjugation
  const bool c64 = a_type == rocblas_datatype_f64_c && b_type == rocblas_datatype_f64_c &&
                   c_type == rocblas_datatype_f64_c && d_type == rocblas_datatype_f64_c &&
                   compute_type == rocblas_datatype_f64_c && c == d && ldc == ldd &&
                   transA != rocblas_operation_conjugate_transpose && transB != rocblas_operation_conjugate_transpose;
  if (t_depth == 0 && (f64 || c64) && get_compute_mode() != OZIMMU_DGEMM && rocblas_ctx(handle, &stream, &host_mode) &&
      try_ozaki(stream, host_mode, to_oz(transA), to_oz(transB), m, n, k, alpha, a, lda, b, ldb, beta, d, ldd, c64))
    return rocblas_status_success;
  if (!fn) return rocblas_status_internal_error;
  DepthGuard guard;
  return fn(handle, transA, transB, m, n, k, alpha, a, a_type, lda, b, b_type, ldb, beta, c, c_type, ldc, d, d_type,
            ldd, compute_type, algo, solution_index, flags);
}

rocblas_status rocblas_dgemm_strided_batched(rocblas_handle handle, rocblas_operation transA,
                                             rocblas_operation transB, rocblas_int m, rocblas_int n, rocblas_int k,
                                             const double *alpha, const double *A, rocblas_int lda,
                                             rocblas_stride stride_a, const double *B, rocblas_int ldb,
                                             rocblas_stride stride_b, const double *beta, double *C, rocblas_int ldc,
                                             rocblas_stride stride_c, rocblas_int batch_count) {
  typedef rocblas_status (*fn_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                 rocblas_int, const double *, const double *, rocblas_int, rocblas_stride,
                                 const double *, rocblas_int, rocblas_stride, const double *, double *, rocblas_int,
                                 rocblas_stride, rocblas_int);
  static fn_t fn = original<fn_t>("rocblas_dgemm_strided_batched");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  if (t_depth == 0 && batch_count > 0 && get_compute_mode() != OZIMMU_DGEMM &&
      rocblas_ctx(handle, &stream, &host_mode)) {
    typedef rocblas_status (*dg_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                   rocblas_int, const double *, const double *, rocblas_int, const double *,
                                   rocblas_int, const double *, double *, rocblas_int);
    static dg_t dg = original<dg_t>("rocblas_dgemm");
    const int st = run_batch(
        batch_count,
        [&](long long i) {
          return try_ozaki(stream, host_mode, to_oz(transA), to_oz(transB), m, n, k, alpha, A + i * stride_a, lda,
                           B + i * stride_b, ldb, beta, C + i * stride_c, ldc);
        },
        [&](long long i) {
          return dg ? (int)dg(handle, transA, transB, m, n, k, alpha, A + i * stride_a, lda, B + i * stride_b, ldb,
                              beta, C + i * stride_c, ldc)
                    : (int)rocblas_status_internal_error;
        });
    if (st >= 0) return (rocblas_status)st;
  }
  if (!fn) return rocblas_status_internal_error;
  DepthGuard guard;
  return fn(handle, transA, transB, m, n, k, alpha, A, lda, stride_a, B, ldb, stride_b, beta, C, ldc, stride_c,
            batch_count);
}

// cublasZgemm_v2 (src/cublas.cu:297-313).  Conjugate-transpose is passed through: the reference maps every non-N
// operation to a plain transpose (src/cublas.cu:50-56), which is wrong for complex data.
rocblas_status rocblas_zgemm(rocblas_handle handle, rocblas_operation transA, rocblas_operation transB, rocblas_int m,
                             rocblas_int n, rocblas_int k, const rocblas_double_complex *alpha,
                             const rocblas_double_complex *A, rocblas_int lda, const rocblas_double_complex *B,
                             rocblas_int ldb, const rocblas_double_complex *beta, rocblas_double_complex *C,
                             rocblas_int ldc) {
  typedef rocblas_status (*fn_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                 rocblas_int, const rocblas_double_complex *, const rocblas_double_complex *,
                                 rocblas_int, const rocblas_double_complex *, rocblas_int,
                                 const rocblas_double_complex *, rocblas_double_complex *, rocblas_int);
  static fn_t fn = original<fn_t>("rocblas_zgemm");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  const bool no_conj = transA != rocblas_operation_conjugate_transpose && transB != rocblas_operation_conjugate_transpose;
  if (t_depth == 0 && no_conj && get_compute_mode() != OZIMMU_DGEMM && rocblas_ctx(handle, &stream, &host_mode) &&
      try_ozaki(stream, host_mode, to_oz(transA), to_oz(transB), m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, true))
    return rocblas_status_success;
  if (!fn) return rocblas_status_internal_error;
  DepthGuard guard;
  return fn(handle, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

// cublasZgemmStridedBatched (src/cublas.cu:494-512)
rocblas_status rocblas_zgemm_strided_batched(rocblas_handle handle, rocblas_operation transA,
                                             rocblas_operation transB, rocblas_int m, rocblas_int n, rocblas_int k,
                                             const rocblas_double_complex *alpha, const rocblas_double_complex *A,
                                             rocblas_int lda, rocblas_stride stride_a,
                                             const rocblas_double_complex *B, rocblas_int ldb,
                                             rocblas_stride stride_b, const rocblas_double_complex *beta,
                                             rocblas_double_complex *C, rocblas_int ldc, rocblas_stride stride_c,
                                             rocblas_int batch_count) {
  typedef rocblas_status (*fn_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                 rocblas_int, const rocblas_double_complex *, const rocblas_double_complex *,
                                 rocblas_int, rocblas_stride, const rocblas_double_complex *, rocblas_int,
                                 rocblas_stride, const rocblas_double_complex *, rocblas_double_complex *, rocblas_int,
                                 rocblas_stride, rocblas_int);
  static fn_t fn = original<fn_t>("rocblas_zgemm_strided_batched");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  const bool no_conj = transA != rocblas_operation_conjugate_transpose && transB != rocblas_operation_conjugate_transpose;
  if (t_depth == 0 && no_conj && batch_count > 0 && get_compute_mode() != OZIMMU_DGEMM &&
      rocblas_ctx(handle, &stream, &host_mode)) {
    typedef rocblas_status (*zg_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                   rocblas_int, const rocblas_double_complex *, const rocblas_double_complex *,
                                   rocblas_int, const rocblas_double_complex *, rocblas_int,
                                   const rocblas_double_complex *, rocblas_double_complex *, rocblas_int);
    static zg_t zg = original<zg_t>("rocblas_zgemm");
    const int st = run_batch(
        batch_count,
        [&](long long i) {
          return try_ozaki(stream, host_mode, to_oz(transA), to_oz(transB), m, n, k, alpha, A + i * stride_a, lda,
                           B + i * stride_b, ldb, beta, C + i * stride_c, ldc, true);
        },
        [&](long long i) {
          return zg ? (int)zg(handle, transA, transB, m, n, k, alpha, A + i * stride_a, lda, B + i * stride_b, ldb,
                              beta, C + i * stride_c, ldc)
                    : (int)rocblas_status_internal_error;
        });
    if (st >= 0) return (rocblas_status)st;
  }
  if (!fn) return rocblas_status_internal_error;
  DepthGuard guard;
  return fn(handle, transA, transB, m, n, k, alpha, A, lda, stride_a, B, ldb, stride_b, beta, C, ldc, stride_c,
            batch_count);
}

// cublasGemmStridedBatchedEx (src/cublas.cu:315-472): all-FP64 real or all-FP64 complex, in place (C == D).
// The name is parenthesised because rocblas.h also defines a backward-compatibility macro of the same name.
rocblas_status (rocblas_gemm_strided_batched_ex)(rocblas_handle handle, rocblas_operation transA,
                                                 rocblas_operation transB, rocblas_int m, rocblas_int n,
                                                 rocblas_int k, const void *alpha, const void *a,
                                                 rocblas_datatype a_type, rocblas_int lda, rocblas_stride stride_a,
                                                 const void *b, rocblas_datatype b_type, rocblas_int ldb,
                                                 rocblas_stride stride_b, const void *beta, const void *c,
                                                 rocblas_datatype c_type, rocblas_int ldc, rocblas_stride stride_c,
                                                 void *d, rocblas_datatype d_type, rocblas_int ldd,
                                                 rocblas_stride stride_d, rocblas_int batch_count,
                                                 rocblas_datatype compute_type, rocblas_gemm_algo algo,
                                                 int32_t solution_index, uint32_t flags) {
  typedef rocblas_status (*fn_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                 rocblas_int, const void *, const void *, rocblas_datatype, rocblas_int,
                                 rocblas_stride, const void *, rocblas_datatype, rocblas_int, rocblas_stride,
                                 const void *, const void *, rocblas_datatype, rocblas_int, rocblas_stride, void *,
                                 rocblas_datatype, rocblas_int, rocblas_stride, rocblas_int, rocblas_datatype,
                                 rocblas_gemm_algo, int32_t, uint32_t);
  static fn_t fn = original<fn_t>("rocblas_gemm_strided_batched_ex");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  const bool in_place = c == d && ldc == ldd && stride_c == stride_d;
  const bool f64 = a_type == rocblas_datatype_f64_r && b_type == rocblas_datatype_f64_r &&
                   c_type == rocblas_datatype_f64_r && d_type == rocblas_datatype_f64_r &&
                   compute_type == rocblas_datatype_f64_r && in_place;
  const bool c64 = a_type == rocblas_datatype_f64_c && b_type == rocblas_datatype_f64_c &&
                   c_type == rocblas_datatype_f64_c && d_type == rocblas_datatype_f64_c &&
                   compute_type == rocblas_datatype_f64_c && in_place &&
                   transA != rocblas_operation_conjugate_transpose && transB != rocblas_operation_conjugate_transpose;
  if (t_depth == 0 && (f64 || c64) && batch_count > 0 && get_compute_mode() != OZIMMU_DGEMM &&
      rocblas_ctx(handle, &stream, &host_mode)) {
    typedef rocblas_status (*ex_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                   rocblas_int, const void *, const void *, rocblas_datatype, rocblas_int,
                                   const void *, rocblas_datatype, rocblas_int, const void *, const void *,
                                   rocblas_datatype, rocblas_int, void *, rocblas_datatype, rocblas_int,
                                   rocblas_datatype, rocblas_gemm_algo, int32_t, uint32_t);
    static ex_t ex = original<ex_t>("rocblas_gemm_ex");
    const size_t es = c64 ? 16 : 8;
    const char *ap = (const char *)a, *bp = (const char *)b;
    char *dp = (char *)d;
    const int st = run_batch(
        batch_count,
        [&](long long i) {
          return try_ozaki(stream, host_mode, to_oz(transA), to_oz(transB), m, n, k, alpha, ap + i * stride_a * es, lda,
                           bp + i * stride_b * es, ldb, beta, dp + i * stride_d * es, ldd, c64);
        },
        [&](long long i) {
          return ex ? (int)ex(handle, transA, transB, m, n, k, alpha, ap + i * stride_a * es, a_type, lda,
                              bp + i * stride_b * es, b_type, ldb, beta, dp + i * stride_d * es, c_type, ldc,
                              dp + i * stride_d * es, d_type, ldd, compute_type, algo, solution_index, flags)
                    : (int)rocblas_status_internal_error;
        });
    if (st >= 0) return (rocblas_status)st;
  }
  if (!fn) return rocblas_status_internal_error;
  DepthGuard guard;
  return fn(handle, transA, transB, m, n, k, alpha, a, a_type, lda, stride_a, b, b_type, ldb, stride_b, beta, c,
            c_type, ldc, stride_c, d, d_type, ldd, stride_d, batch_count, compute_type, algo, solution_index, flags);
}

// ---- hipBLAS (only reached when an application binds hipBLAS statically or resolves these first) --------

hipblasStatus_t hipblasDgemm(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int m,
                             int n, int k, const double *alpha, const double *AP, int lda, const double *BP, int ldb,
                             const double *beta, double *CP, int ldc) {
  typedef hipblasStatus_t (*fn_t)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int,
                                  const double *, const double *, int, const double *, int, const double *, double *,
                                  int);
  static fn_t fn = original<fn_t>("hipblasDgemm");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  if (t_depth == 0 && get_compute_mode() != OZIMMU_DGEMM && hipblas_ctx(handle, &stream, &host_mode) &&
      try_ozaki(stream, host_mode, hb_to_oz(transA), hb_to_oz(transB), m, n, k, alpha, AP, lda, BP, ldb, beta, CP,
                ldc))
    return HIPBLAS_STATUS_SUCCESS;
  if (!fn) return HIPBLAS_STATUS_INTERNAL_ERROR;
  // no DepthGuard: the vendor hipblasDgemm calls rocblas_dgemm, which is the normal intercept point
  return fn(handle, transA, transB, m, n, k, alpha, AP, lda, BP, ldb, beta, CP, ldc);
}

hipblasStatus_t hipblasGemmEx(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int m,
                              int n, int k, const void *alpha, const void *A, hipDataType aType, int lda,
                              const void *B, hipDataType bType, int ldb, const void *beta, void *C, hipDataType cType,
                              int ldc, hipblasComputeType_t computeType, hipblasGemmAlgo_t algo) {
  typedef hipblasStatus_t (*fn_t)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int,
                                  const void *, const void *, hipDataType, int, const void *, hipDataType, int,
                                  const void *, void *, hipDataType, int, hipblasComputeType_t, hipblasGemmAlgo_t);
  static fn_t fn = original<fn_t>("hipblasGemmEx");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  const bool f64 = aType == HIP_R_64F && bType == HIP_R_64F && cType == HIP_R_64F && computeType == HIPBLAS_COMPUTE_64F;
  if (t_depth == 0 && f64 && get_compute_mode() != OZIMMU_DGEMM && hipblas_ctx(handle, &stream, &host_mode) &&
      try_ozaki(stream, host_mode, hb_to_oz(transA), hb_to_oz(transB), m, n, k, (const double *)alpha,
                (const double *)A, lda, (const double *)B, ldb, (const double *)beta, (double *)C, ldc))
    return HIPBLAS_STATUS_SUCCESS;
  if (!fn) return HIPBLAS_STATUS_INTERNAL_ERROR;
  return fn(handle, transA, transB, m, n, k, alpha, A, aType, lda, B, bType, ldb, beta, C, cType, ldc, computeType,
            algo);
}

// The remaining hipBLAS twins.  They never run the Ozaki path themselves twice: when the predicate rejects a call
// the vendor hipBLAS routine forwards to the rocBLAS entry point above, which is the normal intercept point.
hipblasStatus_t hipblasZgemm(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int m,
                             int n, int k, const hipDoubleComplex *alpha, const hipDoubleComplex *AP, int lda,
                             const hipDoubleComplex *BP, int ldb, const hipDoubleComplex *beta, hipDoubleComplex *CP,
                             int ldc) {
  typedef hipblasStatus_t (*fn_t)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int,
                                  const hipDoubleComplex *, const hipDoubleComplex *, int, const hipDoubleComplex *,
                                  int, const hipDoubleComplex *, hipDoubleComplex *, int);
  static fn_t fn = original<fn_t>("hipblasZgemm");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  const bool no_conj = transA != HIPBLAS_OP_C && transB != HIPBLAS_OP_C;
  if (t_depth == 0 && no_conj && get_compute_mode() != OZIMMU_DGEMM && hipblas_ctx(handle, &stream, &host_mode) &&
      try_ozaki(stream, host_mode, hb_to_oz(transA), hb_to_oz(transB), m, n, k, alpha, AP, lda, BP, ldb, beta, CP, ldc,
                true))
    return HIPBLAS_STATUS_SUCCESS;
  if (!fn) return HIPBLAS_STATUS_INTERNAL_ERROR;
  return fn(handle, transA, transB, m, n, k, alpha, AP, lda, BP, ldb, beta, CP, ldc);
}

hipblasStatus_t hipblasDgemmStridedBatched(hipblasHandle_t handle, hipblasOperation_t transA,
                                           hipblasOperation_t transB, int m, int n, int k, const double *alpha,
                                           const double *AP, int lda, long long strideA, const double *BP, int ldb,
                                           long long strideB, const double *beta, double *CP, int ldc,
                                           long long strideC, int batchCount) {
  typedef hipblasStatus_t (*fn_t)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int,
                                  const double *, const double *, int, long long, const double *, int, long long,
                                  const double *, double *, int, long long, int);
  static fn_t fn = original<fn_t>("hipblasDgemmStridedBatched");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  if (t_depth == 0 && batchCount > 0 && get_compute_mode() != OZIMMU_DGEMM &&
      hipblas_ctx(handle, &stream, &host_mode)) {
    typedef hipblasStatus_t (*dg_t)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int,
                                    const double *, const double *, int, const double *, int, const double *, double *,
                                    int);
    static dg_t dg = original<dg_t>("hipblasDgemm");
    const int st = run_batch(
        batchCount,
        [&](long long i) {
          return try_ozaki(stream, host_mode, hb_to_oz(transA), hb_to_oz(transB), m, n, k, alpha, AP + i * strideA, lda,
                           BP + i * strideB, ldb, beta, CP + i * strideC, ldc);
        },
        [&](long long i) {
          return dg ? (int)dg(handle, transA, transB, m, n, k, alpha, AP + i * strideA, lda, BP + i * strideB, ldb, beta,
                              CP + i * strideC, ldc)
                    : (int)HIPBLAS_STATUS_INTERNAL_ERROR;
        });
    if (st >= 0) return (hipblasStatus_t)st;
  }
  if (!fn) return HIPBLAS_STATUS_INTERNAL_ERROR;
  return fn(handle, transA, transB, m, n, k, alpha, AP, lda, strideA, BP, ldb, strideB, beta, CP, ldc, strideC,
            batchCount);
}

hipblasStatus_t hipblasZgemmStridedBatched(hipblasHandle_t handle, hipblasOperation_t transA,
                                           hipblasOperation_t transB, int m, int n, int k,
                                           const hipDoubleComplex *alpha, const hipDoubleComplex *AP, int lda,
                                           long long strideA, const hipDoubleComplex *BP, int ldb, long long strideB,
                                           const hipDoubleComplex *beta, hipDoubleComplex *CP, int ldc,
                                           long long strideC, int batchCount) {
  typedef hipblasStatus_t (*fn_t)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int,
                                  const hipDoubleComplex *, const hipDoubleComplex *, int, long long,
                                  const hipDoubleComplex *, int, long long, const hipDoubleComplex *, hipDoubleComplex *,
                                  int, long long, int);
  static fn_t fn = original<fn_t>("hipblasZgemmStridedBatched");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  const bool no_conj = transA != HIPBLAS_OP_C && transB != HIPBLAS_OP_C;
  if (t_depth == 0 && no_conj && batchCount > 0 && get_compute_mode() != OZIMMU_DGEMM &&
      hipblas_ctx(handle, &stream, &host_mode)) {
    typedef hipblasStatus_t (*zg_t)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int,
                                    const hipDoubleComplex *, const hipDoubleComplex *, int, const hipDoubleComplex *,
                                    int, const hipDoubleComplex *, hipDoubleComplex *, int);
    static zg_t zg = original<zg_t>("hipblasZgemm");
    const int st = run_batch(
        batchCount,
        [&](long long i) {
          return try_ozaki(stream, host_mode, hb_to_oz(transA), hb_to_oz(transB), m, n, k, alpha, AP + i * strideA, lda,
                           BP + i * strideB, ldb, beta, CP + i * strideC, ldc, true);
        },
        [&](long long i) {
          return zg ? (int)zg(handle, transA, transB, m, n, k, alpha, AP + i * strideA, lda, BP + i * strideB, ldb, beta,
                              CP + i * strideC, ldc)
                    : (int)HIPBLAS_STATUS_INTERNAL_ERROR;
        });
    if (st >= 0) return (hipblasStatus_t)st;
  }
  if (!fn) return HIPBLAS_STATUS_INTERNAL_ERROR;
  return fn(handle, transA, transB, m, n, k, alpha, AP, lda, strideA, BP, ldb, strideB, beta, CP, ldc, strideC,
            batchCount);
}

hipblasStatus_t hipblasGemmStridedBatchedEx(hipblasHandle_t handle, hipblasOperation_t transA,
                                            hipblasOperation_t transB, int m, int n, int k, const void *alpha,
                                            const void *A, hipDataType aType, int lda, hipblasStride strideA,
                                            const void *B, hipDataType bType, int ldb, hipblasStride strideB,
                                            const void *beta, void *C, hipDataType cType, int ldc,
                                            hipblasStride strideC, int batchCount, hipblasComputeType_t computeType,
                                            hipblasGemmAlgo_t algo) {
  typedef hipblasStatus_t (*fn_t)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int,
                                  const void *, const void *, hipDataType, int, hipblasStride, const void *,
                                  hipDataType, int, hipblasStride, const void *, void *, hipDataType, int,
                                  hipblasStride, int, hipblasComputeType_t, hipblasGemmAlgo_t);
  static fn_t fn = original<fn_t>("hipblasGemmStridedBatchedEx");
  hipStream_t stream = nullptr;
  bool host_mode = false;
  const bool f64 = aType == HIP_R_64F && bType == HIP_R_64F && cType == HIP_R_64F && computeType == HIPBLAS_COMPUTE_64F;
  const bool c64 = aType == HIP_C_64F && bType == HIP_C_64F && cType == HIP_C_64F &&
                   computeType == HIPBLAS_COMPUTE_64F && transA != HIPBLAS_OP_C && transB != HIPBLAS_OP_C;
  if (t_depth == 0 && (f64 || c64) && batchCount > 0 && get_compute_mode() != OZIMMU_DGEMM &&
      hipblas_ctx(handle, &stream, &host_mode)) {
    typedef hipblasStatus_t (*ex_t)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int,
                                    const void *, const void *, hipDataType, int, const void *, hipDataType, int,
                                    const void *, void *, hipDataType, int, hipblasComputeType_t, hipblasGemmAlgo_t);
    static ex_t ex = original<ex_t>("hipblasGemmEx");
    const size_t es = c64 ? 16 : 8;
    const char *ap = (const char *)A, *bp = (const char *)B;
    char *cp = (char *)C;
    const int st = run_batch(
        batchCount,
        [&](long long i) {
          return try_ozaki(stream, host_mode, hb_to_oz(transA), hb_to_oz(transB), m, n, k, alpha, ap + i * strideA * es,
                           lda, bp + i * strideB * es, ldb, beta, cp + i * strideC * es, ldc, c64);
        },
        [&](long long i) {
          return ex ? (int)ex(handle, transA, transB, m, n, k, alpha, ap + i * strideA * es, aType, lda,
                              bp + i * strideB * es, bType, ldb, beta, cp + i * strideC * es, cType, ldc, computeType,
                              algo)
                    : (int)HIPBLAS_STATUS_INTERNAL_ERROR;
        });
    if (st >= 0) return (hipblasStatus_t)st;
  }
  if (!fn) return HIPBLAS_STATUS_INTERNAL_ERROR;
  return fn(handle, transA, transB, m, n, k, alpha, A, aType, lda, strideA, B, bType, ldb, strideB, beta, C, cType, ldc,
            strideC, batchCount, computeType, algo);
}

} // extern "C"
