// interpose.cpp — the LD_PRELOAD drop-in boundary: rocBLAS / hipBLAS FP64 GEMM entry points.
//
// Replaces the reference's cuBLAS hijack layer /root/reference/src/cublas.cu:103-513:
//   cublasCreate_v2 / cublasDestroy_v2   (:104-131)  -> rocblas_create_handle / rocblas_destroy_handle
//   cublasGemmEx                         (:133-278)  -> rocblas_gemm_ex[_64], hipblasGemmEx[_64], hipblasGemmExWithFlags[_64]
//   cublasDgemm_v2                       (:280-295)  -> rocblas_dgemm[_64], hipblasDgemm[_64]
//   cublasZgemm_v2                       (:297-313)  -> rocblas_zgemm[_64], hipblasZgemm[_64] (and the f64_c case of gemm_ex)
//   cublasGemmStridedBatchedEx           (:315-472)  -> rocblas_gemm_strided_batched_ex[_64], hipblasGemmStridedBatchedEx[_64],
//                                                       hipblasGemmStridedBatchedExWithFlags[_64]
//   cublasDgemmStridedBatched            (:474-492)  -> rocblas_dgemm_strided_batched[_64], hipblasDgemmStridedBatched[_64]
//   cublasZgemmStridedBatched            (:494-512)  -> rocblas_zgemm_strided_batched[_64], hipblasZgemmStridedBatched[_64]
//   sgemm compute mode                   (:169-186, :355-376) -> ozimmu_hip_gemm_f32 (FP32 vendor GEMM on converted copies)
// Originals are found with dlsym(RTLD_NEXT) (src/utils.hpp:117-141).  The ILP64 (`_64`) twins matter because libhipblas
// imports the rocBLAS `_64` symbols and HPL-style ILP64 builds call them directly: without them such callers would
// bypass the shim silently.
//
// Documented deviations from the reference (SURVEY.md §8a "quirks"):
//   * `n` is compared with OZIMMU_INTERCEPT_THRESHOLD_N (the reference compares it with _K, src/cublas.cu:145);
//   * the global handle is created lazily on first use and never dereferenced when absent (:144);
//   * one handle PER DEVICE (workspace on the device that runs the GEMM); all are released when the LAST vendor
//     handle created through this shim is destroyed, not on any destroy (:117-126);
//   * the strided-batched entry points run the whole batch through ONE set of launches (ozimmu_hip_gemm_strided_batched)
//     instead of the reference's sequential per-matrix loop (:380-406);
//   * a failure BEFORE C was touched falls back to the vendor routine; a failure after C was modified is reported as an
//     error status (the reference overwrites every status with SUCCESS, :215-219) -- never a silent double update;
//   * device pointer mode and out-of-place gemm_ex (C != D) are passed through; conjugate-transposed complex operands are
//     computed as such (OZIMMU_OP_C), not as plain transposes (:50-56);
//   * a thread-local guard keeps the shim from re-intercepting calls made underneath itself
//     (hipBLAS -> rocBLAS, and this library's own native-DGEMM fallback);
//   * the caller's stream travels WITH the call (ozimmu_hip_gemm_on_stream takes the handle's lock before it switches
//     streams), so host threads that share a device cannot enqueue on each other's streams.
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

#include "config.h"
#include "diagnostics.h"
#include "handle.h"

using namespace ozhip;

namespace {

std::mutex g_mtx;
// src/cublas.cu:58 keeps ONE global handle; its workspace lives on whatever device was current at creation, so a
// process that drives several GPUs would hand device-0 memory to device-1 kernels.  One handle per device here.
std::map<int, ozimmu_hip_handle_t> g_handles;
std::atomic<int> g_live_vendor_handles{0};
// ozimmu_hip_intercept_stats: FP64 GEMM calls that reached an interposed entry point with a compute mode other than `dgemm`
// / that ran on the Ozaki path / that were left to the vendor routine (predicate, thresholds, status 3) / that failed (status 4)
std::atomic<unsigned long long> g_stat_seen{0}, g_stat_taken{0}, g_stat_declined{0}, g_stat_failed{0};
thread_local int t_depth = 0;

struct DepthGuard {
  DepthGuard() { ++t_depth; }
  ~DepthGuard() { --t_depth; }
};

template <class F> F original(const char *name) { return reinterpret_cast<F>(vendor_symbol(name)); }
// the vendor's definition of the entry point being defined: same prototype, next in the lookup order
#define OZ_ORIGINAL(name) static const auto fn = original<decltype(&name)>(#name)

// src/cublas.cu:18-48
ozimmu_compute_mode_t get_compute_mode() { return ozimmu_hip_compute_mode_from_str(counted_getenv("OZIMMU_COMPUTE_MODE")); }

// src/cublas.cu:60-86
// `stream`: the caller's stream.  The handle of a device is created on its first intercepted call (device allocations);
// if that call happens while its stream is being captured into a graph, nothing may be allocated: the call is left to the
// vendor routine and the handle is created by a later, uncaptured call.
ozimmu_hip_handle_t get_global_handle(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mtx);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0; // no device: ozimmu_hip_create below fails and is reported
  if (g_handles.find(dev) == g_handles.end() || !g_handles[dev]) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return nullptr;
    (void)hipGetLastError();
  }
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
  if (const char *thr = counted_getenv("OZIMMU_AUTO_AVG_MANTISSA_LOSS_THRESHOLD")) {
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
  const char *v = counted_getenv("OZIMMU_ENABLE_CULIP_PROFILING");
  return v && std::string(v) != "0";
}

const char *op_str(ozimmu_operation_t op) { return op == OZIMMU_OP_N ? "N" : "T"; }

// outcome of an attempt to run a call on the Ozaki path
enum class Try {
  NotTaken, // predicate rejected it, or the path failed before C was touched: the vendor routine may run
  Done,     // C holds the result
  Failed    // the path failed after C had been modified: report an error, never fall back
};

struct GemmCall { // one (possibly strided-batched) GEMM in BLAS terms; strides in elements of the operand type
  ozimmu_operation_t op_a, op_b;
  long long m, n, k;
  const void *alpha, *A;
  long long lda, stride_a;
  const void *B;
  long long ldb, stride_b;
  const void *beta;
  void *C;
  long long ldc, stride_c;
  long long batch; // 1 for the non-batched entry points
  bool cplx;
  bool strided_batched_entry; // reached through a *StridedBatched* entry point (CULiP tag)
};

// The intercept predicate + the Ozaki path.
Try try_ozaki(hipStream_t stream, bool host_pointer_mode, const GemmCall &g, ozimmu_compute_mode_t mode) {
  if (t_depth > 0) return Try::NotTaken;
  if (mode == OZIMMU_DGEMM) return Try::NotTaken;
  if (!host_pointer_mode || g.m < 0 || g.n < 0 || g.k < 0 || g.batch < 1 || !g.alpha || !g.beta) return Try::NotTaken;
  // Argument errors are the vendor routine's to report (invalid_size / invalid_pointer): a leading dimension below 1 or a
  // negative batch stride would be cast to a huge size_t and fault in the kernels, a null matrix likewise.  (The reference
  // inherits these checks from the cuBLAS calls underneath it.)
  if (g.lda < 1 || g.ldb < 1 || g.ldc < 1 || g.stride_a < 0 || g.stride_b < 0 || g.stride_c < 0) return Try::NotTaken;
  if (g.m > 0 && g.n > 0 && (!g.C || (g.k > 0 && (!g.A || !g.B)))) return Try::NotTaken;
  // the kernels index rows / columns with 32 bits and the slice width is defined up to k = 2^30 (src/split.cu:520-536)
  if (g.m >= (1ll << 31) || g.n >= (1ll << 31) || g.k > (1ll << 30) || g.batch >= (1ll << 31)) return Try::NotTaken;
  ozimmu_hip_handle_t h = get_global_handle(stream);
  if (!h) return Try::NotTaken;
  // src/cublas.cu:143-148 (with the threshold_n fix)
  if (!((unsigned long long)g.m >= h->intercept_threshold_m && (unsigned long long)g.n >= h->intercept_threshold_n &&
        (unsigned long long)g.k >= h->intercept_threshold_k))
    return Try::NotTaken;

  const bool prof = culip_enabled();
  timespec t0{}, t1{};
  if (prof) { // src/culip.cu:19-39: stream sync, CLOCK_MONOTONIC
    hipStreamSynchronize(stream);
    clock_gettime(CLOCK_MONOTONIC, &t0);
  }
  int err;
  {
    DepthGuard guard; // auto mode may fall back to the vendor DGEMM underneath
    const ozimmu_element_kind_t kind = g.cplx ? OZIMMU_COMPLX : OZIMMU_REAL;
    if (!g.strided_batched_entry)
      err = ozimmu_hip_gemm_on_stream(h, stream, g.op_a, g.op_b, (size_t)g.m, (size_t)g.n, (size_t)g.k, g.alpha, g.A,
                                      (size_t)g.lda, g.B, (size_t)g.ldb, g.beta, g.C, (size_t)g.ldc, mode, kind);
    else
      err = ozimmu_hip_gemm_strided_batched(h, stream, g.op_a, g.op_b, (size_t)g.m, (size_t)g.n, (size_t)g.k, g.alpha,
                                            g.A, (size_t)g.lda, g.stride_a, g.B, (size_t)g.ldb, g.stride_b, g.beta, g.C,
                                            (size_t)g.ldc, g.stride_c, (size_t)g.batch, mode, kind);
  }
  if (prof) {
    hipStreamSynchronize(stream);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const unsigned long ns =
        ((long)t1.tv_sec - (long)t0.tv_sec) * 1000000000l + ((long)t1.tv_nsec - (long)t0.tv_nsec);
    // name formats of src/cublas.cu:157-162 (gemm) and :342-350 (strided batched: one line per call)
    if (!g.strided_batched_entry)
      std::printf("[CULiP Result][%s%s-%s%s-m%lld-n%lld-k%lld] %luns\n", g.cplx ? "Z" : "D",
                  ozimmu_hip_get_compute_mode_name_str(mode), op_str(g.op_a), op_str(g.op_b), g.m, g.n, g.k, ns);
    else
      std::printf("[CULiP Result][%s%s_stridedBatched-%s%s-m%lld-n%lld-k%lld-batch_count%lld] %luns\n",
                  g.cplx ? "Z" : "D", ozimmu_hip_get_compute_mode_name_str(mode), op_str(g.op_a), op_str(g.op_b), g.m,
                  g.n, g.k, g.batch, ns);
    std::fflush(stdout);
  }
  if (err == 0) return Try::Done;
  if (err == 4) {
    log_error("Ozaki path failed after C had been modified (status 4): reporting an error to the caller");
    return Try::Failed;
  }
  log_error("Ozaki path failed (status " + std::to_string(err) + "); falling back to the vendor GEMM");
  return Try::NotTaken;
}

// ---- vendor-specific glue ------------------------------------------------------------------------------------------
struct RB { // rocBLAS
  typedef rocblas_status status;
  typedef rocblas_handle handle;
  typedef rocblas_operation operation;
  static constexpr status ok = rocblas_status_success, err = rocblas_status_internal_error;
  static constexpr bool guard_forward = true; // the vendor routine may call other rocBLAS entry points
  static bool ctx(handle h, hipStream_t *stream, bool *host_mode) {
    static const auto get_stream = original<decltype(&rocblas_get_stream)>("rocblas_get_stream");
    static const auto get_pm = original<decltype(&rocblas_get_pointer_mode)>("rocblas_get_pointer_mode");
    if (!h || !get_stream || !get_pm) return false;
    rocblas_pointer_mode pm = rocblas_pointer_mode_host;
    if (get_stream(h, stream) != rocblas_status_success) return false;
    if (get_pm(h, &pm) != rocblas_status_success) return false;
    *host_mode = pm == rocblas_pointer_mode_host;
    return true;
  }
  static ozimmu_operation_t to_oz(operation op) { // (src/cublas.cu:50-56 maps every non-N to T: wrong for complex data)
    return op == rocblas_operation_none ? OZIMMU_OP_N : op == rocblas_operation_conjugate_transpose ? OZIMMU_OP_C : OZIMMU_OP_T;
  }
};
struct HB { // hipBLAS: only reached when an application binds hipBLAS statically or resolves these first
  typedef hipblasStatus_t status;
  typedef hipblasHandle_t handle;
  typedef hipblasOperation_t operation;
  static constexpr status ok = HIPBLAS_STATUS_SUCCESS, err = HIPBLAS_STATUS_INTERNAL_ERROR;
  // no DepthGuard on the forward: the vendor hipBLAS routine calls the rocBLAS entry point, the normal intercept point
  static constexpr bool guard_forward = false;
  static bool ctx(handle h, hipStream_t *stream, bool *host_mode) {
    static const auto get_stream = original<decltype(&hipblasGetStream)>("hipblasGetStream");
    static const auto get_pm = original<decltype(&hipblasGetPointerMode)>("hipblasGetPointerMode");
    if (!h || !get_stream || !get_pm) return false;
    hipblasPointerMode_t pm = HIPBLAS_POINTER_MODE_HOST;
    if (get_stream(h, stream) != HIPBLAS_STATUS_SUCCESS) return false;
    if (get_pm(h, &pm) != HIPBLAS_STATUS_SUCCESS) return false;
    *host_mode = pm == HIPBLAS_POINTER_MODE_HOST;
    return true;
  }
  static ozimmu_operation_t to_oz(operation op) {
    return op == HIPBLAS_OP_N ? OZIMMU_OP_N : op == HIPBLAS_OP_C ? OZIMMU_OP_C : OZIMMU_OP_T;
  }
};

// Common body of every interposed GEMM: try the Ozaki path when `eligible` (types / in-place checks of the caller),
// else (or when it declines) forward to the vendor definition.  Conjugate-transposed complex operands take the Ozaki path
// as OZIMMU_OP_C (the reference maps every non-N operation to a plain transpose, src/cublas.cu:50-56: wrong for complex data).
template <class V, class Forward>
typename V::status entry(bool eligible, typename V::handle handle, typename V::operation ta, typename V::operation tb,
                         GemmCall g, bool have_fn, Forward forward) {
  ozimmu_compute_mode_t mode = OZIMMU_DGEMM;
  if (t_depth == 0 && eligible && g.batch > 0 &&
      (mode = get_compute_mode()) != OZIMMU_DGEMM) { // the one per-call read of OZIMMU_COMPUTE_MODE (src/cublas.cu:18-48)
    hipStream_t stream = nullptr;
    bool host_mode = false;
    if (V::ctx(handle, &stream, &host_mode)) {
      g.op_a = V::to_oz(ta);
      g.op_b = V::to_oz(tb);
      const Try t = try_ozaki(stream, host_mode, g, mode);
      g_stat_seen.fetch_add(1, std::memory_order_relaxed);
      (t == Try::Done ? g_stat_taken : t == Try::Failed ? g_stat_failed : g_stat_declined).fetch_add(1, std::memory_order_relaxed);
      if (t == Try::Done) return V::ok;
      if (t == Try::Failed) return V::err;
    }
  }
  if (!have_fn) return V::err;
  if (V::guard_forward) {
    DepthGuard guard;
    return forward();
  }
  return forward();
}

GemmCall call(long long m, long long n, long long k, const void *alpha, const void *A, long long lda, const void *B,
              long long ldb, const void *beta, void *C, long long ldc, bool cplx, long long stride_a = 0,
              long long stride_b = 0, long long stride_c = 0, long long batch = 1, bool sb_entry = false) {
  GemmCall g{};
  g.strided_batched_entry = sb_entry;
  g.m = m; g.n = n; g.k = k;
  g.alpha = alpha; g.A = A; g.lda = lda; g.stride_a = stride_a;
  g.B = B; g.ldb = ldb; g.stride_b = stride_b;
  g.beta = beta; g.C = C; g.ldc = ldc; g.stride_c = stride_c;
  g.batch = batch;
  g.cplx = cplx;
  return g;
}

// lifecycle (src/cublas.cu:104-131), shared by the rocBLAS and the hipBLAS create / destroy hooks
void on_vendor_handle_created() {
  g_live_vendor_handles++;
  // the reference pre-sizes the workspace for a 1024^3 fp64_int8_9 GEMM here (src/cublas.cu:12-16, :109-110)
  const ozimmu_compute_mode_t mode = get_compute_mode();
  if (is_int8_mode(mode) || mode == OZIMMU_FP64_INT8_AUTO)
    if (ozimmu_hip_handle_t h = get_global_handle(nullptr))
      ozimmu_hip_reallocate_working_memory(
          h, ozimmu_hip_working_memory_size(OZIMMU_OP_N, OZIMMU_OP_N, 1024, 1024, 1024, OZIMMU_REAL, OZIMMU_FP64_INT8_9));
}
void on_vendor_handle_destroyed() {
  if (g_live_vendor_handles.fetch_sub(1) != 1) return;
  std::lock_guard<std::mutex> lock(g_mtx);
  if (g_handles.empty()) return;
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

} // namespace

extern "C" {

int ozimmu_hip_intercept_stats(unsigned long long *out, int count) {
  if (!out || count < 0) return 1;
  unsigned long long v[4 + ozhip::PICK_HIST_SLOTS];
  v[0] = g_stat_seen.load(std::memory_order_relaxed);
  v[1] = g_stat_taken.load(std::memory_order_relaxed);
  v[2] = g_stat_declined.load(std::memory_order_relaxed);
  v[3] = g_stat_failed.load(std::memory_order_relaxed);
  ozhip::pick_histogram(v + 4);
  for (int i = 0; i < count; i++) out[i] = i < 4 + ozhip::PICK_HIST_SLOTS ? v[i] : 0ull;
  return 0;
}

// ---- lifecycle (src/cublas.cu:104-131) -----------------------------------------------------------------

rocblas_status rocblas_create_handle(rocblas_handle *handle) {
  OZ_ORIGINAL(rocblas_create_handle);
  if (!fn) return rocblas_status_internal_error;
  const rocblas_status st = fn(handle);
  if (st == rocblas_status_success && t_depth == 0) on_vendor_handle_created();
  return st;
}

rocblas_status rocblas_destroy_handle(rocblas_handle handle) {
  OZ_ORIGINAL(rocblas_destroy_handle);
  if (!fn) return rocblas_status_internal_error;
  if (t_depth == 0) on_vendor_handle_destroyed();
  return fn(handle);
}

// hipblasCreate / hipblasDestroy: a dynamically linked hipBLAS reaches rocblas_create_handle above, a
// hipBLAS that binds rocBLAS statically does not.  The lifecycle runs here once; the nested rocBLAS hook (if the vendor
// routine comes through it) sees the depth guard and stays out.
hipblasStatus_t hipblasCreate(hipblasHandle_t *handle) {
  OZ_ORIGINAL(hipblasCreate);
  if (!fn) return HIPBLAS_STATUS_INTERNAL_ERROR;
  hipblasStatus_t st;
  {
    DepthGuard guard;
    st = fn(handle);
  }
  if (st == HIPBLAS_STATUS_SUCCESS && t_depth == 0) on_vendor_handle_created();
  return st;
}

hipblasStatus_t hipblasDestroy(hipblasHandle_t handle) {
  OZ_ORIGINAL(hipblasDestroy);
  if (!fn) return HIPBLAS_STATUS_INTERNAL_ERROR;
  if (t_depth == 0) on_vendor_handle_destroyed();
  DepthGuard guard;
  return fn(handle);
}

// ---- {d,z}gemm, 32- and 64-bit index twins ---------------------------------------------------------------------------
#define OZ_GEMM(V, NAME, OP, INT, T, CPLX)                                                                              \
  V::status NAME(V::handle handle, OP transA, OP transB, INT m, INT n, INT k, const T *alpha, const T *A, INT lda,      \
                 const T *B, INT ldb, const T *beta, T *C, INT ldc) {                                                   \
    OZ_ORIGINAL(NAME);                                                                                                  \
    return entry<V>(true, handle, transA, transB, call(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, CPLX), fn != nullptr, \
                    [&] { return fn(handle, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc); });          \
  }
OZ_GEMM(RB, rocblas_dgemm, rocblas_operation, rocblas_int, double, false)
OZ_GEMM(RB, rocblas_dgemm_64, rocblas_operation, int64_t, double, false)
OZ_GEMM(RB, rocblas_zgemm, rocblas_operation, rocblas_int, rocblas_double_complex, true)
OZ_GEMM(RB, rocblas_zgemm_64, rocblas_operation, int64_t, rocblas_double_complex, true)
OZ_GEMM(HB, hipblasDgemm, hipblasOperation_t, int, double, false)
OZ_GEMM(HB, hipblasDgemm_64, hipblasOperation_t, int64_t, double, false)
OZ_GEMM(HB, hipblasZgemm, hipblasOperation_t, int, hipDoubleComplex, true)
OZ_GEMM(HB, hipblasZgemm_64, hipblasOperation_t, int64_t, hipDoubleComplex, true)
#undef OZ_GEMM

// ---- {d,z}gemm_strided_batched ------------------------------------------------------------------------------------------
#define OZ_GEMM_SB(V, NAME, OP, INT, STRIDE, T, CPLX)                                                                   \
  V::status NAME(V::handle handle, OP transA, OP transB, INT m, INT n, INT k, const T *alpha, const T *A, INT lda,      \
                 STRIDE stride_a, const T *B, INT ldb, STRIDE stride_b, const T *beta, T *C, INT ldc, STRIDE stride_c,  \
                 INT batch_count) {                                                                                     \
    OZ_ORIGINAL(NAME);                                                                                                  \
    return entry<V>(true, handle, transA, transB,                                                                       \
                    call(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, CPLX, stride_a, stride_b, stride_c, batch_count, true), \
                    fn != nullptr, [&] {                                                                                \
                      return fn(handle, transA, transB, m, n, k, alpha, A, lda, stride_a, B, ldb, stride_b, beta, C,    \
                                ldc, stride_c, batch_count);                                                            \
                    });                                                                                                 \
  }
OZ_GEMM_SB(RB, rocblas_dgemm_strided_batched, rocblas_operation, rocblas_int, rocblas_stride, double, false)
OZ_GEMM_SB(RB, rocblas_dgemm_strided_batched_64, rocblas_operation, int64_t, rocblas_stride, double, false)
OZ_GEMM_SB(RB, rocblas_zgemm_strided_batched, rocblas_operation, rocblas_int, rocblas_stride, rocblas_double_complex, true)
OZ_GEMM_SB(RB, rocblas_zgemm_strided_batched_64, rocblas_operation, int64_t, rocblas_stride, rocblas_double_complex, true)
OZ_GEMM_SB(HB, hipblasDgemmStridedBatched, hipblasOperation_t, int, long long, double, false)
OZ_GEMM_SB(HB, hipblasDgemmStridedBatched_64, hipblasOperation_t, int64_t, long long, double, false)
OZ_GEMM_SB(HB, hipblasZgemmStridedBatched, hipblasOperation_t, int, long long, hipDoubleComplex, true)
OZ_GEMM_SB(HB, hipblasZgemmStridedBatched_64, hipblasOperation_t, int64_t, long long, hipDoubleComplex, true)
#undef OZ_GEMM_SB

// ---- rocblas_gemm_ex / rocblas_gemm_strided_batched_ex: all-FP64 real or all-FP64 complex, in place (C == D) ------------
// (src/cublas.cu:146-148 accepts CUDA_R_64F / CUDA_C_64F operands only; cuBLAS has no separate D)
static bool rb_all(rocblas_datatype t, rocblas_datatype a, rocblas_datatype b, rocblas_datatype c, rocblas_datatype d,
                   rocblas_datatype compute) {
  return a == t && b == t && c == t && d == t && compute == t;
}
#define OZ_RB_GEMM_EX(NAME, INT)                                                                                        \
  rocblas_status NAME(rocblas_handle handle, rocblas_operation transA, rocblas_operation transB, INT m, INT n, INT k,   \
                      const void *alpha, const void *a, rocblas_datatype a_type, INT lda, const void *b,                \
                      rocblas_datatype b_type, INT ldb, const void *beta, const void *c, rocblas_datatype c_type,       \
                      INT ldc, void *d, rocblas_datatype d_type, INT ldd, rocblas_datatype compute_type,                \
                      rocblas_gemm_algo algo, int32_t solution_index, uint32_t flags) {                                 \
    OZ_ORIGINAL(NAME);                                                                                                  \
    const bool f64 = rb_all(rocblas_datatype_f64_r, a_type, b_type, c_type, d_type, compute_type);                      \
    const bool c64 = rb_all(rocblas_datatype_f64_c, a_type, b_type, c_type, d_type, compute_type);                      \
    return entry<RB>((f64 || c64) && c == d && ldc == ldd, handle, transA, transB,                                      \
                     call(m, n, k, alpha, a, lda, b, ldb, beta, d, ldd, c64), fn != nullptr, [&] {                      \
                       return fn(handle, transA, transB, m, n, k, alpha, a, a_type, lda, b, b_type, ldb, beta, c,       \
                                 c_type, ldc, d, d_type, ldd, compute_type, algo, solution_index, flags);               \
                     });                                                                                                \
  }
OZ_RB_GEMM_EX(rocblas_gemm_ex, rocblas_int)
OZ_RB_GEMM_EX(rocblas_gemm_ex_64, int64_t)
#undef OZ_RB_GEMM_EX

// The 32-bit name is parenthesised because rocblas.h also defines a backward-compatibility macro of the same name.
#define OZ_RB_GEMM_SB_EX(DECL, NAME, INT)                                                                               \
  rocblas_status DECL(rocblas_handle handle, rocblas_operation transA, rocblas_operation transB, INT m, INT n, INT k,   \
                      const void *alpha, const void *a, rocblas_datatype a_type, INT lda, rocblas_stride stride_a,      \
                      const void *b, rocblas_datatype b_type, INT ldb, rocblas_stride stride_b, const void *beta,       \
                      const void *c, rocblas_datatype c_type, INT ldc, rocblas_stride stride_c, void *d,                \
                      rocblas_datatype d_type, INT ldd, rocblas_stride stride_d, INT batch_count,                       \
                      rocblas_datatype compute_type, rocblas_gemm_algo algo, int32_t solution_index, uint32_t flags) {  \
    static const auto fn = original<decltype(&NAME)>(#NAME);                                                           \
    const bool f64 = rb_all(rocblas_datatype_f64_r, a_type, b_type, c_type, d_type, compute_type);                      \
    const bool c64 = rb_all(rocblas_datatype_f64_c, a_type, b_type, c_type, d_type, compute_type);                      \
    return entry<RB>((f64 || c64) && c == d && ldc == ldd && stride_c == stride_d, handle, transA, transB,              \
                     call(m, n, k, alpha, a, lda, b, ldb, beta, d, ldd, c64, stride_a, stride_b, stride_d, batch_count, true), \
                     fn != nullptr, [&] {                                                                               \
                       return fn(handle, transA, transB, m, n, k, alpha, a, a_type, lda, stride_a, b, b_type, ldb,      \
                                 stride_b, beta, c, c_type, ldc, stride_c, d, d_type, ldd, stride_d, batch_count,       \
                                 compute_type, algo, solution_index, flags);                                            \
                     });                                                                                                \
  }
OZ_RB_GEMM_SB_EX((rocblas_gemm_strided_batched_ex), rocblas_gemm_strided_batched_ex, rocblas_int)
OZ_RB_GEMM_SB_EX(rocblas_gemm_strided_batched_ex_64, rocblas_gemm_strided_batched_ex_64, int64_t)
#undef OZ_RB_GEMM_SB_EX

// ---- hipblasGemmEx[WithFlags][_64] / hipblasGemmStridedBatchedEx[WithFlags][_64] -------------------------------------------
static bool hb_all(hipDataType t, hipDataType a, hipDataType b, hipDataType c, hipblasComputeType_t compute) {
  return a == t && b == t && c == t && compute == HIPBLAS_COMPUTE_64F;
}
#define OZ_HB_GEMM_EX(NAME, INT, FLAGS_PARAM, FLAGS_ARG)                                                                \
  hipblasStatus_t NAME(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, INT m, INT n,      \
                       INT k, const void *alpha, const void *A, hipDataType aType, INT lda, const void *B,              \
                       hipDataType bType, INT ldb, const void *beta, void *C, hipDataType cType, INT ldc,               \
                       hipblasComputeType_t computeType, hipblasGemmAlgo_t algo FLAGS_PARAM) {                          \
    OZ_ORIGINAL(NAME);                                                                                                  \
    const bool f64 = hb_all(HIP_R_64F, aType, bType, cType, computeType);                                               \
    const bool c64 = hb_all(HIP_C_64F, aType, bType, cType, computeType);                                               \
    return entry<HB>(f64 || c64, handle, transA, transB, call(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, c64),       \
                     fn != nullptr, [&] {                                                                               \
                       return fn(handle, transA, transB, m, n, k, alpha, A, aType, lda, B, bType, ldb, beta, C, cType,  \
                                 ldc, computeType, algo FLAGS_ARG);                                                     \
                     });                                                                                                \
  }
#define OZ_COMMA ,
OZ_HB_GEMM_EX(hipblasGemmEx, int, , )
OZ_HB_GEMM_EX(hipblasGemmEx_64, int64_t, , )
OZ_HB_GEMM_EX(hipblasGemmExWithFlags, int, OZ_COMMA hipblasGemmFlags_t flags, OZ_COMMA flags)
OZ_HB_GEMM_EX(hipblasGemmExWithFlags_64, int64_t, OZ_COMMA hipblasGemmFlags_t flags, OZ_COMMA flags)
#undef OZ_HB_GEMM_EX

#define OZ_HB_GEMM_SB_EX(NAME, INT, FLAGS_PARAM, FLAGS_ARG)                                                             \
  hipblasStatus_t NAME(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, INT m, INT n,      \
                       INT k, const void *alpha, const void *A, hipDataType aType, INT lda, hipblasStride strideA,      \
                       const void *B, hipDataType bType, INT ldb, hipblasStride strideB, const void *beta, void *C,     \
                       hipDataType cType, INT ldc, hipblasStride strideC, INT batchCount,                               \
                       hipblasComputeType_t computeType, hipblasGemmAlgo_t algo FLAGS_PARAM) {                          \
    OZ_ORIGINAL(NAME);                                                                                                  \
    const bool f64 = hb_all(HIP_R_64F, aType, bType, cType, computeType);                                               \
    const bool c64 = hb_all(HIP_C_64F, aType, bType, cType, computeType);                                               \
    return entry<HB>(f64 || c64, handle, transA, transB,                                                                \
                     call(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, c64, strideA, strideB, strideC, batchCount, true),    \
                     fn != nullptr, [&] {                                                                               \
                       return fn(handle, transA, transB, m, n, k, alpha, A, aType, lda, strideA, B, bType, ldb,         \
                                 strideB, beta, C, cType, ldc, strideC, batchCount, computeType, algo FLAGS_ARG);       \
                     });                                                                                                \
  }
OZ_HB_GEMM_SB_EX(hipblasGemmStridedBatchedEx, int, , )
OZ_HB_GEMM_SB_EX(hipblasGemmStridedBatchedEx_64, int64_t, , )
OZ_HB_GEMM_SB_EX(hipblasGemmStridedBatchedExWithFlags, int, OZ_COMMA hipblasGemmFlags_t flags, OZ_COMMA flags)
OZ_HB_GEMM_SB_EX(hipblasGemmStridedBatchedExWithFlags_64, int64_t, OZ_COMMA hipblasGemmFlags_t flags, OZ_COMMA flags)
#undef OZ_HB_GEMM_SB_EX
#undef OZ_COMMA

} // extern "C"
