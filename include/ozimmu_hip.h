/*
 * ozimmu_hip.h — C ABI of libozimmu_hip.so: Ozaki-scheme (INT8 MFMA) DGEMM for MI355X (gfx950).
 *
 * Two boundaries, both drop-ins for enp1s0/ozIMMU's (citations relative to /root/reference):
 *
 * 1. LD_PRELOAD interposer (replaces src/cublas.cu:103-513, the `extern "C"` cuBLAS symbols).
 *    The library DEFINES the following rocBLAS / hipBLAS entry points with their vendor prototypes
 *    (<rocblas/rocblas.h>, <hipblas/hipblas.h>); anything it does not take over is forwarded to the
 *    next definition found with dlsym(RTLD_NEXT, ...) exactly like src/utils.hpp:117-141:
 *
 *      rocblas_create_handle, rocblas_destroy_handle       <- cublasCreate_v2 / cublasDestroy_v2 (src/cublas.cu:104-131)
 *      rocblas_dgemm[_64]                                  <- cublasDgemm_v2 (src/cublas.cu:280-295)
 *      rocblas_gemm_ex[_64] (all-f64_r / all-f64_c case)   <- cublasGemmEx   (src/cublas.cu:133-278)
 *      rocblas_dgemm_strided_batched[_64]                  <- cublasDgemmStridedBatched (src/cublas.cu:474-492)
 *      rocblas_zgemm[_64]                                  <- cublasZgemm_v2 (src/cublas.cu:297-313)
 *      rocblas_zgemm_strided_batched[_64]                  <- cublasZgemmStridedBatched (src/cublas.cu:494-512)
 *      rocblas_gemm_strided_batched_ex[_64]                <- cublasGemmStridedBatchedEx (src/cublas.cu:315-472)
 *      hipblas{D,Z}gemm[_64], hipblasGemmEx[WithFlags][_64],   <- same, for applications that bind hipBLAS directly
 *      hipblas{D,Z}gemmStridedBatched[_64],                       (hipBLAS itself calls the rocBLAS entry points
 *      hipblasGemmStridedBatchedEx[WithFlags][_64]                 above, including the ILP64 `_64` ones)
 *
 *    Environment (src/cublas.cu:18-48, :62-83; src/handle.cu:25-30; src/utils.hpp:88-115; README.md:54-77):
 *      OZIMMU_COMPUTE_MODE = dgemm | sgemm | fp64_int8_3..fp64_int8_18 | fp64_int8_auto  (read per call;
 *                            unset/unknown -> dgemm = pass through; sgemm -> FP32 vendor GEMM on converted copies,
 *                            ozimmu_hip_gemm_f32)
 *      OZIMMU_INTERCEPT_THRESHOLD_M / _N / _K (default 1024), OZIMMU_AUTO_AVG_MANTISSA_LOSS_THRESHOLD,
 *      OZIMMU_INFO, OZIMMU_ERROR, OZIMMU_MALLOC_ASYNC, OZIMMU_ENABLE_CULIP_PROFILING.
 *
 * 2. Direct library API (replaces include/ozimmu/ozimmu.hpp:47-100, the `mtk::ozimmu::` C++ API that the
 *    reference's own test harness calls, test/main_test.cu:242).  Same names, argument order and return
 *    conventions, flattened to C: no exceptions cross this boundary.
 *
 * All pointers to matrices are DEVICE pointers (column-major); alpha/beta are HOST pointers
 * (src/gemm.cu:405 dereferences them on the host).  Work is enqueued on the handle's stream; nothing here
 * synchronises the device.
 */
#ifndef OZIMMU_HIP_H
#define OZIMMU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ozimmu_hip_handle *ozimmu_hip_handle_t; /* include/ozimmu/ozimmu.hpp:9-11 */

/* ozimmu.hpp:12 has op_n, op_t only: the reference's cuBLAS hook maps CUBLAS_OP_C to op_t (src/cublas.cu:50-56), so its
 * ZGEMM with a conjugate-transposed operand computes A^T B instead of A^H B.  OZIMMU_OP_C is the conjugate transpose, computed
 * correctly (complex element kind: the imaginary part of that operand enters with the opposite sign; real: the same as
 * OZIMMU_OP_T). */
typedef enum { OZIMMU_OP_N = 0, OZIMMU_OP_T = 1, OZIMMU_OP_C = 2 } ozimmu_operation_t;

/* include/ozimmu/ozimmu.hpp:14-37 — same order, so the integer values match the reference enum */
typedef enum {
  OZIMMU_SGEMM = 0,
  OZIMMU_DGEMM = 1,
  OZIMMU_FP64_INT8_3 = 2,
  OZIMMU_FP64_INT8_4,
  OZIMMU_FP64_INT8_5,
  OZIMMU_FP64_INT8_6,
  OZIMMU_FP64_INT8_7,
  OZIMMU_FP64_INT8_8,
  OZIMMU_FP64_INT8_9,
  OZIMMU_FP64_INT8_10,
  OZIMMU_FP64_INT8_11,
  OZIMMU_FP64_INT8_12,
  OZIMMU_FP64_INT8_13,
  OZIMMU_FP64_INT8_14,
  OZIMMU_FP64_INT8_15,
  OZIMMU_FP64_INT8_16,
  OZIMMU_FP64_INT8_17,
  OZIMMU_FP64_INT8_18,
  OZIMMU_FP64_INT8_AUTO
} ozimmu_compute_mode_t;

typedef enum { OZIMMU_MALLOC_SYNC = 0, OZIMMU_MALLOC_ASYNC = 1 } ozimmu_malloc_mode_t; /* ozimmu.hpp:41 */
/* ozimmu.hpp:38 data_t, same order */
typedef enum {
  OZIMMU_DATA_FP64 = 0,
  OZIMMU_DATA_FP32,
  OZIMMU_DATA_FP16,
  OZIMMU_DATA_INT8,
  OZIMMU_DATA_ORIGINAL,
  OZIMMU_DATA_NONE
} ozimmu_data_t;
typedef enum { OZIMMU_REAL = 0, OZIMMU_COMPLX = 1 } ozimmu_element_kind_t;            /* ozimmu.hpp:43-46 */
typedef enum { OZIMMU_MATRIX_A = 0, OZIMMU_MATRIX_B = 1 } ozimmu_matrix_t;            /* src/config.hpp:9 */

/* ozimmu.hpp:48-49 (src/handle.cu:6-52).  Return 0 on success. */
int ozimmu_hip_create(ozimmu_hip_handle_t *handle, ozimmu_malloc_mode_t mm);
/* What the launch policy knows about the handle's device (probed once per device at the first create): out[0] = CUs,
 * out[1] = XCDs (L2 domains), out[2] = the time in us of one v_mfma_i32_32x32x32_i8 under sustained full-entropy load that the
 * policy plans with (the measured value clamped to +-25 % of the nominal 0.0194), out[3] = the measured value itself.  No
 * counterpart in the reference (it has no kernel choice: src/gemm.cu:315-329 calls cuBLAS). */
int ozimmu_hip_device_info(ozimmu_hip_handle_t handle, double out[4]);
/* Diagnostics of the kernel choice (csrc/kernel_policy.h; the reference has none: src/gemm.cu:315-329 calls cuBLAS per pair).
 * ozimmu_hip_policy_predict: the cost model's predicted GEMM-stage time in us of every kernel for pass `pass` (0: the single
 *   / first diagonal pass, 1: the second pass of fp64_int8_13..18) of an m x n x k product (its first K chunk) at `num_split`
 *   slices: out_us[0..5] = K-split, classic, wide 32x32x32, wide paired 16x16x64, wide k64 (B through LDS), wide k64 (B in
 *   registers); < 0 = not eligible.  *pick = index of the kernel the policy takes (k64 in registers: 4 + 8).  handle may be
 *   NULL: the prediction is then made for the nominal device (256 CUs) - host arithmetic only, no GPU needed.
 * ozimmu_hip_policy_params: read (set = 0) / replace (set = 1) the first `count` fitted constants of the model.
 * ozimmu_hip_last_kernel: out[0], out[1] = the pick (as above) of the last slice-GEMM launch of this handle for its first /
 *   second pass; -1 = none. */
int ozimmu_hip_policy_predict(ozimmu_hip_handle_t handle, int num_split, int pass, size_t m, size_t n, size_t k, size_t batch,
                              double out_us[6], int *pick);
int ozimmu_hip_policy_params(double *params, int count, int set);
/* The launch policy is bound to the DEVICE ID a handle was created on, never to "the current device" of the calling thread
 * (the reference keeps one global handle for one device: src/cublas.cu:58; a process that drives several GPUs through this
 * library gets one handle - workspace, topology, kernel attributes - per device).
 * ozimmu_hip_device_topology: set = 0: inout = {CUs, XCDs, planned MFMA time, measured MFMA time} of slot `device` (what
 *   ozimmu_hip_device_info returns for a handle of that device; nominal values until a handle was created there); set = 1:
 *   replace the slot - test flavour only (fake devices on a box without a GPU), the library that ships returns 2.
 * ozimmu_hip_policy_predict_device: ozimmu_hip_policy_predict with the topology of slot `device`. */
int ozimmu_hip_device_topology(int device, double inout[4], int set);
int ozimmu_hip_policy_predict_device(int device, int num_split, int pass, size_t m, size_t n, size_t k, size_t batch,
                                     double out_us[6], int *pick);
int ozimmu_hip_last_kernel(ozimmu_hip_handle_t handle, int out[2]);
/* Measured kernel choice (csrc/kernel_tuner.h; no counterpart in the reference, whose GEMMs cuBLAS plans): a plain real GEMM
 * whose (mode, op_A, op_B, m, n, k, beta != 0) the handle is called with for the 8th time (OZIMMU_HIP_AUTOTUNE_AFTER) starts a
 * measurement: the kernels the model predicts within 25 % (K <= 512; 12 % up to K = 2048) of its best - all of them return the
 * same bits - run in turn, each call bracketed by two events on the caller's stream, and the fastest is kept once four rounds
 * of times (the first one discarded) are in (no synchronisation: finished event pairs are collected by later calls); the shape
 * is measured again 64, 512, 4096 ... calls later.  Calls 1 .. 7 of a shape run what the model picks, untimed.  Never on a
 * handle that has been captured into a graph.  OZIMMU_HIP_AUTOTUNE=0 switches it off.
 * ozimmu_hip_tuner_state: -1 = shape unknown to this handle, 0 = counting calls or measuring, 1 = decided;
 * out[0] = prediction slot (0..5, as in ozimmu_hip_policy_predict) of the decided kernel or -1, out[1] = candidates; the most
 * recently used entry of any layout / beta class.  _ex: exactly that class (op_A < 0: any), out[2] = calls seen, out[3] =
 * measurements finished. */
int ozimmu_hip_tuner_state(ozimmu_hip_handle_t handle, int num_split, size_t m, size_t n, size_t k, int out[2]);
int ozimmu_hip_tuner_state_ex(ozimmu_hip_handle_t handle, int num_split, int op_A, int op_B, int beta_nonzero, size_t m, size_t n,
                              size_t k, int out[4]);
/* Measurement aid (bench.py: roofline.frac_of_measured_mfma_ceiling): the INT8 rate the matrix pipe of `device` (< 0: the
 * current one) sustains on v_mfma_i32_16x16x64_i8 alone - random operands in registers, no memory traffic - over the last
 * 60 % of `seconds` (<= 30) of back-to-back launches, in TOPS (2 ops per MAC).  Synchronises the device.  0 = ok. */
int ozimmu_hip_mfma_ceiling(int device, double seconds, double *tops);
/* Process-wide diagnostics of the LD_PRELOAD boundary (csrc/interpose.cpp; the reference only has its log lines,
 * src/cublas.cu:153-166): out[0] = FP64 GEMM calls that reached an interposed entry point with a compute mode other than
 * `dgemm`, out[1] = ... that ran on the Ozaki path, out[2] = ... left to the vendor routine (thresholds, pointer mode,
 * status 3), out[3] = ... failed after C was modified (status 4); out[4 + c] = slice-GEMM launches (first diagonal pass)
 * of kernel code c (the codes of ozimmu_hip_last_kernel: 0 k2, 1 classic, 2 wide, 3 x16, 4 k64, 12 k64 in registers,
 * 16 k2 one launch), direct API calls included.  `count` entries are written (28 are defined). */
int ozimmu_hip_intercept_stats(unsigned long long *out, int count);
int ozimmu_hip_destroy(ozimmu_hip_handle_t handle);

/* ozimmu.hpp:50-51 set_cuda_stream (src/handle.cu:54-61). `hip_stream` is a hipStream_t. */
void ozimmu_hip_set_stream(ozimmu_hip_handle_t handle, void *hip_stream);

/* ozimmu.hpp:53-57 (src/handle.cu:246-265): stage breakdown split_A/split_B/int8tc/accumulate_in_f64/copy_result */
void ozimmu_hip_enable_profiling(ozimmu_hip_handle_t handle);
void ozimmu_hip_disable_profiling(ozimmu_hip_handle_t handle);
void ozimmu_hip_print_profiler_result(ozimmu_hip_handle_t handle, const char *tag, int csv);
void ozimmu_hip_clear_profiler_result(ozimmu_hip_handle_t handle);

/* ozimmu.hpp:59-61 (the reference never defines the getter in its namespace: src/handle.cu:272-274) */
void ozimmu_hip_set_auto_mantissa_loss_threashold(ozimmu_hip_handle_t handle, double threshold);
double ozimmu_hip_get_auto_mantissa_loss_threashold(ozimmu_hip_handle_t handle);

/* ozimmu.hpp:69-74.  Returns the new size if the workspace grew, else 0 (src/handle.cu:63-93). */
size_t ozimmu_hip_reallocate_working_memory(ozimmu_hip_handle_t handle, size_t size_in_byte);
/* size one GEMM needs (the gemm_list_t overload, src/handle.cu:95-144, one entry at a time) */
size_t ozimmu_hip_working_memory_size(ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m, size_t n,
                                      size_t k, ozimmu_element_kind_t element_kind,
                                      ozimmu_compute_mode_t compute_mode);

/* ozimmu.hpp:75-82 (src/gemm.cu:524-653).  0 ok; 1 invalid shape / alignment (src/gemm.cu:554-556);
 * 2 unsupported mode value; 3 device / vendor failure with C untouched (the reference would throw); 4 failure after
 * C had been modified (complex path: beta scaling and the four products update C in place) -- C is undefined and
 * must not be handed to a fallback GEMM.  element_kind OZIMMU_COMPLX: a, b, c are interleaved double-complex,
 * alpha/beta point to {re, im} (gemm_int8<cuDoubleComplex>, src/gemm.cu:412-521).
 * BLAS quick returns and out-of-range sizes run on the vendor GEMM: k == 0, alpha == 0 (C = beta*C, A and B are not
 * read), k > 2^30 (no exact slice width, src/split.cu:520-536), m or n >= 2^31. */
int ozimmu_hip_gemm(ozimmu_hip_handle_t handle, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m,
                    size_t n, size_t k, const void *alpha, const void *a_ptr, size_t lda, const void *b_ptr,
                    size_t ldb, const void *beta, void *c_ptr, size_t ldc,
                    ozimmu_compute_mode_t compute_mode, ozimmu_element_kind_t element_kind);

/* The same with the stream passed along: the handle's stream is switched and the GEMM enqueued under one lock, which
 * is what the interposer needs when several host threads drive one device on different streams (the reference keeps
 * one unguarded global handle, src/cublas.cu:58, :149-151). */
int ozimmu_hip_gemm_on_stream(ozimmu_hip_handle_t handle, void *hip_stream, ozimmu_operation_t op_A,
                              ozimmu_operation_t op_B, size_t m, size_t n, size_t k, const void *alpha,
                              const void *a_ptr, size_t lda, const void *b_ptr, size_t ldb, const void *beta,
                              void *c_ptr, size_t ldc, ozimmu_compute_mode_t compute_mode,
                              ozimmu_element_kind_t element_kind);

/* cublas{D,Z}gemmStridedBatched / cublasGemmStridedBatchedEx (src/cublas.cu:315-512): matrix i of the batch starts
 * stride_x * i elements after the first (strides may be 0 for A and B).  The reference loops over the batch
 * (src/cublas.cu:380-406); here the whole batch goes through one set of launches (batch index = a grid dimension,
 * batch-strided workspace), per-matrix results identical to ozimmu_hip_gemm.  fp64_int8_auto decides per matrix and
 * therefore runs the sequential form.  Status as ozimmu_hip_gemm; 4 also when an earlier part of the batch was
 * already updated. */
int ozimmu_hip_gemm_strided_batched(ozimmu_hip_handle_t handle, void *hip_stream, ozimmu_operation_t op_A,
                                    ozimmu_operation_t op_B, size_t m, size_t n, size_t k, const void *alpha,
                                    const void *a_ptr, size_t lda, long long stride_a, const void *b_ptr, size_t ldb,
                                    long long stride_b, const void *beta, void *c_ptr, size_t ldc, long long stride_c,
                                    size_t batch_count, ozimmu_compute_mode_t compute_mode,
                                    ozimmu_element_kind_t element_kind);

/* ozimmu.hpp:84-94 (src/split.cu:454-518).  Blocks on one 128-byte D2H copy like the reference (:404-408).
 * Returns a fp64_int8_N mode or OZIMMU_DGEMM when no N in 3..18 meets the threshold. */
ozimmu_compute_mode_t ozimmu_hip_auto_mode_select(ozimmu_hip_handle_t handle, ozimmu_operation_t op_A,
                                                  ozimmu_operation_t op_B, size_t m, size_t n, size_t k,
                                                  const void *a_ptr, size_t lda, const void *b_ptr,
                                                  size_t ldb, ozimmu_element_kind_t element_kind,
                                                  double mantissa_loss_threshold);

/* ozimmu.hpp:96 (src/handle.cu:146-192): "fp64_int8_9", ...; NULL for an invalid enum value */
const char *ozimmu_hip_get_compute_mode_name_str(ozimmu_compute_mode_t mode);
/* inverse, as used by src/cublas.cu:18-48: unknown / NULL -> OZIMMU_DGEMM */
ozimmu_compute_mode_t ozimmu_hip_compute_mode_from_str(const char *name);
/* ozimmu.hpp:96 (src/handle.cu:195-226): element type the mode computes its result in: fp32 for `sgemm`, fp64 for `dgemm`
 * and every fp64_int8_* mode; OZIMMU_DATA_ORIGINAL for an invalid enum value (where the reference aborts) */
ozimmu_data_t ozimmu_hip_get_output_type(ozimmu_compute_mode_t mode);
/* ozimmu.hpp:98 (src/handle.cu:228-245): 8 / 4 / 2 / 1 bytes for fp64 / fp32 / fp16 / int8, 0 for original, none and
 * invalid values */
size_t ozimmu_hip_get_data_size_in_byte(ozimmu_data_t d);
/* ozimmu.hpp:100 (src/split.cu:520-536) */
uint32_t ozimmu_hip_get_bits_per_int8(uint32_t k);
/* number of slices of a mode: 3..18, 0 otherwise (src/config.cu:28-80) */
int ozimmu_hip_get_num_split(ozimmu_compute_mode_t mode);

/* ---- stage-level entry points (the reference's internal stage functions, exposed for parity tests) ---- */

/* src/split.hpp:12-19 split_int8<double>: slices of A (matrix_A: m x k view of op(A)) or of B (matrix_B:
 * the k x n op(B), rows = n), written in the REFERENCE layout out[s][row][ldo] (ldo >= padded k) plus
 * max_exp[row].  All device pointers.  Internally runs the production kernels (tiled planes) and then a
 * re-layout kernel. */
int ozimmu_hip_split_int8(ozimmu_hip_handle_t handle, int8_t *out_ptr, uint32_t ldo, double *max_exp_ptr,
                          size_t m, size_t n, const double *in_ptr, size_t ld, ozimmu_operation_t op,
                          ozimmu_matrix_t matrix, unsigned num_split, unsigned bits_per_int8);

/* Diagnostic: the row partition the wide slice GEMM's launch planner picks for an m x n output with tiles of `wa` (and
 * wa-1) 32-row blocks on `cus` compute units: out[0] = rows of full-height tiles, out[1] = rows of reduced tiles,
 * out[2] = makespan in 32-row-block units.  reference = 1 evaluates the same search on a literal simulation of the
 * dispatch (test builds only: returns 2 otherwise); both must agree (tests/test_abi.py).  Host only, no device work;
 * no counterpart in the reference (cuBLAS plans its own tiles). */
int ozimmu_hip_tile_plan(size_t m, size_t n, int wa, int cus, int reference, double *out);

/* INT32 sums per diagonal t = i+j (t = 2..S+1) of the slice products, out[t-2][n][m] (device, int32):
 * what the fused kernel holds in its accumulators before the FP64 recombination; equals the sum over the
 * reference's per-pair cublasGemmEx results (src/gemm.cu:315-329) with i+j = t. */
int ozimmu_hip_diagonal_sums(ozimmu_hip_handle_t handle, ozimmu_operation_t op_A, ozimmu_operation_t op_B,
                             size_t m, size_t n, size_t k, const double *a_ptr, size_t lda,
                             const double *b_ptr, size_t ldb, unsigned num_split, int32_t *out);

/* src/split.hpp:21-27 get_mantissa_loss_total for both operands: counters[S-3], S = 3..18 (host array of 16) */
int ozimmu_hip_mantissa_loss(ozimmu_hip_handle_t handle, ozimmu_operation_t op_A, ozimmu_operation_t op_B,
                             size_t m, size_t n, size_t k, const double *a_ptr, size_t lda,
                             const double *b_ptr, size_t ldb, uint64_t counters[16]);

/* number of getenv() lookups this library has made so far (every lookup goes through one counted wrapper): an intercepted
 * call makes at most three (OZIMMU_COMPUTE_MODE, the auto threshold, the CULiP switch: src/cublas.cu:18-48, :72-83,
 * src/culip.cu:41-50); the OZIMMU_HIP_* development switches are read once per process (csrc/config.h) */
unsigned long long ozimmu_hip_getenv_calls(void);

/* native FP64 GEMM through the real rocBLAS (the `dgemm` mode of src/gemm.cu:639-645); also bench.py's
 * rocBLAS comparison point.  0 ok, nonzero rocblas_status otherwise. */
int ozimmu_hip_native_dgemm(ozimmu_hip_handle_t handle, ozimmu_operation_t op_A, ozimmu_operation_t op_B,
                            size_t m, size_t n, size_t k, const double *alpha, const double *a_ptr,
                            size_t lda, const double *b_ptr, size_t ldb, const double *beta, double *c_ptr,
                            size_t ldc);

/* the `sgemm` compute mode: C = f64( alpha32 * op(f32(A)) * op(f32(B)) + beta32 * f32(C) ) through the vendor
 * SGEMM (real) / CGEMM (complex) -- mtk::ozimmu::dgemm_f32<T>, src/cublas_helper.cu:83-133, reached from the
 * interposer at src/cublas.cu:169-186.  alpha/beta: host pointers to 1 (real) or 2 (complex) doubles.  C is not
 * read when beta == 0.  The FP32 copies live in the handle's workspace ((mk + kn + mn) * 4 bytes, x2 complex).
 * ozimmu_hip_gemm(..., OZIMMU_SGEMM, ...) forwards here (the reference's library entry throws for that mode).
 * 0 ok, 1 bad arguments, 3 device/vendor failure. */
int ozimmu_hip_gemm_f32(ozimmu_hip_handle_t handle, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m,
                        size_t n, size_t k, const void *alpha, const void *a_ptr, size_t lda, const void *b_ptr,
                        size_t ldb, const void *beta, void *c_ptr, size_t ldc, ozimmu_element_kind_t element_kind);

/* timing of the last ozimmu_hip_gemm per stage in milliseconds (HIP events on the handle's stream; only
 * when profiling is enabled).  stages: 0 split_A, 1 split_B, 2 int8tc (fused with accumulate/copy) */
int ozimmu_hip_last_stage_ms(ozimmu_hip_handle_t handle, float ms[3]);

const char *ozimmu_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif
