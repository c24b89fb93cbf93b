// kernels.h — launch interface between the host pipeline (api.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace ozhip {

// per-XCD advisory lines in the workspace (phase hints, claim counters of the persistent wide kernel): one 256-byte line per
// XCD, sized for the largest XCD count the kernels accept (topology.h; MI355X: 8, a CPX partition: 1)
constexpr int MAX_XCDS = 16;
constexpr int PHASE_LINE_WORDS = 64;
constexpr int PHASE_LINES_BYTES = MAX_XCDS * PHASE_LINE_WORDS * 4;

constexpr int SINGLE_PASS_MAX_S = 12; // register budget: 16*S accumulators + 4*S B-fragments + 2 A-fragments <= 256 VGPRs

struct SliceGemmArgs {
  const int8_t *a_planes; // tiled planes of op(A): rows = M (layout.h)
  const int8_t *b_planes; // tiled planes of op(B): rows = N
  uint32_t KB;            // k-blocks per row-block in the planes
  uint32_t kb0, kb1;      // k-block range of this pass (INT32-overflow chunking)
  uint32_t M, N;
  uint32_t tiles_m, tiles_n; // output tiles of the launch (filled in by launch_slice_gemm)
  uint32_t tiles_m2;         // wide kernel: rows of tiles one block lower, below the tiles_m full-height rows
  int L;                     // bits per slice (get_bits_per_int8)
  const double *ea;          // [M] 2^(e_max+1) per row of op(A)
  const double *eb;          // [N] per column of op(B)
  double alpha, beta;
  double *c;
  size_t ldc;
  int cplx;        // 1: c is complex (interleaved), this launch adds (alpha + i*alpha_im) * product to it
  double alpha_im;
  double *acc; // [N][M] FP64 partial sums (multi-pass only)
  int acc_in;  // start the fma chain from acc instead of 0
  int final;   // 1: scale + alpha/beta -> C; 0: -> acc
  uint32_t nxcd;   // XCDs of the device (topology.h; filled in by launch_slice_gemm): runs of tiles, phase lines, claim counters
  uint32_t *phase; // nxcd advisory words (one per XCD, 256 bytes apart): k-block the XCD's workgroups are at; zeroed per call
  // wide kernel, persistent form: per-XCD claim counters {big tiles, small tiles} at queue[PHASE_LINE_WORDS * xcd + {0, 1}], zeroed per
  // call (they live in the phase lines: words 16 + 2 * qslot); nullptr: one tile per workgroup
  uint32_t *queue;
  uint32_t qslot; // set by the host pipeline: counter pair for this launch (a call may need several launches)
  uint32_t phase_min_kb; // passes of at most this many k-blocks run without the phase hint (filled in by launch_slice_gemm)
  uint32_t spec_claim_kb; // persistent k64 kernels: passes of at most this many k-blocks draw the next tile's ticket one tile ahead
                          // (slice_gemm_w_kernel.h; filled in by launch_slice_gemm; 0: never)
  uint32_t epi_overlap; // k64 register kernels: recombine the finished row blocks in the shadow of the last step's MFMAs (filled in by launch_slice_gemm)
  uint32_t throttle; // 0: off; else a workgroup that runs ahead of `phase` sleeps (probed every 16th k-step)
  // test hook, compiled only with -DOZIMMU_HIP_TEST_HOOKS (libozimmu_hip_test.so, loaded by the tests that need a hook; the
  // library that ships, libozimmu_hip.so, carries none: ozimmu_amd/build.py): INT32 diagonal sums
  // [S][N][M] instead of / besides the FP64 epilogue.  The fields stay in both flavours (one argument layout).
  int32_t *dump;
  int dump_only;
  uint32_t rba; // row-blocks held by the A planes (filled in by launch_slice_gemm: rows are padded to TILE_ROWS)
  // strided batch (grid.y = matrix index b): every workspace pointer above (planes, ea, eb, acc) moves by b * ws_stride
  // BYTES, c by b * c_stride ELEMENTS (doubles, or double-complex when cplx); batch <= 1: a single product
  uint32_t batch;
  size_t ws_stride;
  long long c_stride;
  // host side only: the device of the handle this launch belongs to - the launch policy plans with ITS topology (topology.h)
  // and the per-device kernel attributes are set for IT, never for "the current device"
  int device;
};

// a batch of equally shaped operands: matrix b reads its input in_stride doubles after matrix 0 and writes into the
// workspace slot b * ws_stride bytes after slot 0 (grid.z = b)
struct Batch {
  uint32_t count = 1;
  long long in_stride = 0;
  size_t ws_stride = 0;
  size_t exps_stride = 0; // bytes between the exponent words of consecutive matrices
  uint32_t tag = 0;       // see SplitJobs::tag
};

hipError_t launch_slice_gemm(int S, const SliceGemmArgs &a, hipStream_t stream);
// Up to four slice GEMMs that accumulate into the same C in the given order (the real products of a ZGEMM), each of them
// a single launch (one diagonal pass, one K chunk).  One fused launch when the K-split kernel applies (at most one 64x64
// tile per CU over all matrices); hipErrorNotSupported otherwise: the caller launches them one by one.
struct SliceGemmMulti {
  SliceGemmArgs g[4];
  int count;
};
hipError_t launch_slice_gemm_fused(int S, const SliceGemmArgs *g, int count, hipStream_t stream);

// C(complex, m x n, ldc) *= beta  (beta == 0: C = 0 without reading it); init_c_complex, src/gemm.cu:199-239
hipError_t launch_scale_c_complex(size_t m, size_t n, double *c, size_t ldc, double beta_re, double beta_im,
                                  hipStream_t stream, uint32_t batch = 1, long long c_stride = 0);

// element (r, k) of the operand view lives at in[r * stride_r + k * stride_k] (strides in doubles); the smaller
// stride is the contiguous axis: 1 for real operands, 2 for the Re or Im part of an interleaved complex operand
struct OperandView {
  const double *in;
  size_t rows, K;
  size_t stride_r, stride_k;
};

// exps[r] = batch.tag | max over k of the biased exponent field (11 bits) of row r (SplitJobs::tag; tag 0: exps zeroed first)
hipError_t launch_row_max_exp(const OperandView &v, uint32_t *exps, hipStream_t stream, const Batch &batch = Batch(),
                              uint32_t *zero_ptr = nullptr, uint32_t zero_words = 0);

// slices -> tiled planes (layout.h) and max_exp[r] = 2^(e_max+1) (0 for zero rows, NaN for poisoned rows)
hipError_t launch_cut(const OperandView &v, const uint32_t *exps, int S, int L, int8_t *planes,
                      double *max_exp, hipStream_t stream, const Batch &batch = Batch());

// One-pass split (split.hip: split_fused_kernel): up to 4 operand views x `batch` matrices in ONE launch, no exponent
// workspace.  For operands small enough that the re-read of a 32-row strip hits in L2 / Infinity Cache.
struct SplitJob {
  OperandView v;
  int8_t *planes;
  double *max_exp;
  long long in_stride; // batch stride of the input in doubles
  uint32_t *exps;      // two-pass kernels: the view's row exponent words (zeroed before the row-max pass)
};
struct SplitJobs {
  SplitJob job[4];
  uint32_t nblk[4]; // workgroups of each view in this launch
  uint32_t nx[4];   // row_max: row groups per view (block = k chunk * nx + row group); cut: strip length
  uint32_t kchunk[4]; // row_max: k values per workgroup
  int count, S, L;
  size_t ws_stride;
  size_t exps_stride; // bytes between the exponent words of consecutive matrices of a batch
  // Exponent words are written with atomicMax(tag | e), e = 11-bit exponent field, tag = call epoch << 11, and a reader
  // takes a word whose upper bits differ from the tag as "no element seen" (0).  With an epoch that grows from call to
  // call in a buffer that holds nothing but such words, the words never need zeroing; tag = 0 is the plain form for
  // freshly zeroed memory.
  uint32_t tag;
  // words that workgroup 0 of the row-max launch clears on the side (the slice GEMM's phase hints / claim counters, which
  // the GEMM launch two kernel boundaries later expects zeroed): saves the separate zero_words launch of a call
  uint32_t *zero_ptr;
  uint32_t zero_words;
};
// the two streaming passes for up to 4 views x `batch` matrices in one launch each (small problems: launch bound)
// zero `bytes` at the head of `count` workspace slots `pitch` bytes apart (exponent words, phase hints, claim counters)
hipError_t launch_zero_words(void *base, size_t bytes, size_t pitch, uint32_t count, hipStream_t stream);
hipError_t launch_row_max_multi(const SplitJob *job, int count, hipStream_t stream, uint32_t batch = 1,
                                size_t ws_stride = 0, size_t exps_stride = 0, uint32_t tag = 0, uint32_t *zero_ptr = nullptr,
                                uint32_t zero_words = 0);
hipError_t launch_cut_multi(const SplitJob *job, int count, int S, int L, hipStream_t stream, uint32_t batch = 1,
                            size_t ws_stride = 0, size_t exps_stride = 0, uint32_t tag = 0);
hipError_t launch_split_fused(const SplitJob *job, int count, int S, int L, hipStream_t stream, uint32_t batch = 1,
                              size_t ws_stride = 0);

// one read of the operands, strips held in registers (split.hip: split_resident_kernel): every view's K <= resident_split_max_k();
// zero_ptr / zero_words as in launch_row_max_multi; cus: CUs of the device (strip height)
size_t resident_split_max_k();
hipError_t launch_split_resident(const SplitJob *job, int count, int S, int L, hipStream_t stream, uint32_t batch,
                                 size_t ws_stride, uint32_t *zero_ptr, uint32_t zero_words, int cus);

// tiled planes -> reference layout [S][rows][ldo] (test hook for ozimmu_hip_split_int8)
hipError_t launch_untile(const int8_t *planes, size_t rows, size_t K, int S, int8_t *out, size_t ldo,
                         hipStream_t stream);

// auto mode: counters[s-3] += sum over elements of max(0, req - s*L), s = 3..18
hipError_t launch_mantissa_loss(const OperandView &v, const uint32_t *exps, int L,
                                unsigned long long *counters, hipStream_t stream, uint32_t tag = 0);

// ---- convert.hip: the `sgemm` compute mode (src/cublas_helper.cu:20-66) ---------------------------------
// column-major rows x cols scalars; complex matrices pass rows = 2 * complex rows and ld = 2 * complex ld
hipError_t launch_convert_f64_to_f32(float *dst, size_t ldd, const double *src, size_t lds, size_t rows, size_t cols,
                                     hipStream_t stream);
hipError_t launch_convert_f32_to_f64(double *dst, size_t ldd, const float *src, size_t lds, size_t rows, size_t cols,
                                     hipStream_t stream);

} // namespace ozhip
