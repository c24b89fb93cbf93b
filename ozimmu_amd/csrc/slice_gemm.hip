// slice_gemm.hip — INT8 x INT8 -> INT32 slice products + error-free FP64 recombination, fused.
//
// Replaces, for the real DGEMM path, the reference's hot loop
//   for each pair (i,j), i+j <= S+1:  cublasGemmEx(INT8->INT32)  (/root/reference/src/gemm.cu:315-329)
//                                     accumulate_in_f64            (src/gemm.cu:77-102, :394-401)
//   axby                                                           (src/gemm.cu:124-158)
// by ONE kernel per pass that keeps its output tile resident in registers -- three kernels share the arithmetic below
// and the epilogue: slice_gemm_w_kernel.h (one workgroup per CU, (32*WA) x 128 tiles, persistent: the kernel of every
// problem that fills the chip), slice_gemm_k2_kernel.h (64x64 tiles, K split inside an 8-wave workgroup: launches with
// at most one tile per CU) and slice_gemm_kernel.h (64x64 / 128x64 tiles, two workgroups per CU: short k loops, few
// slices); slice_gemm_launch.h chooses.  In all of them:
//
//  * every k-step (32 k-bytes) the workgroup stages ALL needed slices of its A-rows and B-rows
//    HBM -> LDS with global_load_lds (1 KiB fragment blocks, layout.h), double buffered;
//  * a wave owns 32x32 sub-tiles and issues one v_mfma_i32_32x32x32_i8 per slice pair and sub-tile,
//    accumulating pairs of equal i+j (same power-of-two weight) into the SAME INT32 accumulator:
//    S accumulators instead of S(S+1)/2 products.  Integer addition is exact, so this only regroups
//    the reference's sum; the INT32 bound count*K*(2^L-1)^2 < 2^31 is enforced by the host through
//    K-chunking (api.cpp: max_k_per_pass);
//  * the epilogue converts each diagonal sum to FP64, applies the reference's power-of-two weight
//    2^(46 - L*(i+j)) with one fma per diagonal (t ascending, largest first), then the reference's
//    `acc / 2^44 * eA[m] * eB[n]`, alpha and beta, and writes C once.
//
// Operand roles are swapped in the MFMA (B-slices as the "A" operand, A-slices as the "B" operand) so
// that a lane's accumulator registers hold one column index m = lane&31 and 16 different n: for a fixed
// register the 32 lanes of a half-wave store 32 consecutive doubles of column-major C (256 B runs).
//
// Roofline: MFMA INT8 (dense 1024 MAC/clk/SIMD).  Algorithmic work per launch: P*2*M*N*K ops.
//
// Translation units: this file (dispatch on S, the complex beta scaling); slice_gemm_launch.h (kernel choice, tile plan,
// launches) is instantiated per compute mode by the slice_gemm_s<lo>_<hi>.hip units listed in OZ_GEMM_PARTS below.
#include <hip/hip_runtime.h>

#include "config.h"
#include "kernel_policy.h"
#include "kernels.h"
#include "topology.h"

namespace ozhip {

// init_c_complex_kernel (src/gemm.cu:199-223) without its read-after-write bug (:218-219 uses the updated c.x
// for c.y): both components are computed from the ORIGINAL c.
__global__ void scale_c_complex_kernel(size_t m, size_t n, double2 *c, size_t ldc, double br, double bi, int zero,
                                       long long c_stride) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= m * n) return;
  c += (long long)blockIdx.y * c_stride; // batch
  double2 *p = c + (tid / m) * ldc + tid % m;
  if (zero) {
    *p = make_double2(0.0, 0.0);
  } else {
    const double2 v = *p;
    *p = make_double2(fma(v.x, br, -(v.y * bi)), fma(v.y, br, v.x * bi));
  }
}

hipError_t launch_scale_c_complex(size_t m, size_t n, double *c, size_t ldc, double beta_re, double beta_im,
                                  hipStream_t stream, uint32_t batch, long long c_stride) {
  if (m * n == 0 || batch == 0) return hipSuccess;
  hipLaunchKernelGGL(scale_c_complex_kernel, dim3((unsigned)((m * n + 255) / 256), batch), dim3(256), 0, stream, m, n,
                     reinterpret_cast<double2 *>(c), ldc, beta_re, beta_im, (beta_re == 0.0 && beta_im == 0.0) ? 1 : 0,
                     c_stride);
  return hipGetLastError();
}

// the translation units of the slice GEMM: X(first S, last S, suffix) - one slice_gemm_<suffix>.hip each (ozimmu_amd/build.py:
// GEMM_PARTS lists the same files), most of them one compute mode: the build is bound by its slowest unit, and a two-pass
// mode instantiates three to four times the kernels of a single-pass one
#define OZ_GEMM_PARTS(X) X(3, 4, s3_4) X(5, 6, s5_6) X(7, 7, s7_7) X(8, 8, s8_8) X(9, 9, s9_9) X(10, 10, s10_10) X(11, 11, s11_11) X(12, 12, s12_12) X(13, 13, s13_13) X(14, 14, s14_14) X(15, 15, s15_15) X(16, 16, s16_16) X(17, 17, s17_17) X(18, 18, s18_18)

#define OZ_DECLARE_PART(LO, HI, NAME)                                                                          \
  hipError_t launch_slice_gemm_##NAME(int S, const SliceGemmArgs &a, hipStream_t stream);                      \
  hipError_t launch_slice_gemm_fused_##NAME(int S, const SliceGemmArgs *g, int count, hipStream_t stream);     \
  bool slice_gemm_traits_##NAME(int S, int pass, PassTraits *out);
OZ_GEMM_PARTS(OZ_DECLARE_PART)
#undef OZ_DECLARE_PART

// what the passes of mode S can run on (kernel_policy.h); pass 0: the single / first pass, 1: the second pass of S > 12
bool slice_gemm_traits(int S, int pass, PassTraits *out) {
#define OZ_TRAITS_PART(LO, HI, NAME) \
  if (S >= LO && S <= HI) return slice_gemm_traits_##NAME(S, pass, out);
  OZ_GEMM_PARTS(OZ_TRAITS_PART)
#undef OZ_TRAITS_PART
  return false;
}

hipError_t launch_slice_gemm_fused(int S, const SliceGemmArgs *g_in, int count, hipStream_t stream) {
  if (count < 1 || count > 4) return hipErrorNotSupported;
  SliceGemmArgs g[4];
  const uint32_t nx = (uint32_t)topology(g_in[0].device).xcds;
  for (int i = 0; i < count; i++) {
    g[i] = g_in[i];
    g[i].nxcd = nx;
    g[i].phase_min_kb = (uint32_t)config().phase_min_kb;
    g[i].spec_claim_kb = (uint32_t)config().spec_claim_kb;
    g[i].epi_overlap = config().epi_overlap ? 1u : 0u;
  }
#define OZ_FUSED_PART(LO, HI, NAME) \
  if (S >= LO && S <= HI) return launch_slice_gemm_fused_##NAME(S, g, count, stream);
  OZ_GEMM_PARTS(OZ_FUSED_PART)
#undef OZ_FUSED_PART
  return hipErrorNotSupported; // two diagonal passes per product
}

hipError_t launch_slice_gemm(int S, const SliceGemmArgs &a_in, hipStream_t stream) {
  SliceGemmArgs a = a_in;
  a.nxcd = (uint32_t)topology(a.device).xcds; // the kernels' tile partition and per-XCD lines follow the device (topology.h)
  a.phase_min_kb = (uint32_t)config().phase_min_kb;
  a.spec_claim_kb = (uint32_t)config().spec_claim_kb;
  a.epi_overlap = config().epi_overlap ? 1u : 0u;
#define OZ_LAUNCH_PART(LO, HI, NAME) \
  if (S >= LO && S <= HI) return launch_slice_gemm_##NAME(S, a, stream);
  OZ_GEMM_PARTS(OZ_LAUNCH_PART)
#undef OZ_LAUNCH_PART
  return hipErrorInvalidValue;
}

} // namespace ozhip
