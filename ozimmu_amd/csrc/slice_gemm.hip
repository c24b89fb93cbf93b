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
// launches) is instantiated for ranges of S by slice_gemm_s3_6.hip, _s7_10, _s11_13, _s14_15, _s16_16, _s17_17, _s18_18
// (ranges balanced by compile time: a two-pass S costs three single-pass ones).
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

hipError_t launch_slice_gemm_s3_6(int S, const SliceGemmArgs &a, hipStream_t stream);
hipError_t launch_slice_gemm_fused_s3_6(int S, const SliceGemmArgs *g, int count, hipStream_t stream);
hipError_t launch_slice_gemm_s7_10(int S, const SliceGemmArgs &a, hipStream_t stream);
hipError_t launch_slice_gemm_fused_s7_10(int S, const SliceGemmArgs *g, int count, hipStream_t stream);
hipError_t launch_slice_gemm_s11_13(int S, const SliceGemmArgs &a, hipStream_t stream);
hipError_t launch_slice_gemm_fused_s11_13(int S, const SliceGemmArgs *g, int count, hipStream_t stream);
hipError_t launch_slice_gemm_s14_15(int S, const SliceGemmArgs &a, hipStream_t stream);
hipError_t launch_slice_gemm_fused_s14_15(int S, const SliceGemmArgs *g, int count, hipStream_t stream);
hipError_t launch_slice_gemm_s16_16(int S, const SliceGemmArgs &a, hipStream_t stream);
hipError_t launch_slice_gemm_fused_s16_16(int S, const SliceGemmArgs *g, int count, hipStream_t stream);
hipError_t launch_slice_gemm_s17_17(int S, const SliceGemmArgs &a, hipStream_t stream);
hipError_t launch_slice_gemm_fused_s17_17(int S, const SliceGemmArgs *g, int count, hipStream_t stream);
hipError_t launch_slice_gemm_s18_18(int S, const SliceGemmArgs &a, hipStream_t stream);
hipError_t launch_slice_gemm_fused_s18_18(int S, const SliceGemmArgs *g, int count, hipStream_t stream);

bool slice_gemm_traits_s3_6(int S, int pass, PassTraits *out);
bool slice_gemm_traits_s7_10(int S, int pass, PassTraits *out);
bool slice_gemm_traits_s11_13(int S, int pass, PassTraits *out);
bool slice_gemm_traits_s14_15(int S, int pass, PassTraits *out);
bool slice_gemm_traits_s16_16(int S, int pass, PassTraits *out);
bool slice_gemm_traits_s17_17(int S, int pass, PassTraits *out);
bool slice_gemm_traits_s18_18(int S, int pass, PassTraits *out);

// what the passes of mode S can run on (kernel_policy.h); pass 0: the single / first pass, 1: the second pass of S > 12
bool slice_gemm_traits(int S, int pass, PassTraits *out) {
  if (S >= 3 && S <= 6) return slice_gemm_traits_s3_6(S, pass, out);
  if (S >= 7 && S <= 10) return slice_gemm_traits_s7_10(S, pass, out);
  if (S >= 11 && S <= 13) return slice_gemm_traits_s11_13(S, pass, out);
  if (S >= 14 && S <= 15) return slice_gemm_traits_s14_15(S, pass, out);
  if (S == 16) return slice_gemm_traits_s16_16(S, pass, out);
  if (S == 17) return slice_gemm_traits_s17_17(S, pass, out);
  if (S == 18) return slice_gemm_traits_s18_18(S, pass, out);
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
  }
  if (S >= 3 && S <= 6) return launch_slice_gemm_fused_s3_6(S, g, count, stream);
  if (S >= 7 && S <= 10) return launch_slice_gemm_fused_s7_10(S, g, count, stream);
  if (S >= 11 && S <= 13) return launch_slice_gemm_fused_s11_13(S, g, count, stream);
  return hipErrorNotSupported; // two diagonal passes per product
}

hipError_t launch_slice_gemm(int S, const SliceGemmArgs &a_in, hipStream_t stream) {
  SliceGemmArgs a = a_in;
  a.nxcd = (uint32_t)topology(a.device).xcds; // the kernels' tile partition and per-XCD lines follow the device (topology.h)
  a.phase_min_kb = (uint32_t)config().phase_min_kb;
  a.spec_claim_kb = (uint32_t)config().spec_claim_kb;
  if (S >= 3 && S <= 6) return launch_slice_gemm_s3_6(S, a, stream);
  if (S >= 7 && S <= 10) return launch_slice_gemm_s7_10(S, a, stream);
  if (S >= 11 && S <= 13) return launch_slice_gemm_s11_13(S, a, stream);
  if (S >= 14 && S <= 15) return launch_slice_gemm_s14_15(S, a, stream);
  if (S >= 16 && S <= 16) return launch_slice_gemm_s16_16(S, a, stream);
  if (S >= 17 && S <= 17) return launch_slice_gemm_s17_17(S, a, stream);
  if (S >= 18 && S <= 18) return launch_slice_gemm_s18_18(S, a, stream);
  return hipErrorInvalidValue;
}

} // namespace ozhip
