// tools/valu_cover_probe.hip — which VALU instructions does a wave's own v_mfma_i32_16x16x64_i8 stream cover?  (round 6)
// The recombination of the Ozaki scheme is FP64 VALU work (v_cvt_f64_i32 + v_fma_f64 per output and diagonal); the k64 register
// kernel issues it between the MFMAs of its last k-step, and the tile trace (profiles/r6_ablate/) shows that step taking the SUM
// of its MFMA time and the chain's time.  One wave per SIMD, 12 independent accumulators, F independent filler instructions of
// one kind behind every MFMA; cycles per MFMA slot with s_memtime: MFMA only / fillers only / both.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_cover_probe.hip -o tools/bin/valu_cover_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

enum Kind { FMA_F64, CVT_F64_I32, ACC_READ, FMA_F32, ADD_U32, MUL_F64, ADD_F64, CVT_FMA, MAD_U64, LDEXP_F64, NKIND };
static const char *kind_name[NKIND] = {"v_fma_f64", "v_cvt_f64_i32", "v_accvgpr_read_b32", "v_fma_f32", "v_add_u32", "v_mul_f64", "v_add_f64",
                                       "v_cvt_f64_i32 + v_fma_f64 (alternating)", "v_mad_u64_u32", "v_ldexp_f64"};

template <int KIND, int F, bool MFMA, bool FILL, bool AGPR = false>
__global__ __launch_bounds__(256) void probe(int iters, unsigned long long *out, double *sink) {
  v4i a, b, acc[12];
  unsigned x = threadIdx.x * 2654435761u + 12345u;
  for (int c = 0; c < 4; c++) {
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    a[c] = (int)x;
    b[c] = (int)(x * 2654435761u);
  }
  for (int i = 0; i < 12; i++) acc[i] = v4i{0, 0, 0, 0};
  double d[8];
  float f[8];
  unsigned u[8];
  unsigned long long q[8];
  int r[8];
  for (int i = 0; i < 8; i++) {
    d[i] = 1.0 + i * 1e-3 + threadIdx.x * 1e-6;
    f[i] = 1.0f + i * 1e-3f;
    u[i] = i + threadIdx.x;
    q[i] = i;
    r[i] = i;
  }
  const double c1 = 0.999999, c2 = 1e-9;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
      if constexpr (MFMA && !AGPR) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if constexpr (MFMA && AGPR) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      if constexpr (FILL) {
#pragma unroll
        for (int k = 0; k < F; k++) {
          const int j = (i * F + k) % 8;
          if constexpr (KIND == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j]) : "v"(c1), "v"(c2));
          if constexpr (KIND == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[j]) : "v"(r[j]));
          if constexpr (KIND == ACC_READ) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(r[j]) : "i"(200 + j) : "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207");
          if constexpr (KIND == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"((float)c1), "v"((float)c2));
          if constexpr (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(r[j]));
          if constexpr (KIND == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[j]) : "v"(c1));
          if constexpr (KIND == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(c2));
          if constexpr (KIND == CVT_FMA) {
            if (k & 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j]) : "v"(c1), "v"(c2));
            else asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[(j + 4) % 8]) : "v"(r[j]));
          }
          if constexpr (KIND == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[j]) : "v"(u[j]), "v"(u[(j + 1) % 8]) : "vcc");
          if constexpr (KIND == LDEXP_F64) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[j]) : "v"(r[j] & 1));
        }
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double s = 0;
  for (int i = 0; i < 8; i++) s += d[i] + f[i] + u[i] + (double)q[i] + r[i];
  int si = 0;
  for (int i = 0; i < 12; i++) si += acc[i][0];
  if (s == 1.2345e300 || si == 0x7fffffff) sink[0] = s + si;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND, int F, bool AGPR = false>
static void run(unsigned long long *d_out, double *sink) {
  const int iters = 2000;
  double cyc[3];
  for (int v = 0; v < 3; v++) {
    for (int rep = 0; rep < 2; rep++) {
      if (v == 0) hipLaunchKernelGGL((probe<KIND, F, true, false, AGPR>), dim3(256), dim3(256), 0, 0, iters, d_out, sink);
      if (v == 1) hipLaunchKernelGGL((probe<KIND, F, false, true, AGPR>), dim3(256), dim3(256), 0, 0, iters, d_out, sink);
      if (v == 2) hipLaunchKernelGGL((probe<KIND, F, true, true, AGPR>), dim3(256), dim3(256), 0, 0, iters, d_out, sink);
      hipDeviceSynchronize();
    }
    unsigned long long h = 0;
    hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    cyc[v] = (double)h / (iters * 12.0);
  }
  std::printf("%s%-42s x %d per MFMA: MFMA only %6.2f | fillers only %6.2f | both %6.2f cycles per slot  -> covered %5.1f %% of the fillers\n", AGPR ? "[MFMA accumulators in AGPRs] " : "", kind_name[KIND], F,
              cyc[0], cyc[1], cyc[2], 100.0 * (1.0 - (cyc[2] - cyc[0]) / cyc[1]));
}

int main() {
  unsigned long long *d_out;
  double *sink;
  hipMalloc(&d_out, 64);
  hipMalloc(&sink, 64);
  run<FMA_F64, 1>(d_out, sink);
  run<FMA_F64, 2>(d_out, sink);
  run<CVT_F64_I32, 1>(d_out, sink);
  run<CVT_F64_I32, 2>(d_out, sink);
  run<CVT_FMA, 2>(d_out, sink);
  run<ACC_READ, 2>(d_out, sink);
  run<MUL_F64, 2>(d_out, sink);
  run<ADD_F64, 2>(d_out, sink);
  run<LDEXP_F64, 2>(d_out, sink);
  run<FMA_F32, 2>(d_out, sink);
  run<FMA_F32, 3>(d_out, sink);
  run<ADD_U32, 3>(d_out, sink);
  run<MAD_U64, 2>(d_out, sink);
  // the kernel's case: the MFMAs accumulate in AGPRs and the recombination reads (other) AGPRs
  run<ACC_READ, 1, true>(d_out, sink);
  run<ACC_READ, 2, true>(d_out, sink);
  run<FMA_F64, 2, true>(d_out, sink);
  run<CVT_FMA, 2, true>(d_out, sink);
  run<ADD_U32, 2, true>(d_out, sink);
  return 0;
}
