// tools/cvt_probe.hip — issue cost of the INT32 -> FP64 conversion in the slice GEMM's epilogue (development probe):
// v_cvt_f64_i32 + fma per element vs the magic-number form (xor + one FP64 add) + fma, one wave per SIMD like the wide
// kernel's epilogue.   hipcc --offload-arch=gfx950 -O3 tools/cvt_probe.hip -o tools/bin/cvt_probe
#include <hip/hip_runtime.h>

#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void probe(const int *in, double *out, int iters, double sc) {
  int v[16];
  for (int i = 0; i < 16; i++) v[i] = in[threadIdx.x * 16 + i];
  double acc[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      double d;
      if (MODE == 0) {
        d = (double)v[i];
      } else if (MODE == 1) {
        d = __hiloint2double(0x43300000, v[i] ^ 0x80000000) - 4503601774854144.0; // 2^52 + 2^31
      } else {
        d = __hiloint2double(v[i], v[i]); // no conversion at all: the fma chain alone
      }
      acc[i & 3] = fma(d, sc, acc[i & 3]);
      v[i] += it;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  int *in;
  double *out;
  hipMalloc(&in, 256 * 16 * 4);
  hipMemset(in, 1, 256 * 16 * 4);
  hipMalloc(&out, 256 * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9;
    for (int r = 0; r < 5; r++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, in, out, iters, 0.5);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, in, out, iters, 0.5);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, in, out, iters, 0.5);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double per = best * 1e-3 / (iters * 16.0) * 2.1e9; // cycles per element per wave at ~2.1 GHz
    printf("%s: %.3f ms, ~%.1f cycles per (convert + fma + int add) per wave\n",
           mode == 0 ? "v_cvt_f64_i32" : mode == 1 ? "magic number (xor + add_f64)" : "no conversion", best, per);
  }
  return 0;
}
