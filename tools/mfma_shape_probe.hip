// tools/mfma_shape_probe.hip — sustained INT8 MFMA rate of the two gfx950 shapes on full-entropy operands (the part is
// power limited there: DESIGN.md §4.2).  One wave per SIMD, operands in registers, no memory traffic:
//   v_mfma_i32_32x32x32_i8   32 K MAC, accumulator 16 registers (read + written per instruction)
//   v_mfma_i32_16x16x64_i8   16 K MAC, accumulator  4 registers
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_shape_probe.hip -o tools/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v4i rnd(unsigned seed, unsigned mask) {
  v4i r;
  unsigned x = seed * 2654435761u + 12345u;
  for (int c = 0; c < 4; c++) {
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    r[c] = (int)(x & mask);
  }
  return r;
}

template <int SHAPE, int NACC, int NOP>
__global__ __launch_bounds__(256) void probe(int iters, unsigned mask, int *out) {
  v4i a[NOP], b[NOP];
  for (int i = 0; i < NOP; i++) {
    a[i] = rnd(threadIdx.x * 131u + i * 7u + blockIdx.x, mask);
    b[i] = rnd(threadIdx.x * 977u + i * 13u + blockIdx.x * 3u, mask);
    asm volatile("" : "+v"(a[i]), "+v"(b[i]));
  }
  if constexpr (SHAPE == 32) {
    v16i acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < NACC; i++)
        acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i % NOP], b[(i * 5 + 1) % NOP], acc[i], 0, 0, 0);
    int s = 0;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    if (s == 0x12345678) out[0] = s;
  } else {
    v4i acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < NACC; i++)
        acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i % NOP], b[(i * 5 + 1) % NOP], acc[i], 0, 0, 0);
    int s = 0;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
    if (s == 0x12345678) out[0] = s;
  }
}

template <int SHAPE, int NACC, int NOP>
static double run(int iters, unsigned mask, int *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ms;
  for (int r = 0; r < 6; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHAPE, NACC, NOP>), dim3(256 * 4), dim3(256), 0, 0, iters, mask, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    if (r) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double macs = (SHAPE == 32 ? 32768.0 : 16384.0) * NACC * (double)iters * 4 /*waves*/ * 1024 /*blocks*/;
  return 2.0 * macs / (ms[ms.size() / 2] * 1e-3) / 1e12;
}

int main() {
  int *out; hipMalloc(&out, 4);
  for (unsigned mask : {0xFFFFFFFFu, 0x7F7F7F7Fu, 0x01010101u, 0u}) {
    // same number of MACs per launch in both shapes; enough accumulators that no instruction waits for its own result
    const double t32 = run<32, 12, 9>(6000, mask, out);
    const double t16 = run<16, 24, 9>(6000, mask, out);
    const double t16b = run<16, 48, 9>(3000, mask, out);
    std::printf("operand byte mask 0x%08x: 32x32x32 %7.1f TOPS | 16x16x64 (24 acc) %7.1f TOPS | 16x16x64 (48 acc) %7.1f TOPS\n",
                mask, t32, t16, t16b);
  }
  return 0;
}
