// tools/l2_bw.hip — per-CU streaming bandwidth out of L2 (development probe).
// 512 workgroups x 256 threads (2 per CU, like slice_gemm_kernel<9>) each re-read an L2-resident window with
// (a) global_load_dwordx4 into registers, (b) global_load_lds_dwordx4 into LDS; reports B/clk/CU at the measured
// time assuming 2.4 GHz and the aggregate TB/s.   hipcc --offload-arch=gfx950 -O3 tools/l2_bw.hip -o tools/bin/l2_bw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

template <int MODE>
__global__ __launch_bounds__(256) void stream(const char *base, size_t window, int iters, int *sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // each XCD (blockIdx & 7) reads its own window so that all of its reads hit its L2 after the first sweep
  const char *p = base + (size_t)(blockIdx.x & 7) * window;
  const size_t chunks = window / 1024; // 1 KiB per wave instruction
  size_t c = ((size_t)blockIdx.x * 4 + wave) * 9 % chunks;
  v4i acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < 9; s++) {
      const char *g = p + c * 1024 + lane * 16;
      if (MODE == 0) {
        const v4i v = *(const v4i *)g;
        acc ^= v;
      } else {
        __builtin_amdgcn_global_load_lds((const AS1 void *)g, (AS3 void *)(smem + wave * 9216 + s * 1024), 16, 0, 0);
      }
      c = c + 1 == chunks ? 0 : c + 1;
    }
    if (MODE == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | 9); // vmcnt(9): keep one group in flight
  }
  if (MODE == 1) acc[0] = *(const int *)(smem + threadIdx.x * 4);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) sink[0] = 1;
}

int main(int argc, char **argv) {
  const size_t window = (argc > 1 ? atol(argv[1]) : 2) << 20; // MiB per XCD
  const int iters = argc > 2 ? atoi(argv[2]) : 2000;
  char *buf;
  int *sink;
  hipMalloc(&buf, window * 8);
  hipMemset(buf, 1, window * 8);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++)
    for (int blocks : {256, 512, 1024}) {
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0)
          hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(256), 0, 0, buf, window, iters, sink);
        else
          hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(256), 36864, 0, buf, window, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) {
          const double bytes = (double)blocks * 4 * 9 * 1024.0 * iters;
          std::printf("%s window %zu MiB/XCD, %4d WGs: %.3f ms  %.2f TB/s  %.1f GB/s/CU  (%.1f B/clk/CU @2.4GHz)\n",
                      mode ? "global_load_lds" : "global_load    ", window >> 20, blocks, ms, bytes / ms / 1e9,
                      bytes / ms / 1e6 / 256, bytes / (ms * 1e-3) / 256 / 2.4e9);
        }
      }
    }
  return 0;
}
