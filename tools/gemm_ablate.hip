// tools/gemm_ablate.hip — within-process A/B timing of slice_gemm_kernel variants (kernel development tool,
// not part of the library).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iozimmu_amd/csrc tools/gemm_ablate.hip -o gpurun_out/gemm_ablate
//   gpurun_out/gemm_ablate [N=8192] [rounds=5]
// Random INT8 planes (full-range 7-bit magnitudes with random sign, like real slices of U[-1,1) data),
// variants interleaved round-robin, median/min per variant (guide §5.4 rules 24/25).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "slice_gemm_kernel.h"

using namespace ozhip;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

__global__ void fill_planes(int8_t *p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)(i * 2654435761u) ^ seed;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    int v = (int)(x & 127u);
    if (x & 0x100u) v = -v;
    p[i] = (int8_t)v;
  }
}

template <int S, int VAR>
static float run(const SliceGemmArgs &a, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  constexpr size_t lds = 2 * 4 * S * FRAG_BYTES;
  static bool done = false;
  if (!done) {
    CK(hipFuncSetAttribute((const void *)slice_gemm_kernel<S, 0, S, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)lds));
    done = true;
  }
  CK(hipMemsetAsync(a.phase, 0, 8 * 256, st));
  CK(hipEventRecord(e0, st));
  hipLaunchKernelGGL((slice_gemm_kernel<S, 0, S, VAR>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds, st, a);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main(int argc, char **argv) {
  const size_t N = argc > 1 ? std::atol(argv[1]) : 8192;
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 5;
  constexpr int S = 9;
  const size_t M = N, K = N;
  const size_t pa = tiled_plane_bytes(M, K, S), pb = tiled_plane_bytes(N, K, S);
  int8_t *A, *B;
  double *ea, *eb, *C;
  uint32_t *phase;
  CK(hipMalloc(&A, pa));
  CK(hipMalloc(&B, pb));
  CK(hipMalloc(&ea, 8 * M));
  CK(hipMalloc(&eb, 8 * N));
  CK(hipMalloc(&C, 8 * M * N));
  CK(hipMalloc(&phase, 8 * 256));
  hipLaunchKernelGGL(fill_planes, dim3(4096), dim3(256), 0, 0, A, pa, 1u);
  hipLaunchKernelGGL(fill_planes, dim3(4096), dim3(256), 0, 0, B, pb, 2u);
  std::vector<double> ones(std::max(M, N), 1.0);
  CK(hipMemcpy(ea, ones.data(), 8 * M, hipMemcpyHostToDevice));
  CK(hipMemcpy(eb, ones.data(), 8 * N, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());

  SliceGemmArgs a{};
  a.a_planes = A;
  a.b_planes = B;
  a.KB = (uint32_t)k_blocks(K);
  a.kb0 = 0;
  a.kb1 = a.KB;
  a.M = (uint32_t)M;
  a.N = (uint32_t)N;
  a.tiles_m = (uint32_t)((M + 63) / 64);
  a.tiles_n = (uint32_t)((N + 63) / 64);
  a.L = 7;
  a.ea = ea;
  a.eb = eb;
  a.alpha = 1.0;
  a.beta = 0.0;
  a.c = C;
  a.ldc = M;
  a.final = 1;
  a.phase = phase;

  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  struct Var {
    const char *name;
    float (*fn)(const SliceGemmArgs &, hipStream_t, hipEvent_t, hipEvent_t);
    bool nophase;
    std::vector<float> ms;
  };
  std::vector<Var> vars = {
      {"baseline", run<S, 0>, false, {}},
      {"baseline, no phase hint", run<S, 0>, true, {}},
      {"pf2", run<S, VAR_PF2>, false, {}},
      {"pf2 + publish every step", run<S, VAR_PF2 | VAR_PH_EVERY>, false, {}},
      {"pf2 + every step + lead 2", run<S, VAR_PF2 | VAR_PH_EVERY | VAR_PH_LEAD2>, false, {}},
      {"baseline + every step", run<S, VAR_PH_EVERY>, false, {}},
      {"global->regs (no LDS write)", run<S, VAR_GLOBAL_TO_REG>, false, {}},
      {"global->LDS, no sync", run<S, VAR_GLOBAL_NO_SYNC>, false, {}},
      {"no-global (lds+mfma)", run<S, VAR_NO_GLOBAL>, false, {}},
      {"mfma-only", run<S, VAR_MFMA_ONLY>, false, {}},
  };
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vars) {
      SliceGemmArgs b = a;
      if (v.nophase) b.phase = nullptr;
      b.phase = v.nophase ? nullptr : phase;
      if (v.nophase) { // run<> memsets a.phase: keep a valid pointer for the memset, null for the kernel
        SliceGemmArgs c = a;
        (void)c;
      }
      float ms;
      if (v.nophase) {
        CK(hipEventRecord(e0, st));
        constexpr size_t lds = 2 * 4 * S * FRAG_BYTES;
        hipLaunchKernelGGL((slice_gemm_kernel<S, 0, S, 0>), dim3(b.tiles_m * b.tiles_n), dim3(256), lds, st, b);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      } else {
        ms = v.fn(b, st, e0, e1);
      }
      if (r > 0) v.ms.push_back(ms);
    }
  { // correctness of the candidate loop against the baseline loop (bitwise on C)
    std::vector<double> c0(M * N), c1(M * N);
    run<S, 0>(a, st, e0, e1);
    CK(hipMemcpy(c0.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
    CK(hipMemset(C, 0xFF, 8 * M * N));
    run<S, VAR_PF2>(a, st, e0, e1);
    CK(hipMemcpy(c1.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < M * N; i++) bad += c0[i] != c1[i];
    std::printf("check prefetch-2 vs baseline: %zu mismatching elements of %zu\n", bad, M * N);
  }
  const double ops = 45.0 * 2.0 * M * N * K;
  std::printf("N=%zu S=%d rounds=%d  (TOPS = 45*2*N^3 / t)\n", N, S, rounds);
  for (auto &v : vars) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms.front();
    std::printf("%-28s median %8.3f ms (%7.1f TOPS)   min %8.3f ms (%7.1f TOPS)\n", v.name, med, ops / med / 1e9, mn,
                ops / mn / 1e9);
  }
  return 0;
}
