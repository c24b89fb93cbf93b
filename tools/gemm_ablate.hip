// tools/gemm_ablate.hip — within-process A/B timing of slice_gemm_kernel variants (kernel development tool,
// not part of the library).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iozimmu_amd/csrc -Itools tools/gemm_ablate.hip -o tools/bin/gemm_ablate
//   tools/bin/gemm_ablate [N=8192] [rounds=5] [plane value mask: 127 = full-entropy slices, 1 = low-toggle data]
// Random INT8 planes (full-range 7-bit magnitudes with random sign, like real slices of U[-1,1) data),
// variants interleaved round-robin, median/min per variant (guide §5.4 rules 24/25).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "slice_gemm_pp_kernel.h" // tools/: experiment, not part of the library

using namespace ozhip;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

__global__ void fill_planes(int8_t *p, size_t n, unsigned seed, unsigned mask) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)(i * 2654435761u) ^ seed;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    int v = (int)(x & mask); // mask 127: full-entropy slices; 1: low-toggle data (power probe)
    if (x & 0x100u) v = -v;
    p[i] = (int8_t)v;
  }
}

template <int S, int VAR, int WM = 2>
static float run(const SliceGemmArgs &a0, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  constexpr size_t lds = 2 * (WM + 2) * S * FRAG_BYTES;
  SliceGemmArgs a = a0;
  a.tiles_m = (a.M + 32 * WM - 1) / (32 * WM);
  static bool done = false;
  if (!done) {
    CK(hipFuncSetAttribute((const void *)slice_gemm_kernel<S, 0, S, VAR, WM>,
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done = true;
  }
  CK(hipMemsetAsync(a.phase, 0, 8 * 256, st));
  CK(hipEventRecord(e0, st));
  hipLaunchKernelGGL((slice_gemm_kernel<S, 0, S, VAR, WM>), dim3(a.tiles_m * a.tiles_n), dim3(128 * WM), lds, st, a);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

template <int S, int VAR>
static float run_throttled(const SliceGemmArgs &a0, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  SliceGemmArgs a = a0;
  a.throttle = 1;
  return run<S, VAR>(a, st, e0, e1);
}

template <int S, int VAR>
static float run_pp(const SliceGemmArgs &a0, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  constexpr size_t lds = 2 * 6 * S * FRAG_BYTES + ((VAR & 1) ? 8192 : 0);
  SliceGemmArgs a = a0;
  a.tiles_m = (a.M + 63) / 64;
  a.tiles_n = (a.N + 127) / 128;
  static bool done = false;
  if (!done) {
    CK(hipFuncSetAttribute((const void *)slice_gemm_pp_kernel<S, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)lds));
    done = true;
  }
  CK(hipMemsetAsync(a.phase, 0, 8 * 256, st));
  CK(hipEventRecord(e0, st));
  hipLaunchKernelGGL((slice_gemm_pp_kernel<S, VAR>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
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
  const unsigned mask = argc > 3 ? (unsigned)std::atoi(argv[3]) : 127u;
  hipLaunchKernelGGL(fill_planes, dim3(4096), dim3(256), 0, 0, A, pa, 1u, mask);
  hipLaunchKernelGGL(fill_planes, dim3(4096), dim3(256), 0, 0, B, pb, 2u, mask);
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
      {"shipped 64x64", run<S, VAR_SHIPPED>, false, {}},
      {"ping-pong 64x128", run_pp<S, 0>, false, {}},
      {"ping-pong 64x128 MUBUF", run_pp<S, 2>, false, {}},
      {"shipped + lead throttle", run_throttled<S, VAR_SHIPPED>, false, {}},
      {"prefetch-1 loop (as S >= 11)", run_throttled<S, VAR_SHIPPED & ~VAR_PF2>, false, {}},
      {"throttle, plain patch order", run_throttled<S, VAR_SHIPPED | VAR_NO_CU_SWIZZLE>, false, {}},
      {"shipped, L2-hot addresses", run<S, VAR_SHIPPED | VAR_HOT>, false, {}},
      {"shipped, L1-hot addresses", run<S, VAR_SHIPPED | VAR_HOT1>, false, {}},
      {"64x64 no-global", run<S, VAR_NO_GLOBAL>, false, {}},
      {"mfma-only low-entropy regs", run<S, VAR_MFMA_ONLY>, false, {}},
      {"mfma-only random regs", run<S, VAR_MFMA_ONLY | VAR_RAND_REGS>, false, {}},
  };
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vars) {
      const float ms = v.fn(a, st, e0, e1);
      if (r > 0) v.ms.push_back(ms);
    }
  { // correctness of the candidate loop against the baseline loop (bitwise on C)
    std::vector<double> c0(M * N), c1(M * N);
    run<S, 0>(a, st, e0, e1);
    CK(hipMemcpy(c0.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
    CK(hipMemset(C, 0xFF, 8 * M * N));
    for (int which = 0; which < 3; which++) {
      CK(hipMemset(C, 0xFF, 8 * M * N));
      if (which == 0) run<S, VAR_SHIPPED>(a, st, e0, e1);
      if (which == 1) run_pp<S, 0>(a, st, e0, e1);
      if (which == 2) run_throttled<S, VAR_SHIPPED>(a, st, e0, e1);
      CK(hipMemcpy(c1.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < M * N; i++) bad += c0[i] != c1[i];
      std::printf("check variant %d vs plain loop: %zu mismatching elements of %zu\n", which, bad, M * N);
    }
  }
  for (int which = 0; which < 2; which++) { // per-phase trace of the ping-pong kernel: first round, mid-kernel
    unsigned long long *tr;
    const size_t ntr = 64 * 8 * 16 * 8;
    CK(hipMalloc(&tr, ntr * 8));
    CK(hipMemset(tr, 0, ntr * 8));
    SliceGemmArgs b = a;
    b.trace = tr;
    b.trace_block0 = which ? 4096 : 0;
    run_pp<S, 3>(b, st, e0, e1);
    std::vector<unsigned long long> h(ntr);
    CK(hipMemcpy(h.data(), tr, ntr * 8, hipMemcpyDeviceToHost));
    const char *names[2][7] = {{"45 MFMA", "wait vmcnt", "barrier", "copy issue", "frag reads", "wait lgkm", "barrier"},
                               {"copy issue", "frag reads", "wait lgkm", "barrier", "45 MFMA", "wait vmcnt", "barrier"}};
    for (int g = 0; g < 2; g++) {
      double sum[7] = {0}, tot = 0;
      int cnt = 0;
      for (int blk = 0; blk < 64; blk++)
        for (int w = 4 * g; w < 4 * g + 4; w++)
          for (int it = 0; it < 15; it++) {
            const unsigned long long *t = &h[((size_t)(blk * 8 + w) * 16 + it) * 8];
            const unsigned long long *tn = t + 8;
            if (!t[0] || !tn[0]) continue;
            for (int k = 0; k < 6; k++) sum[k] += (double)(t[k + 1] - t[k]);
            sum[6] += (double)(tn[0] - t[6]);
            tot += (double)(tn[0] - t[0]);
            cnt++;
          }
      std::printf("pp trace G%d, workgroups %u..: %.0f ticks per k-step (%d samples):", g, b.trace_block0, tot / cnt, cnt);
      for (int k = 0; k < 7; k++) std::printf("  %s %.0f", names[g][k], sum[k] / cnt);
      std::printf("\n");
    }
    CK(hipFree(tr));
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
