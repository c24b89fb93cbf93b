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
#include <cstring>
#include <map>
#include <vector>

#include "slice_gemm_w_kernel.h"

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

// wide kernel (one 4-wave workgroup per CU, WA blocks per wave)
template <int S, int WA, int VARW, int STAG = 0, int DMA0 = -1, int DMAE = 4, int TAIL = 6, bool MIX = false, bool DYN = false>
static float run_w(const SliceGemmArgs &a0, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  constexpr int NA = (VARW & VARW_NA3) ? 3 : 2;
  constexpr int NBUF_B = (VARW & VARW_B1) ? 1 : 2;
  constexpr size_t lds = (VARW & VARW_K64) ? (size_t)(2 * WA + ((VARW & VARW_BREG) ? 0 : 4 * NBUF_B)) * (2 * S) * FRAG_BYTES
                                           : (size_t)(NA * WA + NBUF_B * 4) * S * FRAG_BYTES + ((VARW & VARW_X16) ? 2 * X_PAD : 0);
  SliceGemmArgs a = a0;
  a.tiles_m = (a.M + 32 * WA - 1) / (32 * WA);
  a.tiles_m2 = 0;
  if (MIX) { // as many (WA-1)-block rows as make the big region a whole number of 256-CU rounds (8192: 80 + 8)
    const uint32_t rows32 = (a.M + 31) / 32;
    for (uint32_t n2 = 0; n2 * (WA - 1) <= rows32; n2++) {
      const uint32_t n3 = (rows32 - n2 * (WA - 1) + WA - 1) / WA;
      if ((n3 * ((a.N + 127) / 128)) % 256 == 0 && n3 * WA + n2 * (WA - 1) == rows32) {
        a.tiles_m = n3;
        a.tiles_m2 = n2;
        break;
      }
    }
  }
  a.tiles_n = (a.N + 127) / 128;
  a.rba = (uint32_t)row_blocks_padded(a.M);
  static bool done = false;
  if (!done) {
    CK(hipFuncSetAttribute((const void *)slice_gemm_w_kernel<S, 0, S, WA, VARW, STAG, DMA0, DMAE, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)lds));
    done = true;
  }
  CK(hipMemsetAsync(a.phase, 0, 8 * 256, st));
  uint32_t grid = (a.tiles_m + a.tiles_m2) * a.tiles_n;
  a.queue = nullptr;
  if (DYN) {
    a.queue = a.phase + 16;
    grid = 256;
  }
  CK(hipEventRecord(e0, st));
  hipLaunchKernelGGL((slice_gemm_w_kernel<S, 0, S, WA, VARW, STAG, DMA0, DMAE, TAIL>), dim3(grid), dim3(256), lds, st, a);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main(int argc, char **argv) {
  const size_t N = argc > 1 ? std::atol(argv[1]) : 8192;
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 5;
#ifndef ABLATE_S
#define ABLATE_S 9
#endif
  constexpr int S = ABLATE_S; // -DABLATE_S=8: the schedule parameters at another slice count
  const size_t M = argc > 4 ? std::atol(argv[4]) : N, K = argc > 5 ? std::atol(argv[5]) : N;
  const size_t pa = tiled_plane_bytes(M, K, S), pb = tiled_plane_bytes(N, K, S);
  int8_t *A, *B;
  double *ea, *eb, *C;
  uint32_t *phase;
  CK(hipMalloc(&A, pa));
  CK(hipMalloc(&B, pb));
  CK(hipMalloc(&ea, 8 * M));
  CK(hipMalloc(&eb, 8 * N));
  CK(hipMalloc(&C, 8 * M * N));
  CK(hipMalloc(&phase, PHASE_LINES_BYTES));
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
  a.nxcd = 8;
  a.phase_min_kb = 32;
  a.spec_claim_kb = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 64u; // argv[6]: claim-ahead threshold in k-blocks (0: off)

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
#if defined(ABLATE_TILE_TRACE)
  // -DABLATE_TILE_TRACE (round 6): where a workgroup of the headline kernel (k64, B in registers, named
  // accumulators, recombination under the last step, persistent with the ticket drawn one tile ahead) spends its time ACROSS tile
  // boundaries.  Run:  tools/bin/gemm_ablate_ttrace N rounds mask M K spec_kb [1: drain the stores before the tile's last stamp]
  //   8192^2 x 256:   tools/bin/gemm_ablate_ttrace 8192 5 127 8192 256
  //   32768^2 x 1024: tools/bin/gemm_ablate_ttrace 32768 3 127 32768 1024
  {
    constexpr int Z = VARW_K64 | VARW_BREG | VARW_ACCN;
    (void)sizeof(Var);
    unsigned long long *trace = nullptr;
    const size_t trace_words = 16384 + (size_t)256 * 40 * 16;
    CK(hipMalloc(&trace, trace_words * 8));
    a.acc = reinterpret_cast<double *>(trace);
    a.epi_overlap = 1;
    a.dump_only = argc > 7 ? std::atoi(argv[7]) : 0;
    std::vector<float> plain, traced;
    for (int r = 0; r < rounds + 1; r++) {
      const float t0 = run_w<S, 2, Z, 0, -1, 8, 12, false, true>(a, st, e0, e1);
      CK(hipMemsetAsync(trace, 0, trace_words * 8, st));
      const float t1 = run_w<S, 2, Z | VARW_TRACE, 0, -1, 8, 12, false, true>(a, st, e0, e1);
      if (r) {
        plain.push_back(t0);
        traced.push_back(t1);
      }
    }
    if (M * N <= (size_t)8192 * 8192) { // bitwise: the overlapped recombination against the plain epilogue of the LDS-staged form
      constexpr int Y = VARW_K64 | VARW_B1;
      std::vector<double> c0(M * N), c1(M * N);
      CK(hipMemset(C, 0xFF, 8 * M * N));
      run_w<S, 2, Y, 0, -1, 8, 12, false, true>(a, st, e0, e1);
      CK(hipMemcpy(c0.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
      CK(hipMemset(C, 0xFF, 8 * M * N));
      run_w<S, 2, Z, 0, -1, 8, 12, false, true>(a, st, e0, e1);
      CK(hipMemcpy(c1.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < M * N; i++) bad += std::memcmp(&c0[i], &c1[i], 8) != 0;
      std::printf("overlapped recombination (B->VGPR) vs plain epilogue (B via LDS): %zu mismatching elements of %zu\n", bad, M * N);
    }
    std::sort(plain.begin(), plain.end());
    std::sort(traced.begin(), traced.end());
    const double ops = (double)(S * (S + 1) / 2) * 2.0 * M * N * K;
    std::printf("# %zu x %zu x %zu, S = %d, k64 B->VGPR named accumulators, overlapped last step, persistent (claim ahead <= %u k-blocks)%s\n", M, N, K, S,
                a.spec_claim_kb, a.dump_only & 1 ? ", stores DRAINED before the tile's last stamp" : "");
    std::printf("kernel: %.3f ms (%.1f TOPS) untraced, %.3f ms with the stamps (median of %d)\n", plain[plain.size() / 2],
                ops / plain[plain.size() / 2] / 1e9, traced[traced.size() / 2], rounds);
    std::vector<unsigned long long> h(trace_words);
    CK(hipMemcpy(h.data(), trace, trace_words * 8, hipMemcpyDeviceToHost));
    // phases of every interior tile (not a workgroup's first or last one), overlapped form, in 10 ns ticks
    const char *names[8] = {"claim + entry (previous tile's end -> ticket in hand)", "prologue issue (first stage's copies, next ticket)",
                            "prologue wait (first stage lands, barrier)", "k loop but the last step", "last step (+ blocks 0..2 recombined, stored)",
                            "last block's chain + stores issued", "whole tile", "shader clock [MHz]"};
    std::vector<double> ph[8], pro[3];
    size_t tiles = 0, plain_tiles = 0;
    for (int wg = 0; wg < 256; wg++) {
      const unsigned long long n = std::min<unsigned long long>(h[wg], 40);
      for (unsigned long long q = 1; q + 1 < n; q++) {
        const unsigned long long *r = &h[16384 + ((size_t)wg * 40 + q) * 16], *rp = r - 16;
        if (!(r[9] & 1)) {
          plain_tiles++;
          continue;
        }
        tiles++;
        ph[0].push_back((double)(r[0] - rp[5]));
        for (int i = 1; i <= 5; i++) ph[i].push_back((double)(r[i] - r[i - 1]));
        ph[6].push_back((double)(r[5] - rp[5]));
        pro[0].push_back((double)(r[10] - r[0]));  // set-up: pointers, lane offsets, phase
        pro[1].push_back((double)(r[11] - r[10])); // the first stage's copies issued
        pro[2].push_back((double)(r[1] - r[11]));  // the next ticket (hook)
        ph[7].push_back((double)(r[7] - r[6]) / ((double)(r[5] - r[0]) * 0.01));
      }
    }
    std::printf("interior tiles traced: %zu overlapped (+ %zu in the plain form, not tabulated); tiles per workgroup: %llu .. %llu\n", tiles, plain_tiles,
                *std::min_element(h.begin(), h.begin() + 256), *std::max_element(h.begin(), h.begin() + 256));
    double clock_mhz = 0;
    for (int i = 0; i < 8; i++) {
      if (ph[i].empty()) continue;
      std::sort(ph[i].begin(), ph[i].end());
      double mean = 0;
      for (double v : ph[i]) mean += v;
      mean /= (double)ph[i].size();
      const double sc = i == 7 ? 1.0 : 0.01; // ticks -> us
      if (i == 7) clock_mhz = ph[i][ph[i].size() / 2];
      std::printf("  %-56s median %8.2f  mean %8.2f  p10 %8.2f  p90 %8.2f %s\n", names[i], ph[i][ph[i].size() / 2] * sc, mean * sc,
                  ph[i][ph[i].size() / 10] * sc, ph[i][ph[i].size() * 9 / 10] * sc, i == 7 ? "" : "us");
    }
    const char *pn[3] = {"  prologue: set-up (pointers, offsets)", "  prologue: first stage's copies issued", "  prologue: next ticket drawn (atomic round trip)"};
    for (int i = 0; i < 3; i++) {
      if (pro[i].empty()) continue;
      std::sort(pro[i].begin(), pro[i].end());
      std::printf("  %-56s median %8.2f  p10 %8.2f  p90 %8.2f us\n", pn[i], pro[i][pro[i].size() / 2] * 0.01, pro[i][pro[i].size() / 10] * 0.01,
                  pro[i][pro[i].size() * 9 / 10] * 0.01);
    }
    const double steps = (double)(a.KB / 2), mfma_cycles = steps * 360.0 * 16.0;
    if (clock_mhz > 0)
      std::printf("  MFMA issue time of a tile at that clock: %.2f us (%g steps x 360 x 16 cycles), of its last step: %.2f us\n", mfma_cycles / clock_mhz,
                  steps, 360.0 * 16.0 / clock_mhz);
    // one workgroup's consecutive tiles, raw (us since its first stamp)
    for (int wg : {0, 100}) {
      const unsigned long long n = std::min<unsigned long long>(h[wg], 6);
      const unsigned long long t0 = h[16384 + (size_t)wg * 40 * 16];
      for (unsigned long long q = 0; q < n; q++) {
        const unsigned long long *r = &h[16384 + ((size_t)wg * 40 + q) * 16];
        std::printf("  wg %3d tile %llu (rb0 %llu tn %llu xcd %llu %s):", wg, q, r[8] >> 32, r[8] & 0xffffffffull, r[9] >> 8, r[9] & 1 ? "ovl" : "plain");
        for (int i = 0; i < 6; i++) std::printf(" %8.2f", (double)(r[i] - t0) * 0.01);
        std::printf("\n");
      }
    }
    return 0;
  }
#elif defined(ABLATE_SCHEDULE) // -DABLATE_SCHEDULE [-DABLATE_S=n -DABLATE_WA=w]: copy spacing / barrier position of the wide kernel's k-step
#ifndef ABLATE_WA
#define ABLATE_WA 3
#endif
  constexpr int W = ABLATE_WA;
  std::vector<Var> vars = {
      {"shipped: dma every 4, tail 6", run_w<S, W, 0, 0, -1, 4, 6, true, true>, false, {}},
      {"dma every 3, tail 6", run_w<S, W, 0, 0, -1, 3, 6, true, true>, false, {}},
      {"dma every 2, tail 6", run_w<S, W, 0, 0, -1, 2, 6, true, true>, false, {}},
      {"dma every 5, tail 6", run_w<S, W, 0, 0, -1, 5, 6, true, true>, false, {}},
      {"dma every 4, tail 4", run_w<S, W, 0, 0, -1, 4, 4, true, true>, false, {}},
      {"dma every 4, tail 8", run_w<S, W, 0, 0, -1, 4, 8, true, true>, false, {}},
      {"dma every 4, tail 5", run_w<S, W, 0, 0, -1, 4, 5, true, true>, false, {}},
      {"pd2, dma every 4, tail 6", run_w<S, W, VARW_NA3, 0, -1, 4, 6, true, true>, false, {}},
      {"barrier on every second k-step only (wrong results: what halving the barriers could return)",
       run_w<S, W, VARW_HALF_BARRIERS, 0, -1, 4, 6, true, true>, false, {}},
      {"one B buffer", run_w<S, W, VARW_B1, 0, -1, 4, 6, true, true>, false, {}},
      {"one B buffer, tail 4", run_w<S, W, VARW_B1, 0, -1, 4, 4, true, true>, false, {}},
      {"one B buffer, tail 3", run_w<S, W, VARW_B1, 0, -1, 4, 3, true, true>, false, {}},
      {"tail 3", run_w<S, W, 0, 0, -1, 4, 3, true, true>, false, {}},
      {"tail 2", run_w<S, W, 0, 0, -1, 4, 2, true, true>, false, {}},
      {"mfma only", run_w<S, W, VARW_MFMA_ONLY, 0, -1, 4, 6, true, true>, false, {}},
  };
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vars) {
      const float ms = v.fn(a, st, e0, e1);
      if (r > 0) v.ms.push_back(ms);
    }
  for (auto &v : vars) {
    std::sort(v.ms.begin(), v.ms.end());
    const double ops = (double)(S * (S + 1) / 2) * 2.0 * M * N * K;
    printf("S=%d WA=%d %s median %8.3f ms (%7.1f TOPS)   min %8.3f ms\n", S, W, v.name, v.ms[v.ms.size() / 2],
           ops / v.ms[v.ms.size() / 2] / 1e9, v.ms[0]);
  }
  return 0;
#elif defined(ABLATE_X16) // -DABLATE_X16 [-DABLATE_S=n -DABLATE_WA=w]: the paired 16x16x64 tile against the 32x32x32 one
#ifndef ABLATE_WA
#define ABLATE_WA 3
#endif
  constexpr int W = ABLATE_WA;
  constexpr int X = VARW_X16;
  std::vector<Var> vars = {
      {"32x32x32 shipped (dma every 4, tail 6)", run_w<S, W, 0, 0, -1, 4, 6, true, true>, false, {}},
      {"16x16x64 paired dma every 8, tail 12", run_w<S, W, X, 0, -1, 8, 12, true, true>, false, {}},
      {"16x16x64 paired dma every 6, tail 12", run_w<S, W, X, 0, -1, 6, 12, true, true>, false, {}},
      {"16x16x64 paired dma every 12, tail 12", run_w<S, W, X, 0, -1, 12, 12, true, true>, false, {}},
      {"16x16x64 paired dma every 16, tail 12", run_w<S, W, X, 0, -1, 16, 12, true, true>, false, {}},
      {"16x16x64 paired dma every 8, tail 8", run_w<S, W, X, 0, -1, 8, 8, true, true>, false, {}},
      {"16x16x64 paired dma every 8, tail 16", run_w<S, W, X, 0, -1, 8, 16, true, true>, false, {}},
      {"16x16x64 paired dma every 8, tail 20", run_w<S, W, X, 0, -1, 8, 20, true, true>, false, {}},
      {"16x16x64 k64 64x128, dma every 8, tail 12", run_w<S, 2, VARW_K64 | VARW_B1, 0, -1, 8, 12, false, true>, false, {}},
      {"16x16x64 k64 64x128, dma every 12, tail 12", run_w<S, 2, VARW_K64 | VARW_B1, 0, -1, 12, 12, false, true>, false, {}},
      {"16x16x64 k64 64x128, dma every 6, tail 16", run_w<S, 2, VARW_K64 | VARW_B1, 0, -1, 6, 16, false, true>, false, {}},
      {"16x16x64 k64 64x128, dma every 12, tail 24", run_w<S, 2, VARW_K64 | VARW_B1, 0, -1, 12, 24, false, true>, false, {}},
      {"16x16x64 k64 64x128 no copies", run_w<S, 2, VARW_K64 | VARW_B1 | VARW_NO_GLOBAL, 0, -1, 8, 12, false, true>, false, {}},
      {"16x16x64 k64 64x128 mfma only", run_w<S, 2, VARW_K64 | VARW_B1 | VARW_MFMA_ONLY, 0, -1, 8, 12, false, true>, false, {}},
      {"32x32x32 64x128 (WA = 2) shipped schedule", run_w<S, 2, 0, 0, -1, 4, 6, false, true>, false, {}},
      {"16x16x64 paired no-epilogue", run_w<S, W, X | VARW_NO_EPILOGUE, 0, -1, 8, 12, true, true>, false, {}},
      {"16x16x64 paired no copies", run_w<S, W, X | VARW_NO_GLOBAL, 0, -1, 8, 12, true, true>, false, {}},
      {"16x16x64 paired mfma only", run_w<S, W, X | VARW_MFMA_ONLY, 0, -1, 8, 12, true, true>, false, {}},
      {"32x32x32 mfma only", run_w<S, W, VARW_MFMA_ONLY, 0, -1, 4, 6, true, true>, false, {}},
  };
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vars) {
      const float ms = v.fn(a, st, e0, e1);
      if (r > 0) v.ms.push_back(ms);
    }
  { // bitwise cross-check of the paired tile against the 32x32x32 tile (interior fast path and, with odd M/N, the edges)
    std::vector<double> c0(M * N), c1(M * N);
    CK(hipMemset(C, 0xFF, 8 * M * N));
    run_w<S, W, 0, 0, -1, 4, 6, true, true>(a, st, e0, e1);
    CK(hipMemcpy(c0.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
    CK(hipMemset(C, 0xFF, 8 * M * N));
    run_w<S, W, X, 0, -1, 8, 12, true, true>(a, st, e0, e1);
    CK(hipMemcpy(c1.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
    size_t bad = 0, first = (size_t)-1;
    for (size_t i = 0; i < M * N; i++)
      if (std::memcmp(&c0[i], &c1[i], 8)) {
        if (first == (size_t)-1) first = i;
        bad++;
      }
    {
      CK(hipMemset(C, 0xFF, 8 * M * N));
      run_w<S, 2, VARW_K64 | VARW_B1, 0, -1, 8, 12, false, true>(a, st, e0, e1);
      std::vector<double> c2(M * N);
      CK(hipMemcpy(c2.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
      size_t bad2 = 0;
      for (size_t i = 0; i < M * N; i++) bad2 += std::memcmp(&c0[i], &c2[i], 8) != 0;
      std::printf("k64 16x16x64 tile vs 32x32x32 tile: %zu mismatching elements of %zu\n", bad2, M * N);
    }
    std::printf("paired 16x16x64 tile vs 32x32x32 tile: %zu mismatching elements of %zu", bad, M * N);
    if (bad) std::printf(" (first at m=%zu n=%zu: %a vs %a)", first % M, first / M, c0[first], c1[first]);
    std::printf("\n");
  }
  for (int which = 0; which < 2; which++) { // cycle stamps (VARW_TRACE) of both tile functions: where a k-step spends its time
    uint32_t *tr;
    const size_t ntr = 32 * 4 * 8 * 8;
    const size_t tr_bytes = 4096 * 8 + 16384 * 3 * 8;
    CK(hipMalloc(&tr, tr_bytes));
    CK(hipMemset(tr, 0, tr_bytes));
    SliceGemmArgs b = a;
    b.acc = reinterpret_cast<double *>(tr);
    if (which == 0) run_w<S, W, VARW_TRACE, 0, -1, 4, 6, true, true>(b, st, e0, e1);
    else run_w<S, W, X | VARW_TRACE, 0, -1, 8, 12, true, true>(b, st, e0, e1);
    std::vector<uint32_t> h(ntr);
    CK(hipMemcpy(h.data(), tr, ntr * 4, hipMemcpyDeviceToHost));
    auto d = [](uint32_t x, uint32_t y) { return (double)(uint32_t)(y - x); };
    double sum[8] = {0};
    int cnt = 0;
    for (int bw = 0; bw < 32 * 4; bw++)
      for (int stp = 0; stp + 1 < 8; stp++) {
        const uint32_t *t = &h[(bw * 8 + stp) * 8], *tn = t + 8;
        if (!t[0] || !tn[0]) continue;
        sum[0] += d(t[5], t[0]); sum[1] += d(t[0], t[1]); sum[2] += d(t[1], t[2]); sum[3] += d(t[2], t[3]);
        sum[4] += d(t[3], t[4]); sum[5] += d(t[5], tn[5]); sum[6] += d(t[6], t[7]);
        cnt++;
      }
    if (cnt)
      std::printf("%s trace (%d samples, shader cycles): step %.0f = start->X %.0f + vmcnt %.0f + lgkmcnt %.0f + barrier %.0f "
                  "+ X->end %.0f ; one copy issue %.0f\n", which ? "16x16x64 paired" : "32x32x32",
                  cnt, sum[5] / cnt, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[6] / cnt);
    CK(hipFree(tr));
  }
  for (auto &v : vars) {
    std::sort(v.ms.begin(), v.ms.end());
    const double ops = (double)(S * (S + 1) / 2) * 2.0 * M * N * K;
    printf("S=%d WA=%d %-40s median %8.3f ms (%7.1f TOPS)   min %8.3f ms\n", S, W, v.name, v.ms[v.ms.size() / 2],
           ops / v.ms[v.ms.size() / 2] / 1e9, v.ms[0]);
  }
  return 0;
#elif defined(ABLATE_K64) // -DABLATE_K64 [-DABLATE_S=n -DOZ_Y_RING=r]: schedule parameters of the k64 tile (64-k steps, 16x16x64)
  constexpr int Y = VARW_K64 | VARW_B1;
  std::vector<Var> vars = {
      {"32x32x32 96x128 shipped", run_w<S, 3, 0, 0, -1, 4, 6, true, true>, false, {}},
      {"k64 64x128 dma every 8, tail 12", run_w<S, 2, Y, 0, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 dma every 4, tail 12", run_w<S, 2, Y, 0, -1, 4, 12, false, true>, false, {}},
      {"k64 64x128 dma every 6, tail 12", run_w<S, 2, Y, 0, -1, 6, 12, false, true>, false, {}},
      {"k64 64x128 dma every 10, tail 12", run_w<S, 2, Y, 0, -1, 10, 12, false, true>, false, {}},
      {"k64 64x128 dma every 8, tail 8", run_w<S, 2, Y, 0, -1, 8, 8, false, true>, false, {}},
      {"k64 64x128 dma every 8, tail 6", run_w<S, 2, Y, 0, -1, 8, 6, false, true>, false, {}},
      {"k64 64x128 dma every 8, tail 16", run_w<S, 2, Y, 0, -1, 8, 16, false, true>, false, {}},
      {"k64 64x128 dma every 8 from slot 2, tail 12", run_w<S, 2, Y, 0, 2, 8, 12, false, true>, false, {}},
      {"k64 64x128 dma every 8, tail 12, static grid", run_w<S, 2, Y, 0, -1, 8, 12, false, false>, false, {}},
      {"k64 64x128 no epilogue", run_w<S, 2, Y | VARW_NO_EPILOGUE, 0, -1, 8, 12, false, true>, false, {}},
  };
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vars) {
      const float ms = v.fn(a, st, e0, e1);
      if (r > 0) v.ms.push_back(ms);
    }
  for (auto &v : vars) {
    std::sort(v.ms.begin(), v.ms.end());
    const double ops = (double)(S * (S + 1) / 2) * 2.0 * M * N * K;
    printf("S=%d ring %d %-46s median %8.3f ms (%7.1f TOPS)   min %8.3f ms\n", S, OZ_Y_RING, v.name, v.ms[v.ms.size() / 2],
           ops / v.ms[v.ms.size() / 2] / 1e9, v.ms[0]);
  }
  return 0;
#elif defined(ABLATE_BREG) // -DABLATE_BREG [-DABLATE_S=n]: k64 tile with the B fragments loaded global -> VGPR (VARW_BREG) against the LDS-staged form
  constexpr int Y = VARW_K64 | VARW_B1, Z = VARW_K64 | VARW_BREG;
#ifndef ABLATE_BREG_PART
#define ABLATE_BREG_PART 1
#endif
  std::vector<Var> vars = {
      {"k64 64x128 B via LDS (round 3) dma 8 tail 12", run_w<S, 2, Y, 0, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR dma 8 tail 12", run_w<S, 2, Z, 0, -1, 8, 12, false, true>, false, {}},
#if ABLATE_BREG_PART == 1 // (a kernel instantiation compiles for a minute: two binaries, built side by side)
      {"k64 64x128 B->VGPR dma 8 tail 6", run_w<S, 2, Z, 0, -1, 8, 6, false, true>, false, {}},
      {"k64 64x128 B->VGPR dma 8 tail 20", run_w<S, 2, Z, 0, -1, 8, 20, false, true>, false, {}},
      {"k64 64x128 B->VGPR no copies", run_w<S, 2, Z | VARW_NO_GLOBAL, 0, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 B via LDS no copies", run_w<S, 2, Y | VARW_NO_GLOBAL, 0, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 mfma only", run_w<S, 2, Y | VARW_MFMA_ONLY, 0, -1, 8, 12, false, true>, false, {}},
#elif ABLATE_BREG_PART == 5 // -DABLATE_S=8 (or 7): the 64x128 B-in-registers form against the 96x128 LDS form of fewer slices
      {"k64 96x128 B via LDS (shipped for S = 7, 8)", run_w<S, 3, Y, 0, -1, 8, 12, true, true>, false, {}},
      {"k64 64x128 B via LDS", run_w<S, 2, Y, 0, -1, 8, 12, false, true>, false, {}},
#elif ABLATE_BREG_PART == 4 // tile boundaries (run with a short K: tools/bin/gemm_ablate_breg4 8192 7 127 8192 512)
      {"k64 64x128 B->VGPR persistent, claim one tile ahead", run_w<S, 2, Z, 0, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR static grid (no claims)", run_w<S, 2, Z, 0, -1, 8, 12, false, false>, false, {}},
      {"k64 64x128 B->VGPR persistent, no epilogue", run_w<S, 2, Z | VARW_NO_EPILOGUE, 0, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR static, no epilogue", run_w<S, 2, Z | VARW_NO_EPILOGUE, 0, -1, 8, 12, false, false>, false, {}},
      {"k64 64x128 B->VGPR persistent, epilogue w/o stores", run_w<S, 2, Z | VARW_EPI_NOSTORE, 0, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 mfma only (persistent)", run_w<S, 2, Y | VARW_MFMA_ONLY, 0, -1, 8, 12, false, true>, false, {}},
#elif ABLATE_BREG_PART == 6 // small problems (run: tools/bin/gemm_ablate_breg6 1024 9): what bounds the half-height (32 x 128) tile's k loop
      {"k64 32x128 B->VGPR static grid", run_w<S, 1, Z, 0, -1, 8, 12, false, false>, false, {}},
      {"k64 32x128 B->VGPR static grid, no copies", run_w<S, 1, Z | VARW_NO_GLOBAL, 0, -1, 8, 12, false, false>, false, {}},
      {"k64 32x128 B->VGPR static grid, no epilogue", run_w<S, 1, Z | VARW_NO_EPILOGUE, 0, -1, 8, 12, false, false>, false, {}},
      {"k64 32x128 mfma only", run_w<S, 1, Y | VARW_MFMA_ONLY, 0, -1, 8, 12, false, false>, false, {}},
      {"k64 64x128 B->VGPR static grid, no copies", run_w<S, 2, Z | VARW_NO_GLOBAL, 0, -1, 8, 12, false, false>, false, {}},
#elif ABLATE_BREG_PART == 3 // what the step's waits cost (wrong results)
      {"k64 64x128 B->VGPR without the vmcnt wait", run_w<S, 2, Z, 100, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR without the lgkmcnt wait", run_w<S, 2, Z, 200, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR without the barrier", run_w<S, 2, Z, 400, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR without all three", run_w<S, 2, Z, 700, -1, 8, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR no copies, without all three", run_w<S, 2, Z | VARW_NO_GLOBAL, 700, -1, 8, 12, false, true>, false, {}},
#else
      {"k64 64x128 B->VGPR dma 4 tail 12", run_w<S, 2, Z, 0, -1, 4, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR dma 6 tail 12", run_w<S, 2, Z, 0, -1, 6, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR dma 10 tail 12", run_w<S, 2, Z, 0, -1, 10, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR dma 12 tail 12", run_w<S, 2, Z, 0, -1, 12, 12, false, true>, false, {}},
      {"k64 64x128 B->VGPR dma 8 from slot 2 tail 12", run_w<S, 2, Z, 0, 2, 8, 12, false, true>, false, {}},
#endif
  };
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vars) {
      const float ms = v.fn(a, st, e0, e1);
      if (r > 0) v.ms.push_back(ms);
    }
  { // bitwise cross-check against the LDS-staged form
    std::vector<double> c0(M * N), c1(M * N);
    CK(hipMemset(C, 0xFF, 8 * M * N));
    run_w<S, 2, Y, 0, -1, 8, 12, false, true>(a, st, e0, e1);
    CK(hipMemcpy(c0.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
    CK(hipMemset(C, 0xFF, 8 * M * N));
    run_w<S, 2, Z, 0, -1, 8, 12, false, true>(a, st, e0, e1);
    CK(hipMemcpy(c1.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < M * N; i++) bad += std::memcmp(&c0[i], &c1[i], 8) != 0;
    std::printf("k64 B->VGPR vs k64 B via LDS: %zu mismatching elements of %zu\n", bad, M * N);
  }
  for (auto &v : vars) {
    std::sort(v.ms.begin(), v.ms.end());
    const double ops = (double)(S * (S + 1) / 2) * 2.0 * M * N * K;
    printf("S=%d %-48s median %8.3f ms (%7.1f TOPS)   min %8.3f ms\n", S, v.name, v.ms[v.ms.size() / 2],
           ops / v.ms[v.ms.size() / 2] / 1e9, v.ms[0]);
  }
  return 0;
#elif defined(ABLATE_SMALL_TILES) // -DABLATE_SMALL_TILES: the 64x128 / 32x128 forms of the wide kernel (problems of ~2 tiles per CU)
  std::vector<Var> vars = {
      {"wide 64x128 persistent dma every 4 (shipped)", run_w<S, 2, 0, 0, -1, 4, 6, false, true>, false, {}},
      {"wide 64x128 persistent dma every 3", run_w<S, 2, 0, 0, -1, 3, 6, false, true>, false, {}},
      {"wide 64x128 persistent dma every 2", run_w<S, 2, 0, 0, -1, 2, 6, false, true>, false, {}},
      {"wide 64x128 persistent dma every 1", run_w<S, 2, 0, 0, -1, 1, 6, false, true>, false, {}},
      {"wide 64x128 persistent dma every 2 tail 10", run_w<S, 2, 0, 0, -1, 2, 10, false, true>, false, {}},
      {"wide 64x128 persistent dma every 2 from slot 1", run_w<S, 2, 0, 0, 1, 2, 6, false, true>, false, {}},
      {"wide 64x128 persistent pd2 dma every 4", run_w<S, 2, VARW_NA3, 0, -1, 4, 6, false, true>, false, {}},
      {"wide 64x128 persistent pd2 dma every 2", run_w<S, 2, VARW_NA3, 0, -1, 2, 6, false, true>, false, {}},
      {"wide 64x128 persistent no-global", run_w<S, 2, VARW_NO_GLOBAL, 0, -1, 4, 6, false, true>, false, {}},
      {"wide 64x128 persistent mfma-only", run_w<S, 2, VARW_MFMA_ONLY, 0, -1, 4, 6, false, true>, false, {}},
      {"wide 96x128 persistent (shipped)", run_w<S, 3, 0, 0, -1, 4, 6, false, true>, false, {}},
      {"wide 32x128 static", run_w<S, 1, 0, 0, -1, 4, 6, false, false>, false, {}},
      {"wide 32x128 persistent", run_w<S, 1, 0, 0, -1, 4, 6, false, true>, false, {}},
      {"wide 32x128 static tail 10", run_w<S, 1, 0, 0, -1, 4, 10, false, false>, false, {}},
      {"wide 32x128 static pd2", run_w<S, 1, VARW_NA3, 0, -1, 4, 6, false, false>, false, {}},
      {"wide 32x128 static mfma-only", run_w<S, 1, VARW_MFMA_ONLY, 0, -1, 4, 6, false, false>, false, {}},
      {"wide 64x128 static", run_w<S, 2, 0, 0, -1, 4, 6, false, false>, false, {}},
      {"classic 64x64", run<S, VAR_SHIPPED>, false, {}},
  };
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vars) {
      const float ms = v.fn(a, st, e0, e1);
      if (r > 0) v.ms.push_back(ms);
    }
  { // bitwise cross-check of the candidates against the plain loop
    std::vector<double> c0(M * N), c1(M * N);
    run<S, 0>(a, st, e0, e1);
    CK(hipMemcpy(c0.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
    for (int which = 0; which < 3; which++) {
      CK(hipMemset(C, 0xFF, 8 * M * N));
      if (which == 0) run_w<S, 2, 0, 0, -1, 2, 6, false, true>(a, st, e0, e1);
      if (which == 1) run_w<S, 2, 0, 0, -1, 1, 6, false, true>(a, st, e0, e1);
      if (which == 2) run_w<S, 1, 0, 0, -1, 4, 6, false, false>(a, st, e0, e1);
      CK(hipMemcpy(c1.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
      printf("candidate %d vs plain loop: %s\n", which, std::memcmp(c0.data(), c1.data(), 8 * M * N) ? "MISMATCH" : "bitwise equal");
    }
  }
  for (auto &v : vars) {
    std::sort(v.ms.begin(), v.ms.end());
    const double ops = 45.0 * 2.0 * M * N * K;
    printf("%s median %8.3f ms (%7.1f TOPS)   min %8.3f ms (%7.1f TOPS)\n", v.name, v.ms[v.ms.size() / 2],
           ops / v.ms[v.ms.size() / 2] / 1e9, v.ms[0], ops / v.ms[0] / 1e9);
  }
  return 0;
#else
  std::vector<Var> vars = {
      {"shipped 64x64 + throttle", run_throttled<S, VAR_SHIPPED>, false, {}},
      {"wide 96x128 pd2", run_w<S, 3, VARW_NA3>, false, {}},
      {"wide 96x128 pd2 mixed heights", run_w<S, 3, VARW_NA3, 0, -1, 4, 6, true>, false, {}},
            {"wide 96x128 pd1 band 4x8 mixed", run_w<S, 3, VARW_BAND4, 0, -1, 4, 6, true>, false, {}},
      {"wide 96x128 pd1", run_w<S, 3, 0>, false, {}},
      {"wide 96x128 pd1 mixed heights", run_w<S, 3, 0, 0, -1, 4, 6, true>, false, {}},
      {"wide 96x128 pd1 mixed persistent+steal", run_w<S, 3, 0, 0, -1, 4, 6, true, true>, false, {}},
      {"wide 96x128 pd1 persistent+steal", run_w<S, 3, 0, 0, -1, 4, 6, false, true>, false, {}},
      {"wide pd1 mixed persistent no-epilogue", run_w<S, 3, VARW_NO_EPILOGUE, 0, -1, 4, 6, true, true>, false, {}},
      {"wide pd1 mixed persistent epilogue w/o stores", run_w<S, 3, VARW_EPI_NOSTORE, 0, -1, 4, 6, true, true>, false, {}},
      {"wide pd1 mixed persistent epilogue w/o chains", run_w<S, 3, VARW_EPI_NOCHAIN, 0, -1, 4, 6, true, true>, false, {}},
      {"wide 96x128 pd1 mixed dma every 6", run_w<S, 3, 0, 0, -1, 6, 6, true>, false, {}},
      {"wide 96x128 pd1 mixed tail 10", run_w<S, 3, 0, 0, -1, 4, 10, true>, false, {}},
      {"wide 96x128 no-global", run_w<S, 3, VARW_NA3 | VARW_NO_GLOBAL>, false, {}},
      {"wide 96x128 mfma-only rand", run_w<S, 3, VARW_NA3 | VARW_MFMA_ONLY>, false, {}},
      {"64x64 mfma-only random regs", run<S, VAR_MFMA_ONLY | VAR_RAND_REGS>, false, {}},
  };
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vars) {
      const float ms = v.fn(a, st, e0, e1);
      if (r > 0) v.ms.push_back(ms);
    }
  { // correctness of the candidate kernels against the plain loop (bitwise on C)
    std::vector<double> c0(M * N), c1(M * N);
    run<S, 0>(a, st, e0, e1);
    CK(hipMemcpy(c0.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
    for (int which = 0; which < 5; which++) {
      CK(hipMemset(C, 0xFF, 8 * M * N));
      if (which == 0) run_throttled<S, VAR_SHIPPED>(a, st, e0, e1);
      if (which == 1) run_w<S, 3, VARW_NA3>(a, st, e0, e1);
      if (which == 2) run_w<S, 3, 0>(a, st, e0, e1);
      if (which == 3) run_w<S, 3, VARW_NA3, 0, -1, 4, 6, true>(a, st, e0, e1);
      if (which == 4) run_w<S, 3, 0, 0, -1, 4, 6, true, true>(a, st, e0, e1);
      CK(hipMemcpy(c1.data(), C, 8 * M * N, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < M * N; i++) bad += c0[i] != c1[i];
      std::printf("check variant %d vs plain loop: %zu mismatching elements of %zu\n", which, bad, M * N);
    }
  }
  { // cycle stamps of the wide kernel (VARW_TRACE): where a k-step spends its time
    uint32_t *tr;
    const size_t ntr = 32 * 4 * 8 * 8;
    const size_t nwg_max = 16384;
    const size_t tr_bytes = 4096 * 8 + nwg_max * 3 * 8;
    CK(hipMalloc(&tr, tr_bytes));
    CK(hipMemset(tr, 0, tr_bytes));
    SliceGemmArgs b = a;
    b.acc = reinterpret_cast<double *>(tr);
    run_w<S, 3, VARW_TRACE, 0, -1, 4, 6, true, true>(b, st, e0, e1);
    std::vector<uint32_t> h(ntr);
    CK(hipMemcpy(h.data(), tr, ntr * 4, hipMemcpyDeviceToHost));
    { // per-CU timeline: when does each CU run dry, how long are the gaps between its workgroups
      std::vector<unsigned long long> w(nwg_max * 3);
      CK(hipMemcpy(w.data(), reinterpret_cast<unsigned long long *>(tr) + 4096, nwg_max * 24, hipMemcpyDeviceToHost));
      std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
      unsigned long long t_begin = ~0ull, t_end = 0;
      size_t nwg = 0;
      for (size_t i = 0; i < nwg_max; i++)
        if (w[3 * i + 2]) {
          cu[w[3 * i]].push_back({w[3 * i + 1], w[3 * i + 2]});
          t_begin = std::min(t_begin, w[3 * i + 1]);
          t_end = std::max(t_end, w[3 * i + 2]);
          nwg++;
        }
      double busy = 0, gaps = 0, tail = 0, head = 0;
      for (auto &kv : cu) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end());
        head += (double)(v.front().first - t_begin);
        tail += (double)(t_end - v.back().second);
        for (size_t i = 0; i < v.size(); i++) {
          busy += (double)(v[i].second - v[i].first);
          if (i) gaps += (double)(v[i].first - v[i - 1].second);
        }
      }
      { // does the tail come from whole XCDs finishing early (static partition of the grid) or from single CUs?
        std::map<unsigned, unsigned long long> xcd_end, xcd_first_idle;
        for (auto &kv : cu) {
          const unsigned x = (unsigned)(kv.first >> 32);
          xcd_end[x] = std::max(xcd_end[x], kv.second.back().second);
          if (!xcd_first_idle.count(x)) xcd_first_idle[x] = kv.second.back().second;
          xcd_first_idle[x] = std::min(xcd_first_idle[x], kv.second.back().second);
        }
        std::printf("wide timeline per XCD (ms before kernel end: last CU done / first CU idle):");
        for (auto &kv : xcd_end)
          std::printf("  x%u %.3f/%.3f", kv.first, (double)(t_end - kv.second) / 1e5, (double)(t_end - xcd_first_idle[kv.first]) / 1e5);
        std::printf("\n");
      }
      const double span = (double)(t_end - t_begin) * cu.size();
      std::printf("wide timeline: %zu workgroups on %zu CUs, kernel span %.3f ms; CU time: busy %.2f %%, idle before first "
                  "workgroup %.2f %%, gaps between workgroups %.2f %%, idle after last workgroup (tail) %.2f %%\n",
                  nwg, cu.size(), (double)(t_end - t_begin) / 1e5, 100 * busy / span, 100 * head / span, 100 * gaps / span,
                  100 * tail / span);
    }
    auto d = [](uint32_t x, uint32_t y) { return (double)(uint32_t)(y - x); };
    double sum[8] = {0};
    int cnt = 0;
    for (int bw = 0; bw < 32 * 4; bw++)
      for (int stp = 0; stp + 1 < 8; stp++) {
        const uint32_t *t = &h[(bw * 8 + stp) * 8], *tn = t + 8;
        if (!t[0] || !tn[0]) continue;
        sum[0] += d(t[5], t[0]);   // step start -> X
        sum[1] += d(t[0], t[1]);   // vmcnt wait
        sum[2] += d(t[1], t[2]);   // lgkmcnt(0)
        sum[3] += d(t[2], t[3]);   // barrier
        sum[4] += d(t[3], t[4]);   // X -> step end (refresh reads + TAIL MFMAs)
        sum[5] += d(t[5], tn[5]);  // whole step
        sum[6] += d(t[6], t[7]);   // one copy (M0 + issue)
        cnt++;
      }
    if (cnt)
      std::printf("wide trace (%d samples, shader cycles): step %.0f = start->X %.0f + vmcnt %.0f + lgkmcnt %.0f + barrier %.0f "
                  "+ X->end %.0f ; one copy issue %.0f\n",
                  cnt, sum[5] / cnt, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[6] / cnt);
    else
      std::printf("wide trace: no samples \n");
    CK(hipFree(tr));
  }
  const double ops = 45.0 * 2.0 * M * N * K;
  std::printf("M=%zu N=%zu K=%zu S=%d rounds=%d  (TOPS = 45*2*MNK / t)\n", M, N, K, S, rounds);
  for (auto &v : vars) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms.front();
    std::printf("%-28s median %8.3f ms (%7.1f TOPS)   min %8.3f ms (%7.1f TOPS)\n", v.name, med, ops / med / 1e9, mn,
                ops / mn / 1e9);
  }
  return 0;
#endif
}
