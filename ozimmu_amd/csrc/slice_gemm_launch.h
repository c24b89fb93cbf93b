// slice_gemm_launch.h — host side of the slice GEMM: kernel choice, tile plan and launches of one compute mode S
// (design notes: slice_gemm.hip).  Included by the slice_gemm_s*.hip translation units, each of which instantiates the
// kernels of a range of S (OZ_S_LO .. OZ_S_HI) behind one entry point OZ_PART(S, args, stream): the ~60 kernel
// instantiations compile in parallel instead of in one five-minute translation unit.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "config.h"
#include "topology.h"
#include "tile_plan.h"
#include "kernel_policy.h"
#include "slice_gemm_k2_kernel.h"
#include "slice_gemm_w_kernel.h"

namespace ozhip {


// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: one bit per device id and kernel
// instantiation (a process may drive several GPUs through one copy of this library).
// (`dev`: SliceGemmArgs::device, the handle's device - current for the duration of the call, api.cpp: WorkspaceUse)
template <class K>
static hipError_t allow_dynamic_lds(K kernel, size_t bytes, std::atomic<uint64_t> &done, int dev) {
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0ull;
  if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
  return e;
}

static int cu_count(const SliceGemmArgs &a) { return topology(a.device).cus; } // CUs of the launch's device (probed once: topology.h)

// ---- classic kernel: 64x64 (or 128x64) workgroups, two waves per SIMD ----------------------------------------------
template <int S, int D0, int ND, int FORCE_WM = 0>
static hipError_t launch_one(const SliceGemmArgs &a0, hipStream_t stream) {
  constexpr int SL = (D0 + ND < S) ? (D0 + ND) : S;
  // Workgroup shape (FORCE_WM: the caller's choice, see launch_S).  Up to 10 staged slices: 4 waves / 64x64, two workgroups per CU (they cover each other's
  // LDS-read phases).  11..13 staged slices do not fit twice in 160 KiB of LDS: one 8-wave 128x64 workgroup per CU
  // (2*(4+2)*SL KiB).  More than 13 (second pass of S >= 14): back to 64x64, one workgroup per CU.
  constexpr int WM = FORCE_WM ? FORCE_WM : ((SL >= 11 && SL <= 13) ? 4 : 2);
  constexpr size_t lds = 2 * (WM + 2) * SL * FRAG_BYTES;
  // the prefetch-2 loop keeps all 2*SL fragments in registers next to the 16*ND accumulators: beyond
  // ~232 of the 256 VGPRs (2 waves/SIMD) it would spill inside the k loop, and with one 8-wave workgroup per CU
  // its LDS read burst is exposed (tools/gemm_ablate.hip: 24.3 vs 19.6 ms): those cases stream the A fragments
  // (prefetch distance 1).
  constexpr int VAR_PRODUCTION = (WM == 2 && 16 * ND + 8 * SL <= 240) ? VAR_SHIPPED : (VAR_SHIPPED & ~VAR_PF2);
  SliceGemmArgs a = a0;
  a.tiles_m = (a.M + 32 * WM - 1) / (32 * WM);
  a.tiles_n = (a.N + 63) / 64;
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = allow_dynamic_lds(slice_gemm_kernel<S, D0, ND, VAR_PRODUCTION, WM>, lds, attr_done, a.device)) return e;
  const uint32_t nb = a.tiles_m * a.tiles_n;
  hipLaunchKernelGGL((slice_gemm_kernel<S, D0, ND, VAR_PRODUCTION, WM>), dim3(nb, a.batch > 1 ? a.batch : 1),
                     dim3(128 * WM), lds, stream, a);
  return hipGetLastError();
}

// ---- K-split kernel: one 8-wave workgroup per CU, 64x64 tiles, for launches with no more tiles than CUs --------------
template <int S, int D0, int ND>
static hipError_t launch_k2(const SliceGemmArgs &a0, hipStream_t stream) {
  using Cfg = K2Cfg<S, D0, ND>;
  SliceGemmArgs a = a0;
  a.tiles_m = (a.M + 63) / 64;
  a.tiles_n = (a.N + 63) / 64;
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = allow_dynamic_lds(slice_gemm_k2_kernel<S, D0, ND>, Cfg::LDS, attr_done, a.device)) return e;
  hipLaunchKernelGGL((slice_gemm_k2_kernel<S, D0, ND>), dim3(a.tiles_m * a.tiles_n, a.batch > 1 ? a.batch : 1), dim3(512),
                     Cfg::LDS, stream, a);
  return hipGetLastError();
}

template <int S>
static hipError_t launch_k2_fused(const SliceGemmArgs *g, int count, hipStream_t stream) {
  if constexpr (S > SINGLE_PASS_MAX_S || !K2Cfg<S, 0, S>::ok) {
    return hipErrorNotSupported;
  } else {
    using Cfg = K2Cfg<S, 0, S>;
    const SliceGemmArgs &a0 = g[0];
    SliceGemmMulti m{};
    m.count = count;
    for (int i = 0; i < count; i++) {
      m.g[i] = g[i];
      m.g[i].tiles_m = (g[i].M + 63) / 64;
      m.g[i].tiles_n = (g[i].N + 63) / 64;
    }
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t err = allow_dynamic_lds(slice_gemm_k2_fused_kernel<S, 0, S>, Cfg::LDS, attr_done, a0.device)) return err;
    hipLaunchKernelGGL((slice_gemm_k2_fused_kernel<S, 0, S>), dim3(m.g[0].tiles_m * m.g[0].tiles_n, a0.batch > 1 ? a0.batch : 1),
                       dim3(512), Cfg::LDS, stream, m);
    return hipGetLastError();
  }
}

// ---- wide kernel: one 4-wave workgroup per CU, (32*WA) x 128 tiles (slice_gemm_w_kernel.h) ------------------------
// WA: as many blocks per wave as the 512-entry register file holds next to the B fragments, the ring and addressing;
// NA: 2 A buffers = prefetch distance 1.  Distance 2 (3 buffers) is never faster and up to 6 % slower (profiles/
// r2_ablate: 17.66 vs 18.74 ms on the slowest box, equal on the fastest): a k-step of this kernel lasts ~2.8 us, enough
// for a copy to land, and prefetching two steps ahead widens the k window the XCD's workgroups keep alive in L2.
// (Which tile function runs where - 32x32x32, paired 16x16x64, k64 with B through LDS or in registers - is decided by the cost
// model of kernel_policy.cpp from the measured times of all of them.)
template <int S, int D0, int ND>
struct WideCfg {
  static constexpr int SL = (D0 + ND < S) ? (D0 + ND) : S;
  static constexpr bool regs_ok(int wa) { return wa * ND * 16 + SL * 4 + 16 + 14 <= 512; }
  // nb: wave-private B buffers (2, or 1 with VARW_B1: slice_gemm_w_kernel.h)
  static constexpr size_t lds(int wa, int na, int nb = 2) { return (size_t)(na * wa + 4 * nb) * SL * FRAG_BYTES; }
  static constexpr size_t LDS_MAX = 160 * 1024;
  static constexpr int pick(int nb) {
    return (regs_ok(4) && lds(4, 2, nb) <= LDS_MAX)   ? 4
           : (regs_ok(3) && lds(3, 2, nb) <= LDS_MAX) ? 3
           : (regs_ok(2) && lds(2, 2, nb) <= LDS_MAX) ? 2
           : (regs_ok(1) && lds(1, 2, nb) <= LDS_MAX) ? 1
                                                      : 0;
  }
  // two B buffers wherever they leave room for tiles of 64+ rows (every single pass, every first pass); the second pass
  // of S = 14..18 stages 14-18 slices: one B buffer there (64x128 tiles instead of 32x128 / the classic kernel)
  static constexpr int NB = pick(2) >= 2 ? 2 : (pick(1) > pick(2) ? 1 : 2);
  static constexpr int WA = pick(NB);
  static constexpr int NA = 2;
  static constexpr bool ok = WA >= 1;
};

// Paired tile (slice_gemm_x_tile.h): the same (32*WA) x 128 tile computed with v_mfma_i32_16x16x64_i8, two slice products
// of a diagonal per instruction.  Registers: the same 16*WA*ND accumulator registers, 2*NQ B pair fragments, the A ring.
template <int S, int D0, int ND>
struct PairedCfg {
  using W = WideCfg<S, D0, ND>;
  static constexpr int SL = W::SL;
  static constexpr int NQ = (SL - 1) / 2 + 1;
  static constexpr int RING = 4;
  // 432 accumulator registers (ND = 9, WA = 3) leave too little for the epilogue's FP64 chains: it spills 67 registers
  // (no loss in the k loop, but that configuration is 3 % slower than the 32x32x32 tile anyway): not instantiated
  static constexpr bool regs_ok(int wa) { return wa * ND * 16 + 2 * NQ * 4 + RING * 4 + 4 + 16 <= 500; }
  static constexpr size_t lds(int wa) { return W::lds(wa, 2, W::NB) + 2 * X_PAD; }
  static constexpr int pick() {
    return (regs_ok(4) && lds(4) <= W::LDS_MAX) ? 4 : (regs_ok(3) && lds(3) <= W::LDS_MAX) ? 3
         : (regs_ok(2) && lds(2) <= W::LDS_MAX) ? 2 : (regs_ok(1) && lds(1) <= W::LDS_MAX) ? 1 : 0;
  }
  static constexpr int WA = pick();
  // only where it keeps the tile height of the 32x32x32 form (a smaller tile gives the instruction's gain back)
  static constexpr bool ok = WA >= 1 && WA == W::WA && W::NA == 2;
  // copies of the next stage every DMAE-th MFMA slot, barrier TAIL slots before the end of a k-step (16-cycle slots)
  static constexpr int DMAE = 8, TAIL = 12;
};

// grid of a wide-kernel launch and its claim counters: persistent workgroups with per-XCD tile queues + stealing when there
// is a zeroed counter pair for this launch (single products only: the phase lines are per call) and more tiles than CUs;
// OZIMMU_HIP_WIDE_STATIC=1: A/B.  Fills a.tiles_*, a.rba, a.queue; returns the number of workgroups.
static uint32_t wide_grid(SliceGemmArgs &a, const WidePlan &pl) {
  a.tiles_m = pl.n_big;
  a.tiles_m2 = pl.n_small;
  a.tiles_n = (a.N + 127) / 128;
  a.rba = (uint32_t)row_blocks_padded(a.M);
  uint32_t nb = (a.tiles_m + a.tiles_m2) * a.tiles_n;
  a.queue = nullptr;
  const uint32_t max_slots = (PHASE_LINE_WORDS - 16) / 2; // words 16 .. 63 of a phase line
  // A few EXACT rounds of tiles (2048^3: 512 tiles of 64 x 128, 4096^3: 2048): nothing to balance, and the static grid - the
  // dispatcher hands the workgroups out in order - saves every tile the round trips of its claim: +1.1 ... 1.4 % at 2 tiles
  // per CU, +0.5 % at 8 and 18; from 32 per CU on the persistent workgroups win (8192^3 +0.7 %, 16384^2 x 1024 +2.2 %: the
  // claimed order keeps an XCD's workgroups on neighbouring panels).  profiles/r4_ablate/r4r_static_grid_vs_queue_ab.txt
  const uint32_t cus = (uint32_t)cu_count(a);
  const bool exact_few_rounds = nb % cus == 0 && nb / cus <= (uint32_t)std::max(0, config().static_rounds) && config().wide_grid == 0;
  if (a.phase && a.batch <= 1 && a.qslot < max_slots && nb > cus && !config().wide_static && !exact_few_rounds) {
    a.queue = a.phase + 16 + 2 * a.qslot;
    nb = (uint32_t)cu_count(a);
    if (config().wide_grid > 0) nb = (uint32_t)config().wide_grid; // tests: few workgroups, many tiles each
  } else if (config().wide_grid > 0 && a.phase && a.batch <= 1 && a.qslot < max_slots) {
    a.queue = a.phase + 16 + 2 * a.qslot;
    nb = std::min<uint32_t>(nb, (uint32_t)config().wide_grid);
  }
  return nb;
}

// k64 tile (slice_gemm_y_tile.h): two k-blocks per step, one slice product over 64 k per 16x16x64 instruction.  LDS: two A
// stages and one (two where they fit) wave-private B stage of 2 * S KiB per row-block; registers: 16 * WA * S accumulator
// registers + 2 * S B fragments.  Single diagonal pass only.
template <int S> // S here: the staged slices = diagonals of the pass (a single pass: the mode's S; a first pass: its ND)
struct K64Cfg {
  static constexpr size_t LDS_MAX = 160 * 1024;
  static constexpr size_t lds(int wa, int nb) { return (size_t)(2 * wa + 4 * nb) * (2 * S) * FRAG_BYTES; }
  static constexpr bool regs_ok(int wa) { return wa * S * 16 + 2 * S * 4 + 4 * 4 + 4 + 16 <= 500; }
  static constexpr int pick_wa() {
    return (regs_ok(4) && lds(4, 1) <= LDS_MAX) ? 4 : (regs_ok(3) && lds(3, 1) <= LDS_MAX) ? 3
         : (regs_ok(2) && lds(2, 1) <= LDS_MAX) ? 2 : 0;
  }
  static constexpr int WA = pick_wa();
  static constexpr int NB = (WA > 0 && lds(WA, 2) <= LDS_MAX) ? 2 : 1;
  static constexpr bool ok = WA >= 2;
  static constexpr size_t LDS = lds(WA > 0 ? WA : 1, NB);
  static constexpr int DMAE = 8, TAIL = 12;
  // VARW_BREG (slice_gemm_y_tile.h): the B fragments global -> VGPR into two named register sets of 2 * S fragments
  // (v[80:223]: S <= 9; the accumulators are named registers too: a[0:255], v[224:255]), LDS holds the two A stages only.  64 x 128 tiles; where the LDS form has the same tile height it
  // stages a third of the bytes into LDS and reads two thirds of the fragments from it.
  static constexpr int BREG_WA = 2;
  static constexpr bool breg_ok = S <= 9 && WA == BREG_WA && BREG_WA * S * 16 + 16 * S + 4 * 4 + 4 + 24 <= 512;
  static constexpr size_t BREG_LDS = (size_t)(2 * BREG_WA) * (2 * S) * FRAG_BYTES;
  // VARW_ACCN | VARW_BHI (round 5): 11 and 12 staged slices.  Two A stages + one B stage of a 64 x 128 tile are 176 / 192 KiB; with
  // the highest B slices (9, 10 / 8 .. 11) loaded global -> named VGPRs and refilled in place (slice_gemm_y_tile.h) the B stage
  // holds 9 / 8 slices: 88 + 72 = 96 + 64 = 160 KiB.  352 / 384 accumulator registers + 88 / 96 of B fragments: the accumulators
  // are named registers (a[0:255] + v[160:255] / v[128:255]); at 12 slices the compiler is left v[0:95] and the A ring is 2 deep.
  static constexpr int HYB_WA = 2, HYB_SLB = S == 12 ? 8 : 9;
  static constexpr bool hyb_ok = S == 11 || S == 12;
  static constexpr size_t HYB_LDS = (size_t)(2 * HYB_WA * 2 * S + 4 * 2 * HYB_SLB) * FRAG_BYTES;
  static_assert(!hyb_ok || (HYB_LDS <= LDS_MAX && !ok), "the hybrid form exists where the LDS form does not fit");
};
// (B global -> VGPR: the step has ONE form - every step prefetches, the last one wraps around - and the loop body holds two
// steps: passes with k-blocks = 0 mod 4; kernel_policy.cpp offers the form only there.)

template <int S, int ND, int WA, int VARW, int DMAE, int TAIL>
static hipError_t launch_wide_kernel(const SliceGemmArgs &a0, const WidePlan &pl, size_t lds, hipStream_t stream) {
  auto kernel = slice_gemm_w_kernel<S, 0, ND, WA, VARW, 0, -1, DMAE, TAIL>;
  SliceGemmArgs a = a0;
  const uint32_t nb = wide_grid(a, pl);
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = allow_dynamic_lds(kernel, lds, attr_done, a.device)) return e;
  hipLaunchKernelGGL(kernel, dim3(nb, a.batch > 1 ? a.batch : 1), dim3(256), lds, stream, a);
  return hipGetLastError();
}
template <int S, int ND>
static hipError_t launch_wide_k64(const SliceGemmArgs &a, const WidePlan &pl, bool breg, hipStream_t stream) {
  using C = K64Cfg<ND>;
  if constexpr (C::hyb_ok) {
    return launch_wide_kernel<S, ND, C::HYB_WA, VARW_K64 | VARW_B1 | VARW_ACCN | VARW_BHI, C::DMAE, C::TAIL>(a, pl, C::HYB_LDS, stream);
  } else {
    if constexpr (C::breg_ok) {
      if (breg && ((a.kb1 - a.kb0) & 3u) == 0)
        return launch_wide_kernel<S, ND, C::BREG_WA, VARW_K64 | VARW_BREG | VARW_ACCN, C::DMAE, C::TAIL>(a, pl, C::BREG_LDS, stream);
    }
    return launch_wide_kernel<S, ND, C::WA, VARW_K64 | (C::NB == 1 ? VARW_B1 : 0), C::DMAE, C::TAIL>(a, pl, C::LDS, stream);
  }
}

template <int S, int D0, int ND, bool X16 = false>
static hipError_t launch_wide(const SliceGemmArgs &a0, const WidePlan &pl, hipStream_t stream) {
  using Cfg = WideCfg<S, D0, ND>;
  using XCfg = PairedCfg<S, D0, ND>;
  constexpr int VARW = (X16 ? VARW_X16 : (Cfg::NA == 3 ? VARW_NA3 : 0)) | (Cfg::NB == 1 ? VARW_B1 : 0);
  constexpr size_t lds = X16 ? XCfg::lds(Cfg::WA) : Cfg::lds(Cfg::WA, Cfg::NA, Cfg::NB);
  constexpr int DMAE = X16 ? XCfg::DMAE : 4, TAIL = X16 ? XCfg::TAIL : 6;
  auto kernel = slice_gemm_w_kernel<S, D0, ND, Cfg::WA, VARW, 0, -1, DMAE, TAIL>;
  SliceGemmArgs a = a0;
  const uint32_t nb = wide_grid(a, pl);
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = allow_dynamic_lds(kernel, lds, attr_done, a.device)) return e;
  hipLaunchKernelGGL(kernel, dim3(nb, a.batch > 1 ? a.batch : 1), dim3(256), lds, stream, a);
  return hipGetLastError();
}

// one persistent launch for the 2..4 products of a ZGEMM (slice_gemm_w_kernel.h: slice_gemm_w_multi_kernel)
template <int S, bool X16>
static hipError_t launch_wide_multi_impl(const SliceGemmArgs *g, int count, const WidePlan &pl, hipStream_t stream) {
  using Cfg = WideCfg<S, 0, S>;
  using XCfg = PairedCfg<S, 0, S>;
  constexpr int VARW = (X16 ? VARW_X16 : (Cfg::NA == 3 ? VARW_NA3 : 0)) | (Cfg::NB == 1 ? VARW_B1 : 0);
  constexpr size_t lds = X16 ? XCfg::lds(Cfg::WA) : Cfg::lds(Cfg::WA, Cfg::NA, Cfg::NB);
  constexpr int DMAE = X16 ? XCfg::DMAE : 4, TAIL = X16 ? XCfg::TAIL : 6;
  auto kernel = slice_gemm_w_multi_kernel<S, 0, S, Cfg::WA, VARW, 0, -1, DMAE, TAIL>;
  SliceGemmMulti m{};
  m.count = count;
  uint32_t nb = 0;
  for (int i = 0; i < count; i++) {
    m.g[i] = g[i];
    m.g[i].qslot = g[0].qslot; // one set of claim counters: a claimed tile is walked through every product
    nb = wide_grid(m.g[i], pl);
  }
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = allow_dynamic_lds(kernel, lds, attr_done, g[0].device)) return e;
  hipLaunchKernelGGL(kernel, dim3(nb), dim3(256), lds, stream, m);
  return hipGetLastError();
}
template <int S, bool BREG = false>
static hipError_t launch_wide_multi_k64(const SliceGemmArgs *g, int count, const WidePlan &pl, bool breg, hipStream_t stream) {
  using C = K64Cfg<S>;
  if constexpr (C::breg_ok && !BREG) {
    if (breg && ((g[0].kb1 - g[0].kb0) & 3u) == 0) return launch_wide_multi_k64<S, true>(g, count, pl, true, stream);
  }
  constexpr int WAK = BREG ? C::BREG_WA : C::WA;
  constexpr size_t LDSK = BREG ? C::BREG_LDS : C::LDS;
  auto kernel = slice_gemm_w_multi_kernel<S, 0, S, WAK, VARW_K64 | (BREG ? (VARW_BREG | VARW_ACCN) : (C::NB == 1 ? VARW_B1 : 0)), 0, -1, C::DMAE, C::TAIL>;
  SliceGemmMulti m{};
  m.count = count;
  uint32_t nb = 0;
  for (int i = 0; i < count; i++) {
    m.g[i] = g[i];
    m.g[i].qslot = g[0].qslot;
    nb = wide_grid(m.g[i], pl);
  }
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = allow_dynamic_lds(kernel, LDSK, attr_done, g[0].device)) return e;
  hipLaunchKernelGGL(kernel, dim3(nb), dim3(256), LDSK, stream, m);
  return hipGetLastError();
}
template <int S>
static hipError_t launch_wide_multi(const SliceGemmArgs *g, int count, const WidePlan &pl, bool x16, hipStream_t stream) {
  if constexpr (PairedCfg<S, 0, S>::ok) {
    if (x16) return launch_wide_multi_impl<S, true>(g, count, pl, stream);
  }
  return launch_wide_multi_impl<S, false>(g, count, pl, stream);
}

// what the pass <S, D0, ND> can run on (kernel_policy.h), from the configuration structs above
template <int S, int D0, int ND>
static constexpr PassTraits pass_traits() {
  PassTraits t{};
  t.S = S;
  t.D0 = D0;
  t.ND = ND;
  t.SL = (D0 + ND < S) ? (D0 + ND) : S;
  int c = 0;
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) c += (i + j >= D0 && i + j < D0 + ND && i + j <= S - 1) ? 1 : 0;
  t.pairs = c;
  t.k2_ok = K2Cfg<S, D0, ND>::ok;
  t.wide_ok = WideCfg<S, D0, ND>::ok;
  t.wide_wa = WideCfg<S, D0, ND>::WA;
  t.x16_ok = WideCfg<S, D0, ND>::ok && PairedCfg<S, D0, ND>::ok;
  if constexpr (D0 == 0) {
    t.k64_ok = K64Cfg<ND>::ok || K64Cfg<ND>::hyb_ok;
    t.k64_wa = K64Cfg<ND>::hyb_ok ? K64Cfg<ND>::HYB_WA : K64Cfg<ND>::WA;
    t.k64_breg_ok = K64Cfg<ND>::breg_ok;
  }
  t.classic_wm4 = S <= 6 && D0 == 0 && ND == S;
  t.classic_form = (t.SL >= 11 && t.SL <= 13) ? 1 : (t.SL >= 14 ? 2 : 0); // launch_one: WM by staged slices (LDS)
  return t;
}

// kernel choice for one pass over the diagonals [D0, D0 + ND): the cost model of kernel_policy.cpp predicts every kernel the
// pass is built for and takes the fastest (OZIMMU_HIP_GEMM_KERNEL / _K64_TILE / _PAIRED_TILE / _K64_BREG override).  `pl`: the
// tile plan of the chosen wide kernel; `breg`: k64 with the B fragments in registers; `wm4`: the classic kernel's 128x64 form.
template <int S, int D0, int ND>
static Pick pick_kernel(const SliceGemmArgs &a, WidePlan &pl, bool &breg, bool &wm4) {
  constexpr PassTraits t = pass_traits<S, D0, ND>();
  PolicyInput in;
  in.M = a.M;
  in.N = a.N;
  in.nkb = a.kb1 - a.kb0;
  in.batch = a.batch > 1 ? a.batch : 1;
  const Prediction r = policy_predict(t, in, topology(a.device), config());
  pl = r.plan[r.breg ? 5 : (int)r.pick];
  breg = r.breg;
  wm4 = r.classic_wm4;
  return r.pick;
}

template <int S, int D0, int ND>
static hipError_t launch_pass(const SliceGemmArgs &a, hipStream_t stream) {
  WidePlan pl;
  bool breg = false, wm4 = false;
  const Pick pick = pick_kernel<S, D0, ND>(a, pl, breg, wm4);
  note_pick(D0 > 0 ? 1 : 0, (int)pick + (breg ? 8 : 0)); // diagnostics: ozimmu_hip_last_kernel
  switch (pick) {
  case Pick::K2:
    if constexpr (K2Cfg<S, D0, ND>::ok) return launch_k2<S, D0, ND>(a, stream);
    break;
  case Pick::WIDE_K64:
    if constexpr (D0 == 0 && (K64Cfg<ND>::ok || K64Cfg<ND>::hyb_ok)) return launch_wide_k64<S, ND>(a, pl, breg, stream);
    break;
  case Pick::WIDE_X16:
    if constexpr (WideCfg<S, D0, ND>::ok && PairedCfg<S, D0, ND>::ok) return launch_wide<S, D0, ND, true>(a, pl, stream);
    break;
  case Pick::WIDE:
    if constexpr (WideCfg<S, D0, ND>::ok) return launch_wide<S, D0, ND>(a, pl, stream);
    break;
  default:
    break;
  }
  if constexpr (S <= 6 && D0 == 0 && ND == S) {
    // few slices = few MFMAs per staged byte: the 128x64 8-wave workgroup (-25 % staged bytes) once there are enough tiles
    if (wm4) return launch_one<S, D0, ND, 4>(a, stream);
  }
  return launch_one<S, D0, ND>(a, stream);
}

// The real products of a ZGEMM (2..4 slice GEMMs that accumulate into the same C in the given order, each a single diagonal
// pass over one K chunk) as ONE launch: the K-split kernel for problems of at most one tile per CU, the persistent wide
// kernel - every claimed tile walked through all products before the next claim - for problems that fill the chip
// (three launch boundaries and three idle tails less).  hipErrorNotSupported: the caller launches them one by one.
template <int S>
static hipError_t launch_fused(const SliceGemmArgs *g, int count, hipStream_t stream) {
  if constexpr (S > SINGLE_PASS_MAX_S) {
    return hipErrorNotSupported;
  } else {
    const SliceGemmArgs &a0 = g[0];
    if (count < 2 || count > 4 || !config().fused_products) return hipErrorNotSupported;
    for (int i = 0; i < count; i++)
      if (g[i].M != a0.M || g[i].N != a0.N || g[i].batch != a0.batch || g[i].kb0 != a0.kb0 || g[i].kb1 != a0.kb1)
        return hipErrorNotSupported;
    WidePlan pl;
    bool breg = false, wm4 = false;
    const Pick pick = pick_kernel<S, 0, S>(a0, pl, breg, wm4);
    note_pick(0, (int)pick + (breg ? 8 : 0));
    if (pick == Pick::K2) return launch_k2_fused<S>(g, count, stream);
    // (S = 7 keeps 448 accumulator registers per wave: walking several argument sets per tile spills inside its k loop)
    if constexpr (WideCfg<S, 0, S>::ok && WideCfg<S, 0, S>::WA * S * 16 <= 432) {
      // Up to ~8 tiles per CU the three saved launch boundaries / idle tails pay (profiles/r3_ablate/r3i_zgemm_one_launch_ab.txt:
      // 2048^3 -5 % time, 4096^3 -1..2 %); beyond that they do not, and the CUs of an XCD drift over four different panel
      // pairs in its L2 (8192^3: +1..2 % time): large products keep one launch each.
      const uint64_t tiles = (uint64_t)(pl.n_big + pl.n_small) * ((a0.N + 127) / 128);
      const bool few_tiles = tiles <= 8ull * (uint64_t)cu_count(a0) || config().wide_grid > 0;
      if constexpr (K64Cfg<S>::ok) {
        if (pick == Pick::WIDE_K64 && a0.batch <= 1 && a0.phase && few_tiles) return launch_wide_multi_k64<S>(g, count, pl, breg, stream);
      }
      if ((pick == Pick::WIDE || pick == Pick::WIDE_X16) && a0.batch <= 1 && a0.phase && few_tiles)
        return launch_wide_multi<S>(g, count, pl, pick == Pick::WIDE_X16, stream);
    }
    return hipErrorNotSupported;
  }
}

// S <= SINGLE_PASS_MAX_S: all S diagonals in one pass.  Larger S: two diagonal ranges, the second pass
// continues the first one's fma chain through the FP64 `acc` workspace (same summation order).
template <int S>
static hipError_t launch_S(SliceGemmArgs a, hipStream_t stream) {
  if constexpr (S <= SINGLE_PASS_MAX_S) {
    return launch_pass<S, 0, S>(a, stream);
  } else {
    constexpr int ND1 = (S + 1) / 2, ND2 = S - ND1;
    SliceGemmArgs a1 = a;
    a1.final = 0; // -> acc
    hipError_t e = launch_pass<S, 0, ND1>(a1, stream);
    if (e != hipSuccess) return e;
    SliceGemmArgs a2 = a;
    a2.acc_in = 1;
    a2.qslot = a.qslot + 1; // own claim counters (the host advances qslot by 2 per call of launch_slice_gemm)
    if (a2.dump) a2.dump += (size_t)ND1 * a.N * a.M;
    return launch_pass<S, ND1, ND2>(a2, stream);
  }
}


template <int S>
static hipError_t dispatch_S(int s, const SliceGemmArgs &a, hipStream_t stream) {
  if constexpr (S > OZ_S_HI) {
    return hipErrorInvalidValue;
  } else {
    if (s == S) return launch_S<S>(a, stream);
    return dispatch_S<S + 1>(s, a, stream);
  }
}

hipError_t OZ_PART(int S, const SliceGemmArgs &a, hipStream_t stream) { return dispatch_S<OZ_S_LO>(S, a, stream); }

// the traits of mode S's passes (pass 0: the single / first pass, 1: the second pass of S > 12) for the diagnostics of api.cpp
template <int S>
static bool traits_S(int s, int pass, PassTraits *out) {
  if constexpr (S > OZ_S_HI) {
    return false;
  } else {
    if (s != S) return traits_S<S + 1>(s, pass, out);
    if constexpr (S <= SINGLE_PASS_MAX_S) {
      if (pass != 0) return false;
      *out = pass_traits<S, 0, S>();
    } else {
      constexpr int ND1 = (S + 1) / 2, ND2 = S - ND1;
      if (pass == 0) *out = pass_traits<S, 0, ND1>();
      else if (pass == 1) *out = pass_traits<S, ND1, ND2>();
      else return false;
    }
    return true;
  }
}
bool OZ_PART_TRAITS(int S, int pass, PassTraits *out) { return traits_S<OZ_S_LO>(S, pass, out); }

template <int S>
static hipError_t dispatch_fused_S(int s, const SliceGemmArgs *g, int count, hipStream_t stream) {
  if constexpr (S > OZ_S_HI) {
    return hipErrorNotSupported;
  } else {
    if (s == S) return launch_fused<S>(g, count, stream);
    return dispatch_fused_S<S + 1>(s, g, count, stream);
  }
}

hipError_t OZ_PART_FUSED(int S, const SliceGemmArgs *g, int count, hipStream_t stream) {
  return dispatch_fused_S<OZ_S_LO>(S, g, count, stream);
}

} // namespace ozhip
