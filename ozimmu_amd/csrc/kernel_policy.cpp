// kernel_policy.cpp — the cost model behind the slice-GEMM kernel choice (kernel_policy.h).
#include "kernel_policy.h"

#include "diagnostics.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <mutex>

namespace ozhip {

// ---- the fitted table ----------------------------------------------------------------------------------------------------
// Unit of work: w = pairs x k-blocks x t_mfma = what one wave spends on the MFMAs of one 32 x 32 block of C over the pass
// (t_mfma: topology().mfma32_us, calibrated per device).  rho = KiB staged into the CU per 32x32x32-MFMA-equivalent of the
// tile: (h + 4) SL / (4 h pairs) for a wide tile of h blocks x 128 columns, SL / pairs for the 64 x 64 tiles.
//   tile time   = blocks per wave x w x (a + beta x rho) + b          [us]
//   kernel time = f + makespan of the tiles on the CUs
// Fitted by tools/policy_fit.py to the times of every forced kernel (whole QUEUED calls - the steady state of a power-limited part -
// with the split time, the same for every kernel, measured next to them; least squares on log((predicted + split) / measured),
// then a few Powell sweeps on a smooth surrogate of the regret).  Round 5 refit: the kernels changed (fp64_int8_11 / 12 on the k64
// tile, the register kernel's recombination under its last step), so the data is this round's - 430 shapes x S in {4 .. 12} incl.
// 80 short-K panels under 4096^2 .. 32768^2 outputs (profiles/r5_policy/policy_q2_*) + the round-4 rows of the modes whose kernels
// did not change.  Regret of the resulting choice (time with the picked kernel / with the best measured one - 1): mean 0.56 %,
// 93 % of the cases within 3 % on the fit data and on 200 held-out cases alike (round-4 constants on the same data: 0.71 %, 92 %);
// what is left is k loops of <= 16 k-blocks, where kernels differ by less than the model's 11 % rms error and a per-class bias
// table fitted to the residuals changes nothing (the misses are shape-specific, not systematic).  `ozimmu_hip_policy_params`
// reads / replaces the table at run time.
enum : int {
  K2_A, K2_BETA, K2_F, K2_STEP,
  CL_A, CL_A1, CL_BETA, CL_B, CL_F, CL_STEP, CL4_A, CL4_STEP,
  W_A, W_BETA, W_B, W_STEP,
  X_A, X_BETA, X_B, X_STEP,
  Y_A, Y_BETA, Y_B, Y_STEP,
  Z_A, Z_BETA, Z_B, Z_STEP,
  WIDE_F,      // launch cost of the wide family (persistent grid, claim counters)
  GAMMA,       // time factor of the MFMA work when (almost) no CU is busy (the part is power limited: the fewer CUs work, the
               // higher they clock); factor = GAMMA + (1 - GAMMA) x busy fraction
  EPI_W, EPI_Y, // epilogue cost per 32-row block and diagonal of a tile (32x32 / 16x16 C layout)
  C_US_PER_MB, // every kernel writes M x N x 8 bytes of C (short K: a fifth of the launch is this store traffic)
  CL_DRIFT,    // the 64 x 64 tiles lose L2 panel sharing over a long k loop (their workgroups drift apart): the classic kernel's
               // MFMA work x (1 + CL_DRIFT x (log2(k-blocks) - 5) / 3) beyond 32 k-blocks (= 1 + CL_DRIFT at K = 8192)
  ST_H,        // wide family: the exposed part of a tile's C stores, per 32-row block (one workgroup per CU: nothing overlaps them)
  CL_EPI,      // the 64 x 64 kernels' recombination per tile and diagonal (a short k loop under a large output: a fifth of the tile)
  CL_C_US_PER_MB, // ... and what the 64 x 64 kernels' store of C costs per MB (8-byte stores from 2 x 256 workgroups; the wide family: C_US_PER_MB + ST_H)
  CL_ROUND,    // 64 x 64 tiles, form 0: what every round of tiles beyond the first costs on top (tile switch with both workgroups of a CU
               // in their prologue / epilogue at once: the steady state of many rounds, which a single round does not show)
  DEV_CUS, DEV_MFMA_US // not model constants: the device of a prediction made without a handle (api.cpp)
};
static_assert(DEV_MFMA_US + 1 == POLICY_PARAMS, "parameter table");
// *_STEP: cost of one k-step of a tile beyond its MFMAs (barrier, waits, loop control): per 32-k step for the 32x32x32 tile
// functions and the 64 x 64 kernels, per 64-k step for the k64 tile.  With few slices a step holds few MFMAs (S = 4: 10 pairs)
// and this term is what separates the kernels.

static double g_params[POLICY_PARAMS] = {
    // Round 6 refit (profiles/r6_policy/fit_r6_log.txt): the register kernel's tile boundary changed (profiles/r6_ablate/), so its rows
    // are this round's - 250 shapes at S = 9, 90 of them short-K panels, every kernel forced - next to the round-5 rows of the
    // other modes.  Regret on those 1 289 cases 0.71 -> 0.63 % mean, cases above 3 % 114 -> 99; on the S = 9 rows alone 1.61 -> 1.20 %.
    /* K2  a, beta, f, step           */ 1.021, 0, 0, 0.001605,
    /* CL  a, a1, beta, b, f, step    */ 1.049, 1.241, 0, 0, 0, 0.1006,
    /* CL4 a, step                    */ 0.994, 0.6352,
    /* W   a, beta, b, step           */ 0.7995, 0.3492, 0.4029, 0.3186,
    /* X   a, beta, b, step           */ 0.7948, 0, 1.3, 0.4383,
    /* Y   a, beta, b, step           */ 0.6875, 0.1166, 2.324, 0.4263,
    /* Z   a, beta, b, step           */ 0.5214, 1.328, 3.155, 0,
    /* wide f, gamma, epi_w, epi_y    */ 1.138, 0.7529, 0.1499, 0.1394,
    /* C us per MB                    */ 0.1487,
    /* cl drift, store per block      */ 0, 0.7503,
    /* cl epi, cl C us per MB, round  */ 0.06178, 0, 3.687,
    0.0, 0.0};

static std::mutex g_params_mtx;
void policy_params_get(double out[POLICY_PARAMS]) {
  std::lock_guard<std::mutex> lock(g_params_mtx);
  std::copy(g_params, g_params + POLICY_PARAMS, out);
}
void policy_params_set(const double *in, int count) {
  std::lock_guard<std::mutex> lock(g_params_mtx);
  for (int i = 0; i < count && i < POLICY_PARAMS; i++)
    if (std::isfinite(in[i])) g_params[i] = in[i];
}

static thread_local int t_last_pick[2] = {-1, -1};
static std::atomic<unsigned long long> g_pick_hist[PICK_HIST_SLOTS];
void note_pick(int pass_index, int code) {
  if (pass_index >= 0 && pass_index < 2) t_last_pick[pass_index] = code;
  if (pass_index == 0 && code >= 0 && code < PICK_HIST_SLOTS) g_pick_hist[code].fetch_add(1, std::memory_order_relaxed);
}
void pick_histogram(unsigned long long out[PICK_HIST_SLOTS]) {
  for (int i = 0; i < PICK_HIST_SLOTS; i++) out[i] = g_pick_hist[i].load(std::memory_order_relaxed);
}
int last_pick(int pass_index) { return pass_index >= 0 && pass_index < 2 ? t_last_pick[pass_index] : -1; }

// kernel_tuner.h: the prediction slot a tuned call of this thread runs (a candidate being timed, or the measured winner)
static thread_local int t_override = -1;
void policy_override(int slot);
void policy_override(int slot) { t_override = slot; }

Prediction policy_predict(const PassTraits &t, const PolicyInput &in, const Topology &topo, const Config &cfg) {
  double p[POLICY_PARAMS];
  policy_params_get(p);
  Prediction r;
  for (double &x : r.us) x = -1.0;
  const uint32_t batch = in.batch > 1 ? in.batch : 1;
  const int cus = topo.cus > 0 ? topo.cus : 1;
  const double w = (double)t.pairs * (double)in.nkb * topo.mfma32_us;
  const double rho64 = (double)t.SL / (double)t.pairs; // 64 x 64 tiles: 4 SL KiB per k-block for 4 x pairs MFMAs
  const uint64_t tiles64 = (uint64_t)((in.M + 63) / 64) * ((in.N + 63) / 64) * batch;
  const Config::Kernel forced = cfg.gemm_kernel;
  const bool is_forced = cfg.forced_kernel();
  const double c_us = p[C_US_PER_MB] * 8e-6 * (double)in.M * (double)in.N * (double)batch; // the store of C (wide family)
  const double c_us_cl = p[CL_C_US_PER_MB] * 8e-6 * (double)in.M * (double)in.N * (double)batch; // ... of the 64 x 64 kernels
  auto boost = [&](double busy_fraction) { // < 1: few busy CUs clock higher under the package power cap
    const double phi = busy_fraction < 1.0 ? busy_fraction : 1.0;
    return p[GAMMA] + (1.0 - p[GAMMA]) * phi;
  };
  const double g64 = boost((double)tiles64 / (double)cus);

  // ---- K-split kernel: one 8-wave workgroup per tile of 64 x 64, at most one tile per CU ---------------------------------
  if (t.k2_ok && (forced == Config::K2 || (!is_forced && tiles64 <= (uint64_t)cus && in.nkb >= 4))) {
    const uint64_t rounds = (tiles64 + (uint64_t)cus - 1) / (uint64_t)cus; // > 1 only when forced
    // (each of the two wave groups walks half of the k-blocks)
    r.us[(int)Pick::K2] = c_us_cl + p[K2_F] + (double)rounds * (w * g64 * (p[K2_A] + p[K2_BETA] * rho64) + 0.5 * in.nkb * p[K2_STEP] + p[CL_EPI] * t.ND);
  }

  // ---- classic kernel: 64 x 64 tiles of 4 waves, two workgroups per CU (S <= 6 on many tiles: 128 x 64 tiles of 8 waves) --
  {
    const uint64_t tiles128 = (uint64_t)((in.M + 127) / 128) * ((in.N + 63) / 64) * batch;
    const double lg = std::log2((double)(in.nkb > 0 ? in.nkb : 1));
    const double drift = g64 * (1.0 + p[CL_DRIFT] * (lg > 5.0 ? (lg - 5.0) / 3.0 : 0.0));
    const double e2 = drift * (p[CL_A] + p[CL_BETA] * rho64), e1 = drift * (p[CL_A1] + p[CL_BETA] * rho64);
    double T;
    if (t.classic_form == 0) {
      const uint64_t slots = 2ull * (uint64_t)cus, q = tiles64 / slots, rem = tiles64 % slots;
      const double st = (double)in.nkb * p[CL_STEP] + p[CL_EPI] * t.ND;
      T = p[CL_F] + (double)q * (2.0 * w * e2 + st + p[CL_B]) + (q > 1 ? (double)(q - 1) * p[CL_ROUND] : 0.0);
      if (rem) T += rem <= (uint64_t)cus ? (w * e1 + st + p[CL_B]) : (2.0 * w * e2 + st + p[CL_B]);
    } else if (t.classic_form == 1) { // 11-13 staged slices: one 8-wave 128 x 64 workgroup per CU (two blocks per SIMD)
      const uint64_t rounds = (tiles128 + (uint64_t)cus - 1) / (uint64_t)cus;
      T = p[CL_F] + (double)rounds * (2.0 * w * drift * (p[CL4_A] + p[CL_BETA] * 0.75 * rho64) + in.nkb * p[CL4_STEP] + p[CL_B] + 2.0 * p[CL_EPI] * t.ND);
    } else { // 14+ staged slices: 64 x 64, one workgroup per CU
      const uint64_t rounds = (tiles64 + (uint64_t)cus - 1) / (uint64_t)cus;
      T = p[CL_F] + (double)rounds * (w * e1 + in.nkb * p[CL_STEP] + p[CL_B] + p[CL_EPI] * t.ND);
    }
    r.us[(int)Pick::CLASSIC] = c_us_cl + T;
    T += c_us_cl;
    if (t.classic_wm4 && tiles128 >= 512) { // (the threshold of launch_S: the 8-wave form needs enough tiles to fill the chip)
      const uint64_t rounds = (tiles128 + (uint64_t)cus - 1) / (uint64_t)cus;
      const double T4 = c_us_cl + p[CL_F] + (double)rounds * (2.0 * w * drift * (p[CL4_A] + p[CL_BETA] * 0.75 * rho64) + in.nkb * p[CL4_STEP] + p[CL_B] + 2.0 * p[CL_EPI] * t.ND);
      if (T4 < T) {
        r.us[(int)Pick::CLASSIC] = T4;
        r.classic_wm4 = true;
      }
    }
  }

  // ---- the wide family: (32 h) x 128 tiles, one 4-wave workgroup per CU, tile heights mixed by the planner ---------------
  const int ncu_eff = std::max(1, cus / (int)batch); // a strided batch fills the chip with all its matrices
  auto wide = [&](int slot, int wa, int ia, int iepi, int kb_per_step) {
    const uint64_t tiles = (uint64_t)((in.M + 32 * wa - 1) / (32 * wa)) * ((in.N + 127) / 128);
    const double g = boost((double)tiles / (double)ncu_eff);
    const double steps = (double)(in.nkb / kb_per_step);
    auto cost = [&](int h) {
      const double rho = (double)(h + 4) * t.SL / (4.0 * h * t.pairs);
      return (double)h * w * g * (p[ia] + p[ia + 1] * rho) + steps * p[ia + 3] + p[ia + 2] + p[iepi] * h * t.ND + p[ST_H] * h;
    };
    r.plan[slot] = plan_wide_costs(in.M, in.N, wa, ncu_eff, cost);
    r.us[slot] = c_us + p[WIDE_F] + r.plan[slot].makespan;
  };
  // the wide kernels keep the k position of a pass in a 32-bit byte offset: a pass stays below 2^32 bytes per row-block
  const bool voff_ok = (uint64_t)in.nkb * (uint64_t)(t.S * 1024) < (1ull << 32);
  if (t.wide_ok && voff_ok) {
    wide((int)Pick::WIDE, t.wide_wa, W_A, EPI_W, 1);
    if (t.x16_ok && cfg.paired_tile != 0) wide((int)Pick::WIDE_X16, t.wide_wa, X_A, EPI_Y, 1);
    if (t.k64_ok && t.D0 == 0 && (in.nkb & 1u) == 0 && (cfg.k64_tile != 0 || forced == Config::K64)) {
      wide((int)Pick::WIDE_K64, t.k64_wa, Y_A, EPI_Y, 2);
      if (t.k64_breg_ok && (in.nkb & 3u) == 0 && cfg.k64_breg != 0) wide(5, t.k64_wa, Z_A, EPI_Y, 2);
    }
  }

  // ---- the choice ----------------------------------------------------------------------------------------------------------
  auto ok = [&](int s) { return r.us[s] >= 0; };
  auto argmin = [&]() {
    int b = (int)Pick::CLASSIC;
    for (int s = 0; s < POLICY_KERNELS; s++)
      if (ok(s) && r.us[s] < r.us[b]) b = s;
    return b;
  };
  constexpr int K64 = (int)Pick::WIDE_K64, K64R = 5;
  int best;
  if (is_forced) { // OZIMMU_HIP_GEMM_KERNEL: that kernel where it exists for the pass (tests, A/B tools)
    const int want = forced == Config::K2 ? (int)Pick::K2 : forced == Config::CLASSIC ? (int)Pick::CLASSIC
                   : forced == Config::WIDE ? (int)Pick::WIDE : forced == Config::X16 ? (int)Pick::WIDE_X16 : K64;
    if (ok(want)) best = want;
    else if (want >= (int)Pick::WIDE && ok((int)Pick::WIDE)) best = (int)Pick::WIDE; // that tile function is not built here
    else best = argmin();
  } else {
    best = argmin();
    // OZIMMU_HIP_K64_TILE=1 / OZIMMU_HIP_PAIRED_TILE=1: that tile function wherever a wide kernel runs and it exists
    if (best >= (int)Pick::WIDE) {
      if (cfg.k64_tile == 1 && ok(K64)) best = K64;
      else if (cfg.paired_tile == 1 && ok((int)Pick::WIDE_X16)) best = (int)Pick::WIDE_X16;
    }
  }
  const bool tuned = !is_forced && t_override >= 0 && t_override < POLICY_KERNELS && ok(t_override);
  if (tuned) best = t_override; // measured choice (kernel_tuner.cpp): every kernel returns the same bits
  // k64: B through LDS or in registers (OZIMMU_HIP_K64_BREG=0 keeps the register form out of `ok`, =1 takes it wherever it exists)
  if (!tuned && (best == K64 || best == K64R) && ok(K64R)) best = (cfg.k64_breg == 1 || r.us[K64R] <= r.us[K64]) ? K64R : K64;
  r.breg = best == 5;
  r.pick = best == 5 ? Pick::WIDE_K64 : (Pick)best;
  if (r.pick != Pick::CLASSIC) r.classic_wm4 = false;
  return r;
}

} // namespace ozhip
