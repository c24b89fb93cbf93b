// kernel_policy.h — which slice-GEMM kernel runs a pass: ONE cost model instead of per-shape rules.
//
// Rounds 1-3 grew the choice between five kernels (K-split, classic 64x64, wide 32x32x32, wide paired 16x16x64, wide k64
// 16x16x64 with B through LDS or in registers) as a list of measured exceptions - bars of 6 / 11 / 12 / 15 / 16 / 24 us of
// k loop, efficiencies 0.62 / 0.9, "0.92 of a block", CU fractions 4/10 and 7/10, S-specific cases.  Every new tile function
// added rules.  This file replaces them by a prediction of each eligible kernel's time from the same few quantities,
//     time = launch + rounds x (blocks per tile x pairs x k-blocks x t_mfma x (a + beta x staged KiB per MFMA) + per-tile cost),
// with the rounds of the wide family taken from the closed-form greedy makespan of tile_plan.h and (a, beta, per-tile,
// launch) fitted per kernel family to measured GEMM-stage times (tools/policy_fit.py; profiles/r4_policy/); the policy takes
// the minimum.  Plain C++ (no device code): a change of the model recompiles in a second, not in the ten minutes the kernel
// instantiations take.  The reference has no counterpart (src/gemm.cu:315-329 calls cuBLAS for every pair).
#pragma once
#include <cstdint>

#include "config.h"
#include "tile_plan.h"
#include "topology.h"

namespace ozhip {

// what a pass <S, D0, ND> can run on, filled in from the compile-time configuration structs (slice_gemm_launch.h)
struct PassTraits {
  int S = 0, D0 = 0, ND = 0;
  int SL = 0;    // staged slices
  int pairs = 0; // slice products of the pass
  bool k2_ok = false;
  bool wide_ok = false;
  int wide_wa = 0;
  bool x16_ok = false;
  bool k64_ok = false;
  int k64_wa = 0;
  bool k64_breg_ok = false;
  bool classic_wm4 = false; // S <= 6 single pass: the 8-wave 128x64 form of the classic kernel exists
  int classic_form = 0;     // what launch_one runs by default: 0 = 64x64 x two workgroups per CU, 1 = 128x64 (8 waves, one per
                            // CU: 11-13 staged slices), 2 = 64x64, one workgroup per CU (14+ staged slices)
};

enum class Pick { K2 = 0, CLASSIC = 1, WIDE = 2, WIDE_X16 = 3, WIDE_K64 = 4 };
constexpr int POLICY_KERNELS = 6; // prediction slots: the five picks + [5] = k64 with the B fragments in registers

struct PolicyInput {
  uint32_t M = 0, N = 0, nkb = 0, batch = 1;
};

struct Prediction {
  double us[POLICY_KERNELS]; // predicted GEMM-stage time; < 0: not eligible
  WidePlan plan[POLICY_KERNELS];
  Pick pick = Pick::CLASSIC;
  bool breg = false;     // pick == WIDE_K64: the B-in-registers form
  bool classic_wm4 = false; // pick == CLASSIC: the 128x64 form
};

// the fitted constants: POLICY_PARAMS doubles, see kernel_policy.cpp for their meaning
constexpr int POLICY_PARAMS = 40;
// the table is read as a snapshot and replaced as a whole under one lock (ozimmu_hip_policy_params: tools/policy_fit.py's loop
// may replace it while another thread's handle is planning a launch)
void policy_params_get(double out[POLICY_PARAMS]);
void policy_params_set(const double *in, int count); // the first `count` entries; non-finite values are ignored

Prediction policy_predict(const PassTraits &t, const PolicyInput &in, const Topology &topo, const Config &cfg);

// bookkeeping for diagnostics: the kernel the calling thread's last pass launched (Pick, + 8 for the k64 register form)
void note_pick(int pass_index, int code);
int last_pick(int pass_index);

// slice_gemm.hip: the traits of mode S's passes (0: the single / first pass, 1: the second pass of S > 12)
bool slice_gemm_traits(int S, int pass, PassTraits *out);

} // namespace ozhip
