// config.h — the OZIMMU_HIP_* switches (kernel choice overrides, tuning knobs of the A/B tools, test hooks), read from the
// environment ONCE, when the library first needs one.  The reference reads only OZIMMU_COMPUTE_MODE and the auto-mode
// threshold per call (/root/reference/src/cublas.cu:18-48, :72-83); so does this library (interpose.cpp).  Everything here
// is a development switch: parity tests and A/B tools that flip switches between calls of one process set
// OZIMMU_HIP_ENV_PER_CALL=1 before the library loads, which makes config() follow the environment on every use.
#pragma once
#include <cstddef>
#include <cstdint>

namespace ozhip {

struct Config {
  bool env_per_call = false;
  // slice GEMM kernel choice: OZIMMU_HIP_GEMM_KERNEL = wide | classic | k2 | x16 | k64 (unset: the policy in slice_gemm_launch.h)
  enum Kernel { AUTO = 0, WIDE, CLASSIC, K2, X16, K64 } gemm_kernel = AUTO;
  int paired_tile = -1;        // OZIMMU_HIP_PAIRED_TILE: 1 / 0 force the 16x16x64 tile function on / off (-1: policy)
  int k64_tile = -1;           // OZIMMU_HIP_K64_TILE: 1 / 0 force the 64-k-step 16x16x64 tile function on / off (-1: policy)
  int k64_breg = -1;           // OZIMMU_HIP_K64_BREG: 1 / 0 force the k64 tile's B fragments global -> VGPR / through LDS (-1: policy)
  bool fused_products = true;  // OZIMMU_HIP_FUSED_PRODUCTS=0: the real products of a small ZGEMM as separate launches
  int wide_small_rows = -1;    // OZIMMU_HIP_WIDE_SMALL_ROWS: rows of reduced-height tiles (measurement override)
  bool wide_static = false;    // OZIMMU_HIP_WIDE_STATIC=1: one tile per workgroup instead of persistent workgroups
  int wide_grid = 0;           // OZIMMU_HIP_WIDE_GRID: persistent workgroups of the wide kernel (tests: few, many tiles each)
  int xcds = 0;                // OZIMMU_HIP_XCDS: pretend the device has this many XCDs (tests of the tile partition)
  bool no_throttle = false;    // OZIMMU_HIP_NO_THROTTLE
  bool no_exp_reuse = false;   // OZIMMU_HIP_NO_EXP_REUSE: auto mode recomputes the row maxima in the GEMM
  int phase_min_kb = 32;       // OZIMMU_HIP_PHASE_MIN_KB: passes of at most this many k-blocks run without the phase hint
  int static_rounds = 20;      // OZIMMU_HIP_STATIC_ROUNDS: up to this many EXACT rounds of tiles run as a static grid instead of persistent workgroups (0: never)
  bool epi_overlap = true;     // OZIMMU_HIP_EPI_OVERLAP=0: the k64 register kernels keep the whole FP64 recombination behind their k loop
  int spec_claim_kb = 64;      // OZIMMU_HIP_SPEC_CLAIM_KB: k loops of at most this many k-blocks claim the next tile one tile ahead (0: never)
  bool no_phase_hint = false;  // OZIMMU_HIP_NO_PHASE_HINT
  bool batch_loop = false;     // OZIMMU_HIP_BATCH_LOOP: strided batches as a per-matrix loop
  size_t split_band_bytes = 0;                  // OZIMMU_HIP_SPLIT_BAND_BYTES
  size_t split_one_pass_bytes = 0;              // OZIMMU_HIP_SPLIT_ONE_PASS_BYTES
  size_t split_multi_bytes = (size_t)512 << 20; // OZIMMU_HIP_SPLIT_MULTI_BYTES
  size_t batch_workspace_bytes = 0;             // OZIMMU_HIP_BATCH_WORKSPACE_BYTES (0: default budget)
  int split_strip = 0;                          // OZIMMU_HIP_SPLIT_STRIP
  int split_resident = -1;                      // OZIMMU_HIP_SPLIT_RESIDENT: 0 never, 8 / 16 / 32 force the strip height (-1: policy)
  // test hooks (tests/test_gpu_robustness.py)
  int test_fail_launch = 0;       // OZIMMU_HIP_TEST_FAIL_LAUNCH=n: the n-th slice-GEMM launch of a call is rejected
  uint32_t test_exp_epoch = 0;    // OZIMMU_HIP_TEST_EXP_EPOCH: jump the exponent-word epoch close to its wrap-around
  bool test_no_stream_order = false; // OZIMMU_HIP_TEST_NO_STREAM_ORDER: drop the cross-stream ordering (negative test)
  bool forced_kernel() const { return gemm_kernel != AUTO; }
};

Config config(); // a snapshot (by value: see config.cpp)

// getenv calls made by this library so far (tests/test_interpose_cpu.py: an intercepted call reads the environment at most
// three times)
unsigned long long getenv_calls();
const char *counted_getenv(const char *name);

} // namespace ozhip
