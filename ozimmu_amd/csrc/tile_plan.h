// tile_plan.h — host-side partition of the rows of C into full-height and reduced tiles of the wide slice GEMM.
// Plain C++ (no device code): slice_gemm_launch.h plans with it on every call, so it must cost microseconds whatever the
// shape is (a 32768 x 32768 output has 131072 tiles of 64 x 128: the first version simulated the dispatch tile round by
// tile round for every candidate partition, 12 ms of host time per call - invisible while calls queue behind each
// other, a third of the call once the caller synchronises between calls; tools/gap_probe.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>

#include "config.h"

namespace ozhip {

// Rows of full-height (WA blocks) and reduced (WA-1 blocks) tiles that cover `rows32` 32-row blocks with the smallest
// makespan on `ncu` CUs.  Workgroups are dispatched in order, big tiles first, each to the first CU that frees up; a
// tile of h blocks costs h + OVH (k loop + prologue/epilogue).
struct WidePlan {
  uint32_t n_big = 0, n_small = 0;
  double makespan = 0; // in block units per CU
  double efficiency = 0;
};

// Makespan of that greedy dispatch in closed form.  After the big tiles, r = nbig % ncu CUs free up at (q + 1) cbig and
// the other ncu - r at q cbig (q = nbig / ncu).  Equal jobs on a greedy list schedule finish on the grid points
// {free time + i csmall}, taken in increasing order: the n-th small tile finishes at the smallest T with
//     (ncu - r) * floor((T - q cbig) / csmall) + r * floor((T - (q + 1) cbig) / csmall) >= n   (negative terms: 0),
// and T lies on one of the two grids, so two binary searches over the step count find it.
inline double simulate_rounds(uint64_t nbig, double cbig, uint64_t nsmall, double csmall, int ncu) {
  const uint64_t q = nbig / (uint64_t)ncu, r = nbig % (uint64_t)ncu;
  const double ta = (double)q * cbig, tb = ta + cbig;
  const double big_last = nbig == 0 ? 0.0 : (r ? tb : ta);
  if (nsmall == 0) return big_last;
  const uint64_t ca = (uint64_t)ncu - r;
  auto count = [&](double T) {
    uint64_t c = 0;
    if (T >= ta) c += ca * (uint64_t)((T - ta) / csmall + 1e-9);
    if (r && T >= tb) c += r * (uint64_t)((T - tb) / csmall + 1e-9);
    return c;
  };
  auto first_on_grid = [&](double t0) { // smallest t0 + i csmall (i >= 1) by which nsmall tiles have finished
    uint64_t lo = 1, hi = nsmall + 1;   // i = nsmall always suffices (ca >= 1)
    while (lo < hi) {
      const uint64_t mid = lo + (hi - lo) / 2;
      if (count(t0 + (double)mid * csmall) >= nsmall) hi = mid; else lo = mid + 1;
    }
    return t0 + (double)lo * csmall;
  };
  double T = first_on_grid(ta);
  if (r) T = std::min(T, first_on_grid(tb));
  return std::max(T, big_last);
}

#ifdef OZIMMU_HIP_TEST_HOOKS
// the dispatch simulated literally on (time -> CU count) buckets: what the closed form must reproduce (tests/test_abi.py)
inline double simulate_rounds_reference(uint64_t nbig, double cbig, uint64_t nsmall, double csmall, int ncu) {
  std::map<double, uint64_t> free_at;
  free_at[0.0] = (uint64_t)ncu;
  double last = 0;
  auto run = [&](uint64_t n, double c) {
    while (n) {
      auto it = free_at.begin();
      const uint64_t take = n < it->second ? n : it->second;
      const double t = it->first + c;
      it->second -= take;
      if (it->second == 0) free_at.erase(it);
      free_at[t] += take;
      if (t > last) last = t;
      n -= take;
    }
  };
  run(nbig, cbig);
  run(nsmall, csmall);
  return last;
}
#endif

template <class Sim>
inline WidePlan plan_wide_with(uint32_t M, uint32_t N, int WA, int ncu, Sim sim) {
  constexpr double OVH = 0.06;
  const uint32_t rows32 = (M + 31) / 32, tn = (N + 127) / 128;
  WidePlan best;
  const uint32_t max_small = WA > 1 ? (rows32 + (WA - 2)) / (WA - 1) : 0;
  for (uint32_t n2 = 0; n2 <= max_small; n2++) {
    const uint32_t covered = (uint32_t)(WA - 1) * n2;
    const uint32_t n3 = covered >= rows32 ? 0 : (rows32 - covered + WA - 1) / WA;
    const double t = sim((uint64_t)n3 * tn, WA + OVH, (uint64_t)n2 * tn, WA - 1 + OVH, ncu);
    if (best.makespan == 0 || t < best.makespan - 1e-9) {
      best.n_big = n3;
      best.n_small = n2;
      best.makespan = t;
    }
    if (n3 == 0) break;
  }
  if (config().wide_small_rows >= 0) { // measurement override: rows of reduced-height tiles
    const uint32_t n2 = std::min<uint32_t>((uint32_t)config().wide_small_rows, max_small);
    const uint32_t covered = (uint32_t)(WA - 1) * n2;
    best.n_small = n2;
    best.n_big = covered >= rows32 ? 0 : (rows32 - covered + WA - 1) / WA;
    best.makespan = sim((uint64_t)best.n_big * tn, WA + OVH, (uint64_t)n2 * tn, WA - 1 + OVH, ncu);
  }
  best.efficiency = (double)rows32 * tn / (best.makespan * ncu);
  return best;
}
inline WidePlan plan_wide(uint32_t M, uint32_t N, int WA, int ncu) { return plan_wide_with(M, N, WA, ncu, simulate_rounds); }

// The same search with the tile costs of the caller's model (kernel_policy.cpp): a tile of h blocks costs cost(h) in the
// caller's unit (microseconds); makespan in that unit.
template <class Cost>
inline WidePlan plan_wide_costs(uint32_t M, uint32_t N, int WA, int ncu, Cost cost) {
  const uint32_t rows32 = (M + 31) / 32, tn = (N + 127) / 128;
  const double cbig = cost(WA), csmall = WA > 1 ? cost(WA - 1) : cbig;
  WidePlan best;
  const uint32_t max_small = WA > 1 ? (rows32 + (WA - 2)) / (WA - 1) : 0;
  auto eval = [&](uint32_t n2) {
    const uint32_t covered = (uint32_t)(WA - 1) * n2;
    const uint32_t n3 = covered >= rows32 ? 0 : (rows32 - covered + WA - 1) / WA;
    const double t = simulate_rounds((uint64_t)n3 * tn, cbig, (uint64_t)n2 * tn, csmall, ncu);
    if (best.makespan == 0 || t < best.makespan - 1e-9) {
      best.n_big = n3;
      best.n_small = n2;
      best.makespan = t;
    }
    return n3;
  };
  if (config().wide_small_rows >= 0) { // measurement override: rows of reduced-height tiles
    eval(std::min<uint32_t>((uint32_t)config().wide_small_rows, max_small));
  } else {
    for (uint32_t n2 = 0; n2 <= max_small; n2++)
      if (eval(n2) == 0) break;
  }
  best.efficiency = cbig > 0 ? (double)rows32 * tn * (cbig / WA) / (best.makespan * ncu) : 0;
  return best;
}

} // namespace ozhip
