// slice_gemm_one_launch.hip — split + slice GEMM of a SMALL problem in ONE launch (round 5).
//
// Reference path: mtk::ozimmu::gemm_int8<double>, /root/reference/src/gemm.cu:344-410 - split_int8 of A, split_int8 of B (two
// kernels + a device sync each, src/split.cu:193-261), then S(S+1)/2 cublasGemmEx calls and the accumulate / axby passes.  For
// the sizes the default OZIMMU_INTERCEPT_THRESHOLD_* = 1024 (src/handle.cu:25-30) sends here first, this library's two-launch
// form - the resident split (9-13 us visible at 1024^3) followed by the K-split slice GEMM (42 us) - pays a kernel boundary
// (drain + ramp-up) and the split's tail in front of a GEMM that rocBLAS runs in one 36 us kernel.
//
// One launch: the grid is the K-split kernel's (one 8-wave workgroup per 64 x 64 tile, at most one per CU, so every
// workgroup is resident from the start).  Phase 1: workgroup w cuts the 8-row strips w, w + grid, ... of op(A) and op(B)
// (split_resident.h: the strip lives in the registers of the 8 waves, one read of the operand) and publishes one READY word
// per strip: tag | 1, tag = the call's epoch << 11 in the handle's exponent-word buffer, which holds nothing but epoch-tagged
// words and is never zeroed per call (api.cpp: exp_words).  Phase 2: the workgroup waits for the 8 + 8 strips its tile reads
// and runs k2_tile (slice_gemm_k2_kernel.h) unchanged.
//
// Visibility without cache maintenance.  MI355X has one L2 per XCD, kept coherent with the others only at kernel boundaries
// (write-back at the end, invalidate at the start); inside a kernel an agent-scope release / acquire pair costs a write-back
// of the producer's whole L2 and an invalidate of the consumer's - measured here with the compiler's fences: 1024 x 1024 x 512
// 84 us instead of 37 (256 workgroups x a full-L2 operation each, and every invalidate throws away the panels the XCD's other
// workgroups are multiplying).  Instead:
//   * producers store slices and row scales WRITE-THROUGH (sc0 sc1, split_resident.h: WT): the data is in memory when the
//     store is acknowledged, so `s_waitcnt vmcnt(0)` + the workgroup barrier + the READY store order it in front of the word;
//   * a strip is a whole number of 128-byte lines of the planes (8 rows x 16 bytes per slice and k-half), and nobody reads a
//     line of the planes in this kernel before the strip's word is seen - so no L2 or L1 can hold a copy that predates the
//     write; copies left by EARLIER calls were dropped by the invalidate at this kernel's start;
//   * READY words are read and written at agent scope (sc1: served by memory, not by an XCD's L2).
// Consumers therefore read the planes through their L2 as the two-launch form does, and no cache is flushed or invalidated.
//
// No deadlock by construction: a workgroup that has polled `spin` times without seeing a strip cuts that strip ITSELF - the
// cut is a pure function of the operand, so a second writer stores the same bytes - and goes on.  If fewer CUs than
// workgroups are available (CU masks, a partition the topology probe does not know, another stream's kernels holding CUs) the
// launch degrades to "every workgroup cuts what it needs" instead of hanging.  The wait is bounded by what it waits FOR
// (ADVICE r5): a poll is an agent-scope load + s_sleep, ~1 us, and the owners' cuts take 10-14 us in all, so the default of
// 96 polls is several times the longest legitimate wait; and a workgroup that has timed out ONCE stops waiting - its owner
// is evidently not resident - and cuts every strip it still misses after a single look (the counter used to restart per
// strip: 16 strips x 20 000 polls = hundreds of ms on CUs other work was waiting for).  Worst case now: ~0.1 ms + 16 cuts.
//
// Results: bit-identical to the two-launch form (same cut, same tile function, same epilogue).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "config.h"
#include "kernel_policy.h"
#include "kernels.h"
#include "one_launch.h"
#include "slice_gemm_k2_kernel.h"
#include "split_resident.h"
#include "topology.h"

namespace ozhip {

struct OneLaunchArgs {
  SplitJob job[2];   // op(A), op(B)
  uint32_t nblk[2];  // 8-row strips of each view (rows padded to TILE_ROWS)
  int L;
  uint32_t *ready;   // one word per strip: A strips first
  uint32_t want;     // tag | 1
  uint32_t spin;     // polls before a workgroup cuts a missing strip itself
};

constexpr int OL_R = 8;     // strip height
constexpr int OL_UNITS = 2; // unit blocks (8 rows x 128 k) per wave: K <= 8 waves x 2 x 128 = 2048

// tile of workgroup `bid`: the mapping of k2_tile (slice_gemm_k2_kernel.h), which picks its tile by blockIdx.x
__device__ __forceinline__ void k2_tile_of_block(const SliceGemmArgs &p, uint32_t bid, uint32_t &tm, uint32_t &tn) {
  const uint32_t nb = p.tiles_m * p.tiles_n, nx = p.nxcd;
  const uint32_t xcd = bid % nx, idx = bid / nx, q = nb / nx, r = nb % nx;
  const uint32_t lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  constexpr uint32_t PH = 4;
  const uint32_t band_tiles = PH * p.tiles_n, nbands = (p.tiles_m + PH - 1u) / PH;
  uint32_t band = lid / band_tiles;
  if (band > nbands - 1) band = nbands - 1;
  const uint32_t rem = lid - band * band_tiles;
  const uint32_t h = (p.tiles_m - band * PH) < PH ? (p.tiles_m - band * PH) : PH;
  tn = rem / h;
  tm = band * PH + rem % h;
}

template <int S>
__global__ __launch_bounds__(512, 1) void split_gemm_k2_kernel(const SliceGemmArgs p, const OneLaunchArgs o) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned row_e[32];
  __shared__ unsigned wg_miss;
  double *const tile = reinterpret_cast<double *>(smem) + (size_t)(threadIdx.x >> 6) * RES_TILE_DOUBLES;
  // (scalars of `o` as locals: lambdas that capture the argument struct by reference make the compiler copy it to scratch)
  const uint32_t nblk0 = o.nblk[0], total = o.nblk[0] + o.nblk[1], want = o.want, spin = o.spin;
  uint32_t *const ready = o.ready;
  const int L = o.L;
  const SplitJob ja = o.job[0], jb = o.job[1];

  auto cut_strip = [&](uint32_t s) {
    const bool is_b = s >= nblk0;
    SplitJob j;
    j.v.in = is_b ? jb.v.in : ja.v.in;
    j.v.rows = is_b ? jb.v.rows : ja.v.rows;
    j.v.K = ja.v.K; // the host checks that the views agree
    j.v.stride_r = is_b ? jb.v.stride_r : ja.v.stride_r;
    j.v.stride_k = is_b ? jb.v.stride_k : ja.v.stride_k;
    j.planes = is_b ? jb.planes : ja.planes;
    j.max_exp = is_b ? jb.max_exp : ja.max_exp;
    j.in_stride = 0;
    j.exps = nullptr;
    const size_t ls = is_b ? s - nblk0 : s;
    if (j.v.stride_k < j.v.stride_r)
      split_resident_strip<OL_R, true, false, OL_UNITS, true>(j, S, L, ls, tile, row_e);
    else
      split_resident_strip<OL_R, false, false, OL_UNITS, true>(j, S, L, ls, tile, row_e);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's write-through stores have reached memory ...
    __syncthreads();                                 // ... and so have every thread's of the workgroup (row_e / tile are free again)
    if (threadIdx.x == 0) __hip_atomic_store(ready + s, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // ---- phase 1: this workgroup's share of the strips; phase 2: wait for the strips of its tile ---------------------------
  // (one loop with ONE call site of the cut: it is inlined in both layouts, ~5 000 instructions each)
  // workgroups beyond the tiles (the host launches as many as the stand-alone split would use, up to one per CU) only cut
  const uint32_t ntiles = p.tiles_m * p.tiles_n;
  const bool has_tile = blockIdx.x < ntiles;
  uint32_t tm = 0, tn = 0;
  if (has_tile) k2_tile_of_block(p, blockIdx.x, tm, tn);
  auto strip_of = [&](uint32_t i) { return i < 8u ? tm * 8u + i : nblk0 + tn * 8u + (i - 8u); };
  uint32_t s = blockIdx.x;
  bool own = s < total;
  uint32_t patience = spin; // polls before self-service; 1 after the first timeout
#pragma unroll 1
  for (;;) {
    if (!own) {
      if (!has_tile) return; // (uniform for the workgroup)
      if (threadIdx.x < 64) { // wave 0 polls: lane i < 16 watches strip i of the tile
        const uint32_t lane = threadIdx.x;
        const uint32_t *w = ready + strip_of(lane < 16u ? lane : 0u);
        bool ok = lane >= 16u;
        unsigned long long miss;
        uint32_t spins = 0;
        do {
          if (!ok) ok = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
          miss = __ballot(!ok);
          if (!miss) break;
          __builtin_amdgcn_s_sleep(2);
        } while (++spins < patience);
        if (lane == 0) wg_miss = (unsigned)miss;
      }
      __syncthreads();
      const unsigned miss = wg_miss;
      if (!miss) break;
      __syncthreads(); // everybody has read wg_miss before wave 0 writes the next round's
      s = strip_of((uint32_t)__ffs(miss) - 1u); // not seen in time: cut it here (same bytes as its owner writes)
      patience = 1u;
    }
    cut_strip(s);
    if (own) {
      s += gridDim.x;
      own = s < total;
    }
  }
  // (no acquire: see "Visibility" above; the copies below are volatile asm behind the barrier that broadcast the poll's verdict)

  k2_tile<S, 0, S>(p, smem);
}

// OZIMMU_HIP_ONE_LAUNCH: 0 never, 1 whenever the form applies (default), read like every development switch (config.h):
// once per process, or per call when the tests follow the environment.  OZIMMU_HIP_ONE_LAUNCH_SPIN: polls before self-service.
static long env_number(const char *name, long dflt) {
  const char *e = counted_getenv(name);
  return (e && *e) ? std::strtol(e, nullptr, 0) : dflt;
}
struct OneLaunchCfg {
  int mode;
  uint32_t spin;
};
static OneLaunchCfg one_launch_cfg() {
  auto read = []() { return OneLaunchCfg{(int)env_number("OZIMMU_HIP_ONE_LAUNCH", 1), (uint32_t)std::max(1l, env_number("OZIMMU_HIP_ONE_LAUNCH_SPIN", 96))}; };
  if (config().env_per_call) return read();
  static const OneLaunchCfg c = read();
  return c;
}

template <int S>
static hipError_t launch_S(const SliceGemmArgs &a, const OneLaunchArgs &o, uint32_t grid, hipStream_t stream) {
  using Cfg = K2Cfg<S, 0, S>;
  static_assert(Cfg::ok, "the K-split tile must exist for this mode");
  // the split's transpose tiles (k-contiguous operands) live in the GEMM's staging buffers
  constexpr size_t TILES = sizeof(double) * RES_MAX_WAVES * RES_TILE_DOUBLES;
  constexpr size_t LDS = Cfg::LDS > TILES ? Cfg::LDS : TILES;
  static std::atomic<uint64_t> attr_done{0};
  const uint64_t bit = a.device < 64 ? (1ull << a.device) : 0ull;
  if (!(bit && (attr_done.load(std::memory_order_acquire) & bit))) {
    const hipError_t e = hipFuncSetAttribute((const void *)split_gemm_k2_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) return e;
    if (bit) attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((split_gemm_k2_kernel<S>), dim3(grid), dim3(512), LDS, stream, a, o);
  return hipGetLastError();
}

size_t one_launch_ready_words(size_t m, size_t n) { return (row_blocks_padded(m) + row_blocks_padded(n)) * (32 / OL_R); }

hipError_t launch_split_gemm_one(int S, const SliceGemmArgs &a_in, const SplitJob job[2], int L, uint32_t *ready, uint32_t tag,
                                 hipStream_t stream) {
  const OneLaunchCfg oc = one_launch_cfg();
  if (oc.mode == 0 || S < 3 || S > 9 || S * L > 64) return hipErrorNotSupported;
  if (a_in.batch > 1 || a_in.kb0 != 0 || a_in.kb1 != a_in.KB || !a_in.final || a_in.acc_in || a_in.cplx) return hipErrorNotSupported;
  // a strip of 8 rows x K lives in 8 waves x OL_UNITS unit blocks of 128 k: K <= 2048, where the resident split runs by policy
  if (job[0].v.K != job[1].v.K || job[0].v.K > (size_t)RES_MAX_WAVES * OL_UNITS * (1024 / OL_R))
    return hipErrorNotSupported;
  const Topology topo = topology(a_in.device);
  PassTraits t;
  if (!slice_gemm_traits(S, 0, &t) || !t.k2_ok) return hipErrorNotSupported;
  PolicyInput in;
  in.M = a_in.M;
  in.N = a_in.N;
  in.nkb = a_in.kb1 - a_in.kb0;
  in.batch = 1;
  const Prediction r = policy_predict(t, in, topo, config());
  if (r.pick != Pick::K2) return hipErrorNotSupported;
  SliceGemmArgs a = a_in;
  a.nxcd = (uint32_t)topo.xcds;
  a.tiles_m = (a.M + 63) / 64;
  a.tiles_n = (a.N + 63) / 64;
  // every workgroup must be resident while it waits: one 512-thread workgroup per CU
  if ((uint64_t)a.tiles_m * a.tiles_n == 0 || (uint64_t)a.tiles_m * a.tiles_n > (uint64_t)topo.cus) return hipErrorNotSupported;
  OneLaunchArgs o{};
  o.job[0] = job[0];
  o.job[1] = job[1];
  o.nblk[0] = (uint32_t)(row_blocks_padded(job[0].v.rows) * (32 / OL_R));
  o.nblk[1] = (uint32_t)(row_blocks_padded(job[1].v.rows) * (32 / OL_R));
  o.L = L;
  o.ready = ready;
  o.want = tag | 1u;
  o.spin = oc.spin;
  // one workgroup per tile, plus split-only workgroups up to one per strip / per CU (every workgroup must be resident)
  const uint32_t ntiles = a.tiles_m * a.tiles_n;
  const uint32_t grid = std::max(ntiles, std::min((uint32_t)topo.cus, o.nblk[0] + o.nblk[1]));
  note_pick(0, (int)Pick::K2 + 16); // diagnostics (ozimmu_hip_last_kernel): "k2_one_launch"
  switch (S) {
  case 3: return launch_S<3>(a, o, grid, stream);
  case 4: return launch_S<4>(a, o, grid, stream);
  case 5: return launch_S<5>(a, o, grid, stream);
  case 6: return launch_S<6>(a, o, grid, stream);
  case 7: return launch_S<7>(a, o, grid, stream);
  case 8: return launch_S<8>(a, o, grid, stream);
  case 9: return launch_S<9>(a, o, grid, stream);
  default: return hipErrorNotSupported;
  }
}

} // namespace ozhip
