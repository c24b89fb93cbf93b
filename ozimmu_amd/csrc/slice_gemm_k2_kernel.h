// slice_gemm_k2_kernel.h — the slice GEMM for problems that do not fill the chip: 64x64 tiles, K split inside the
// workgroup.
//
// When a launch has no more 64x64 tiles than the device has CUs (1024^3: 256 tiles on 256 CUs), the classic kernel
// runs ONE 4-wave workgroup per CU: one wave per SIMD, so every LDS read burst and every barrier is exposed and the
// matrix pipe idles more than half of the time (48.6 us at 1024^3 S=9 against ~20 us of MFMA issue).  INT32 diagonal
// sums are exact, so K can be split without changing a bit of the result: here a workgroup has EIGHT waves in two
// groups; group g accumulates its half of the k-blocks of the same 64x64 tile from its own pair of LDS buffers, group
// 1 hands its INT32 accumulators to group 0 through LDS after the loop (147 KiB for ND = 9, ~1 us), and group 0 runs
// the shared epilogue.  Each SIMD holds one wave of either group.
//
// The hardware barrier is shared by the groups, so their k-steps are offset by half a step around it:
//   group 0:  B_t | copy(t+1) | read fragments(t) | 45 MFMAs(t)                                    | B_t+1
//   group 1:  B_t | copy(t+1) | second half MFMAs(t-1) | read fragments(t) | first half MFMAs(t)   | B_t+1
// while one group waits for LDS the other one issues MFMAs.  A wave reads all 2*SL fragments of a step into registers
// before its first MFMA (16*ND + 8*SL <= 224 registers of the 256 a wave has at two waves per SIMD), so the buffer of
// step t is free for the copy of step t+2 one barrier later, and every copy has a full step to land.
#pragma once
#include "slice_gemm_w_kernel.h" // glds16: the LDS-DMA copy as inline asm

namespace ozhip {

template <int S, int D0, int ND>
struct K2Cfg {
  static constexpr int SL = (D0 + ND < S) ? (D0 + ND) : S;
  static constexpr size_t STAGE = 4 * SL * FRAG_BYTES;     // 2 A row-blocks + 2 B row-blocks of one k-step
  static constexpr size_t REDUCE = (size_t)ND * 16 * 1024; // 4 waves x ND x 16 registers x 64 lanes x 4 bytes
  static constexpr size_t LDS = 4 * STAGE > REDUCE ? 4 * STAGE : REDUCE;
  static constexpr bool ok = LDS <= 160 * 1024 && 16 * ND + 8 * SL <= 224;
};

// one 64x64 tile (picked by blockIdx.x) of the product described by p
template <int S, int D0, int ND>
__device__ __forceinline__ void k2_tile(const SliceGemmArgs &p, char *smem) {
  using Cfg = K2Cfg<S, D0, ND>;
  constexpr int SL = Cfg::SL;
  constexpr int STAGE = (int)Cfg::STAGE;
  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;
  const int wm = wave & 1, wn = wave >> 1;

  // workgroup -> tile: a contiguous run of ids per XCD (hardware places workgroup b on XCD b % nxcd), bands of 4 tile rows
  // with the columns outermost, so that the 32 workgroups of an XCD form a 4 x 8 patch sharing 12 panels in its L2
  const uint32_t nb = p.tiles_m * p.tiles_n;
  uint32_t lid;
  {
    const uint32_t nx = p.nxcd; // XCDs of the device (topology.h)
    const uint32_t bid = blockIdx.x, xcd = bid % nx, idx = bid / nx, q = nb / nx, r = nb % nx;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  uint32_t tm, tn;
  {
    constexpr uint32_t PH = 4;
    const uint32_t band_tiles = PH * p.tiles_n, nbands = (p.tiles_m + PH - 1u) / PH;
    uint32_t band = lid / band_tiles;
    if (band > nbands - 1) band = nbands - 1;
    const uint32_t rem = lid - band * band_tiles;
    const uint32_t h = (p.tiles_m - band * PH) < PH ? (p.tiles_m - band * PH) : PH;
    tn = rem / h;
    tm = band * PH + rem % h;
  }

  // this group's k-blocks: [kbeg, kbeg + nkg); both groups run `nit` barrier rounds
  const uint32_t nk = p.kb1 - p.kb0, nit = (nk + 1u) >> 1;
  const uint32_t kbeg = p.kb0 + (grp ? nit : 0u), nkg = grp ? nk - nit : nit;

  // staging: wave w of a group copies row-block w of {A0, A1, B0, B1}, SL fragment blocks per k-step
  char *const sg = smem + grp * (2 * STAGE);
  const int8_t *src_u = (wave < 2 ? p.a_planes + (size_t)(2 * tm + wave) * p.KB * (size_t)(S * FRAG_BYTES)
                                  : p.b_planes + (size_t)(2 * tn + (wave - 2)) * p.KB * (size_t)(S * FRAG_BYTES)) +
                        (size_t)kbeg * (S * FRAG_BYTES);
  const uint32_t lane_off = (uint32_t)lane * 16u;
  // Copies as inline asm (glds16, slice_gemm_w_kernel.h): the compiler counts the LDS-DMA builtin on lgkmcnt as well and
  // then waits for ALL fragment reads (lgkmcnt(0)) before a step's first MFMA; with the copies invisible to it the
  // fragment waits are counted and the MFMAs start when the first fragments arrive.  The copies are ordered by the
  // explicit vmcnt(0) in front of every barrier below.
  const uint32_t lds_wave = (uint32_t)(size_t)((OZ_AS3 char *)sg) + (uint32_t)wave * (SL * FRAG_BYTES);
  auto stage = [&](uint32_t step) {
    const int8_t *sb = uniform_ptr(src_u + (size_t)step * (S * FRAG_BYTES));
    const uint32_t lds_buf = lds_wave + (step & 1u) * STAGE;
    static_for<SL>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int G = 4, g0 = s / G * G; // the immediate offset (0 .. 3072) advances the LDS address too
      glds16<(s % G) * FRAG_BYTES>(sb + g0 * FRAG_BYTES, lane_off, lds_buf, (uint32_t)(g0 * FRAG_BYTES));
    });
  };
  auto sync = [&]() { // own copies landed, then everybody's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  v16i acc[ND];
#pragma unroll
  for (int d = 0; d < ND; d++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[d][r] = 0;

  // the slice pairs of this pass in (i, j) order; group 1 cuts the list in two around the barrier
  constexpr int NPAIR = []() {
    int c = 0;
    for (int i = 0; i < SL; i++)
      for (int j = 0; j < SL; j++) c += (i + j >= D0 && i + j < D0 + ND && i + j <= S - 1) ? 1 : 0;
    return c;
  }();
  constexpr int CUT = NPAIR / 2;
  v4i bf[SL], af[SL];
  auto read_fragments = [&](uint32_t step) {
    const char *la = sg + (step & 1u) * STAGE + wm * (SL * FRAG_BYTES) + lane * 16;
    const char *lb = sg + (step & 1u) * STAGE + (2 + wn) * (SL * FRAG_BYTES) + lane * 16;
#pragma unroll
    for (int j = 0; j < SL; j++) bf[j] = *(const v4i *)(lb + j * FRAG_BYTES);
#pragma unroll
    for (int i = 0; i < SL; i++) af[i] = *(const v4i *)(la + i * FRAG_BYTES);
  };
  auto mfmas = [&](auto lo_tag, auto hi_tag) { // pairs [lo, hi) of the list
    constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
    int c = 0;
#pragma unroll
    for (int i = 0; i < SL; i++)
#pragma unroll
      for (int j = 0; j < SL; j++) {
        const int d = i + j;
        if (d >= D0 && d < D0 + ND && d <= S - 1) {
          if (c >= LO && c < HI) acc[d - D0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[j], af[i], acc[d - D0], 0, 0, 0);
          c++;
        }
      }
  };
  using I0 = std::integral_constant<int, 0>;
  using IC = std::integral_constant<int, CUT>;
  using IN = std::integral_constant<int, NPAIR>;

  // The two groups run separate loops (the barriers pair up by count: `nit` each) so that no MFMA sits under a
  // condition: accumulators that flow through conditional blocks cost the compiler hundreds of copies and spills.
  // The scheduling barriers keep the reads of step t behind the MFMAs that still use the registers of step t-1.
  if (nkg) stage(0);
  if (grp == 0) {
    for (uint32_t t = 0; t < nit; t++) {
      sync(); // copies of step t landed (vmcnt(0) precedes the barrier); step t-1 is done with the other buffer
      if (t + 1 < nit) stage(t + 1);
      read_fragments(t);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(I0{}, IN{});
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    if (nkg) {
      sync();
      if (1 < nkg) stage(1);
      read_fragments(0);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(I0{}, IC{});
      __builtin_amdgcn_sched_barrier(0);
    }
    for (uint32_t t = 1; t < nkg; t++) {
      sync(); // step t landed; the fragments of step t-1 are in registers: its buffer takes step t+1
      if (t + 1 < nkg) stage(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(IC{}, IN{});
      __builtin_amdgcn_sched_barrier(0);
      read_fragments(t);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(I0{}, IC{});
      __builtin_amdgcn_sched_barrier(0);
    }
    if (nkg) mfmas(IC{}, IN{});
    if (nkg < nit) sync(); // odd number of k-blocks: group 0 ran one more round
  }

  // ---- group 1 -> group 0: INT32 accumulators through LDS (the staging buffers are dead) -------------------------
  __syncthreads();
  char *red = smem + (size_t)wave * (ND * 4096) + lane * 16;
  if (grp) {
#pragma unroll
    for (int d = 0; d < ND; d++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        *(v4i *)(red + (d * 4 + q) * 1024) = v4i{acc[d][4 * q], acc[d][4 * q + 1], acc[d][4 * q + 2], acc[d][4 * q + 3]};
  }
  __syncthreads();
  if (grp) return; // (the fused-products kernel meets group 0 again at its next barrier)
#pragma unroll
  for (int d = 0; d < ND; d++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const v4i o = *(const v4i *)(red + (d * 4 + q) * 1024);
#pragma unroll
      for (int e = 0; e < 4; e++) acc[d][4 * q + e] += o[e];
    }
    // one diagonal at a time, the sums pinned here: left alone the optimiser sinks the adds into the epilogue (next to the
    // conversions that consume them) and keeps all 4 * ND LDS reads alive until then - 144 registers at ND = 9, 180 bytes of
    // scratch per lane, in the build WITHOUT the dump hook only, which is the one that ships
    asm volatile("" : "+v"(acc[d]));
  }

  recombine_and_store<D0, ND, 1>(p, [&](int, int d, int r) { return acc[d][r]; }, tm * 64 + wm * 32 + (lane & 31),
                                 tn * 64 + wn * 32 + 4 * (lane >> 5));
}

template <int S, int D0, int ND>
__global__ __launch_bounds__(512, 1) void slice_gemm_k2_kernel(const SliceGemmArgs p_in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const SliceGemmArgs p = batch_view(p_in);
  k2_tile<S, D0, ND>(p, smem);
}

// Up to four products that accumulate into the same C one after the other -- the real products (Im,Im) (Re,Re) (Im,Re)
// (Re,Im) of a ZGEMM -- in ONE launch: a workgroup runs its tile of every product in the given order, so each element
// of C sees the same sequence of updates as with four launches (bit-identical), and a small ZGEMM saves three launch /
// ramp-up / drain rounds of a ~45 us kernel.
template <int S, int D0, int ND>
__global__ __launch_bounds__(512, 1) void slice_gemm_k2_fused_kernel(const SliceGemmMulti m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#pragma unroll 1
  for (int i = 0; i < m.count; i++) {
    __syncthreads(); // group 0 is done with the previous product's LDS (accumulator hand-over) and its epilogue
    const SliceGemmArgs p = batch_view(m.g[i]);
    k2_tile<S, D0, ND>(p, smem);
  }
}

} // namespace ozhip
