// slice_gemm_pp_kernel.h — "ping-pong" form of the fused INT8 slice GEMM: ONE 8-wave workgroup per CU whose two
// wave groups alternate explicitly between computing and loading.
//
// Why (measured on the 4-wave kernel, tools/gemm_ablate.hip): per k-step a wave spends ~1500 cycles issuing its 45
// MFMAs and ~2400 cycles in everything else (LDS-DMA issue ~1250, fragment reads ~430, two barriers, waits).  With two
// independent workgroups per CU the other workgroup's wave covers that gap only when the pair happens to run in
// anti-phase; the pairing drifts, and the non-MFMA part is the longer one anyway, so the matrix pipes sit at ~72 %.
// Here the alternation is structural:
//
//   * workgroup = 8 waves = groups G0 (waves 0-3) and G1 (waves 4-7); wave w and wave w+4 share a SIMD;
//   * tile = 64 (M) x 128 (N): both groups use the same 2 A row-blocks, G0 the B row-blocks 0,1, G1 the B row-blocks
//     2,3 -> 6*S KiB per stage instead of 8*S KiB for two 64x64 workgroups (-25 % staged bytes per MFMA);
//   * time runs in half-steps separated by one s_barrier.  In half-step h the group (h & 1) COMPUTES its k-step from
//     registers (45 MFMAs per wave) while the other group LOADS: it reads the 2*S fragments of its next k-step from
//     LDS into registers and issues its share of the LDS-DMA copies for the stage after that.  G0 computes k-step t in
//     half-step 2t, G1 in half-step 2t+1;
//   * two stage buffers.  A buffer slot is refilled in the load phase after its last reader: G0 copies A0,A1,B0 of
//     stage t+2 in half-step 2t+1 (G1 read A of stage t in 2t), G1 copies B1 of stage t+2 and B2,B3 of stage t+1 in
//     half-step 2t -- 3*S blocks per group, so the two load phases cost the same.  A wave waits for its own copies
//     (vmcnt(0)) at the end of its next compute phase, one barrier before anybody reads them.
//
// Same arithmetic as slice_gemm_kernel (same accumulators, same epilogue): results are bit-identical.
#pragma once
#include "slice_gemm_kernel.h"

namespace ozhip {

template <int S, int VAR = 0>
__global__ __launch_bounds__(512, 2) void slice_gemm_pp_kernel(const SliceGemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__) // the host pass only needs the stub (arrays of buffer resources do not parse there)
  constexpr int SL = S, ND = S, D0 = 0;
  constexpr int STAGE_BYTES = 6 * SL * FRAG_BYTES; // slots: A0 A1 B0 B1 B2 B3
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = wave >> 2, w = wave & 3, wm = w & 1, wn = w >> 1;

  // ---- workgroup -> tile: XCD-contiguous ids, 8 (M) x 4 (N) tiles = 512 x 512 outputs per XCD patch ------------
  const uint32_t nb = p.tiles_m * p.tiles_n;
  uint32_t lid;
  {
    const uint32_t bid = blockIdx.x, xcd = bid & 7u, idx = bid >> 3, q = nb >> 3, r = nb & 7u;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  uint32_t tm, tn;
  {
    const uint32_t band_tiles = 8u * p.tiles_n, nbands = (p.tiles_m + 7u) >> 3;
    uint32_t band = lid / band_tiles;
    if (band > nbands - 1) band = nbands - 1;
    const uint32_t rem = lid - band * band_tiles;
    const uint32_t h = (p.tiles_m - band * 8u) < 8u ? (p.tiles_m - band * 8u) : 8u;
    tn = rem / h;
    tm = band * 8u + rem % h;
  }

  // ---- staging shares ---------------------------------------------------------------------------------------------
  // A stage holds 6*S fragment blocks, numbered slot * S + s with slots A0 A1 B0 | B1 B2 B3.  G0 copies the first 3*S
  // (two stages ahead, in its load half-step), G1 the last 3*S (B1 two stages ahead -- its buffer is free as soon as
  // G0 has read it -- and B2, B3 one stage ahead), so both groups issue the same number of copies per k-step.
  // Wave w of a group takes the blocks q = w, w + 4, ... of its 3*S.
  constexpr int NSH = (3 * S + 3) / 4;
  const int8_t *a_src = p.a_planes + (size_t)(2u * tm) * p.KB * (size_t)(S * FRAG_BYTES);
  const int8_t *b_src = p.b_planes + (size_t)(4u * tn) * p.KB * (size_t)(S * FRAG_BYTES);
  const size_t rb_stride = (size_t)p.KB * (size_t)(S * FRAG_BYTES);
  const uint32_t lane_off = (uint32_t)lane * 16u;
  auto block_src = [&](int qg) -> const int8_t * { // global address of block qg of the stage of k-block 0
    const int slot = qg / S, s = qg - slot * S;
    return (slot < 2 ? a_src + slot * rb_stride : b_src + (slot - 2) * rb_stride) + s * FRAG_BYTES;
  };
  const int8_t *sh_src[NSH];
#pragma unroll
  for (int i = 0; i < NSH; i++) sh_src[i] = block_src(3 * S * g + (w + 4 * i < 3 * S ? w + 4 * i : 0));
  // VAR & 2: MUBUF form of the copies (one buffer resource per share block over its row-block, always < 4 GiB):
  // SGPR resource + ONE lane-offset VGPR + SGPR byte offset -- cheaper to issue under MFMA load
  __amdgpu_buffer_rsrc_t sh_rsrc[NSH];
#pragma unroll
  for (int i = 0; i < NSH; i++)
    sh_rsrc[i] = __builtin_amdgcn_make_buffer_rsrc((void *)sh_src[i], 0, (int)(p.KB * (uint32_t)(S * FRAG_BYTES)), 0x00020000);
  auto copy_share = [&](int i, uint32_t kb, int buf, int qg) {
    if constexpr ((VAR & 2) != 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(sh_rsrc[i], (OZ_AS3 void *)(smem + buf * STAGE_BYTES + qg * FRAG_BYTES), 16,
                                               lane_off, kb * (uint32_t)(S * FRAG_BYTES), 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const OZ_AS1 void *)(sh_src[i] + (size_t)kb * (S * FRAG_BYTES) + lane_off),
                                       (OZ_AS3 void *)(smem + buf * STAGE_BYTES + qg * FRAG_BYTES), 16, 0, OZ_GLDS_AUX);
  };
  auto copy_block = [&](const int8_t *gsrc, uint32_t kb, int buf, int qg) {
    __builtin_amdgcn_global_load_lds((const OZ_AS1 void *)(gsrc + (size_t)kb * (S * FRAG_BYTES) + lane_off),
                                     (OZ_AS3 void *)(smem + buf * STAGE_BYTES + qg * FRAG_BYTES), 16, 0, OZ_GLDS_AUX);
  };

  v16i acc[ND];
#pragma unroll
  for (int d = 0; d < ND; d++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[d][r] = 0;

  // ---- circular K with the per-XCD phase hint (see slice_gemm_kernel.h) ------------------------------------------
  const uint32_t nk = p.kb1 - p.kb0;
  uint32_t *phase = p.phase ? p.phase + 64u * (blockIdx.x & 7u) : nullptr;
  uint32_t koff = 0;
  if (phase && nk > 1) {
    if (threadIdx.x == 0)
      *(volatile uint32_t *)smem = __hip_atomic_load(phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    koff = (*(volatile uint32_t *)smem + 2u) % nk;
    __syncthreads();
  }
  koff = __builtin_amdgcn_readfirstlane(koff);
  auto kblock = [&](uint32_t t) { // k-block of the t-th k-step of this workgroup
    const uint32_t x = koff + t;
    return p.kb0 + (x >= nk ? x - nk : x);
  };

  const char *la0 = smem + wm * (SL * FRAG_BYTES) + lane * 16;               // A slot wm
  const char *lb0 = smem + (2 + 2 * g + wn) * (SL * FRAG_BYTES) + lane * 16; // B slot of this group
  v4i bf[SL], af[SL];
  auto read_frags = [&](int buf) {
#pragma unroll
    for (int j = 0; j < SL; j++) bf[j] = *(const v4i *)(lb0 + buf * STAGE_BYTES + j * FRAG_BYTES);
#pragma unroll
    for (int i = 0; i < SL; i++) af[i] = *(const v4i *)(la0 + buf * STAGE_BYTES + i * FRAG_BYTES);
  };
  auto compute = [&]() {
#pragma unroll
    for (int i = 0; i < SL; i++)
#pragma unroll
      for (int j = 0; j < SL; j++)
        if (i + j <= S - 1) acc[i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[j], af[i], acc[i + j], 0, 0, 0);
  };
  auto half_barrier = [&]() { // raw barrier: __syncthreads() would also drain vmcnt(0) of the loading group
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  auto publish = [&](uint32_t t) {
    if (phase && threadIdx.x == 0)
      __hip_atomic_store(phase, koff + t >= nk ? koff + t - nk : koff + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (nk) {
    // prologue, all 8 waves: stage 0 complete, and A0 A1 B0 B1 of stage 1 (G1 copies B2, B3 of stage 1 at h = 0)
    for (int qq = wave; qq < 10 * S; qq += 8) {
      const int st = qq >= 6 * S, qg = qq - st * 6 * S;
      if (!st || nk > 1) copy_block(block_src(qg), kblock(st), st, qg);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    half_barrier();
    // The two groups run separate loops (identical barrier counts): keeping them apart keeps the register
    // allocation of each simple -- one loop with `if (g == 0) compute else load` spilled 392 VGPRs.
    if (g == 0) {
      read_frags(0); // half-step -1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      half_barrier();
      for (uint32_t t = 0; t < nk; t++) {
        const int cur = t & 1;
        const bool tr = (VAR & 1) != 0 && p.trace && t >= 64 && t < 80 && lane == 0 && blockIdx.x - p.trace_block0 < 64;
        // stamps go to spare LDS (a global store would sit in vmcnt and distort the waits) and are dumped at the end
        unsigned long long *trp = (unsigned long long *)(smem + 2 * STAGE_BYTES) + (size_t)(wave * 16 + (t & 15)) * 8;
        if (tr) trp[0] = clock64();
        compute();                                       // half-step 2t: k-step t
        if (tr) trp[1] = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my copies of stage t+1 (issued one half-step ago) landed
        publish(t);
        if (tr) trp[2] = clock64();
        half_barrier();
        if (tr) trp[3] = clock64();
        // half-step 2t+1: first the copies (A0 A1 B0 of stage t+2 into the buffer stage t just left) so that they
        // get the whole half-step plus the next compute phase to land, then the fragments of k-step t+1
        if (t + 2 < nk) {
          const uint32_t kb2 = kblock(t + 2);
#pragma unroll
          for (int i = 0; i < NSH; i++)
            if (w + 4 * i < 3 * S) copy_share(i, kb2, cur, w + 4 * i);
        }
        if (tr) trp[4] = clock64();
        if (t + 1 < nk) read_frags(cur ^ 1);
        if (tr) trp[5] = clock64();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tr) trp[6] = clock64();
        half_barrier();
      }
    } else {
      half_barrier(); // half-step -1: nothing to do
      for (uint32_t t = 0; t < nk; t++) {
        const int cur = t & 1;
        const bool tr = (VAR & 1) != 0 && p.trace && t >= 64 && t < 80 && lane == 0 && blockIdx.x - p.trace_block0 < 64;
        // stamps go to spare LDS (a global store would sit in vmcnt and distort the waits) and are dumped at the end
        unsigned long long *trp = (unsigned long long *)(smem + 2 * STAGE_BYTES) + (size_t)(wave * 16 + (t & 15)) * 8;
        if (tr) trp[0] = clock64();
        // half-step 2t: copies first (see G0), then the fragments of k-step t
        const uint32_t kb1 = kblock(t + 1), kb2 = kblock(t + 2);
#pragma unroll
        for (int i = 0; i < NSH; i++) {
          const int q = w + 4 * i;
          if (q < S) { // B1 of stage t+2 (G0 read B1 of stage t one half-step ago)
            if (t + 2 < nk) copy_share(i, kb2, cur, 3 * S + q);
          } else if (q < 3 * S) { // B2, B3 of stage t+1
            if (t + 1 < nk) copy_share(i, kb1, cur ^ 1, 3 * S + q);
          }
        }
        if (tr) trp[1] = clock64();
        read_frags(cur);
        if (tr) trp[2] = clock64();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tr) trp[3] = clock64();
        half_barrier();
        if (tr) trp[4] = clock64();
        compute(); // half-step 2t+1: k-step t
        if (tr) trp[5] = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tr) trp[6] = clock64();
        half_barrier();
      }
    }
  }

  if constexpr ((VAR & 1) != 0) {
    __syncthreads();
    if (p.trace && blockIdx.x - p.trace_block0 < 64 && nk >= 80)
      for (int i = threadIdx.x; i < 8 * 16 * 8; i += 512)
        p.trace[(size_t)(blockIdx.x - p.trace_block0) * (8 * 16 * 8) + i] = ((unsigned long long *)(smem + 2 * STAGE_BYTES))[i];
  }
  recombine_and_store<D0, ND>(p, acc, tm * 64 + wm * 32 + (lane & 31), tn * 128 + g * 64 + wn * 32 + 4 * (lane >> 5));
#endif
}

} // namespace ozhip
