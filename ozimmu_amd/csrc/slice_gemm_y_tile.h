// slice_gemm_y_tile.h — the "k64" tile of the wide slice GEMM: v_mfma_i32_16x16x64_i8 with ONE slice product over 64 k per
// instruction.  Included by slice_gemm_w_kernel.h (same persistent workgroups; VARW_K64 selects this tile function).
//
// The paired tile (slice_gemm_x_tile.h) keeps the 32-k step of the staged image and pays for it with a zero operand half
// on every even diagonal (25 matrix-pipe slots for 45 products at S = 9) and 1.8x the LDS fragment reads.  This tile
// stages TWO k-blocks per step instead: every instruction is a full product A_i B_j over 64 k (45 slots for 45 products,
// each doing the work of half a 32x32x32 instruction at 18 % less energy per MAC), and an A fragment (16 rows x 64 k)
// feeds 2 * (S - i) instructions.  160 KiB of LDS then hold two A stages and ONE wave-private B stage of a 64 x 128 tile at
// S = 9 (not of a 96 x 128 one): per k the tile reads 25 % fewer fragment bytes from LDS than the 32x32x32 tile does, but
// stages 29 % more bytes per MAC from L2.  All B fragments of a step (2 * S * 4 registers) are in registers before the
// first copy of the next stage is issued, so that stage's B may land in the buffer the current one came from (VARW_B1).
//
// Staged image per row-block and step: the 2 * S consecutive 1 KiB blocks [k-block][slice] of the planes (layout.h) - one
// linear run, so the copies are w_tile's with a run of 2 * S blocks.  Fragment of slice s, rows 16 h + r, k-group g:
//     base + ((row-block * 2 + (g >> 1)) * S + s) KiB + (g & 1) * 512 + (16 h + r) * 16
// i.e. one per-lane offset (g >> 1) * SL KiB + (g & 1) * 512 + r * 16 for A and B alike (SL = staged slices).  Single-pass modes
// and the FIRST diagonal pass of the two-pass modes (diagonals 0 .. ND-1 need the slices 0 .. ND-1 only: the staged run of a
// k-block is then the first SL of its S blocks); an even number of k-blocks per pass.
//
// VARW_BREG (round 4): the B row-block of a wave is read by that wave alone, and the planes already hold it in fragment order
// (16 bytes per lane), so staging it through LDS is a round trip for nothing: 72 of the 108 KiB a step stages at S = 9, written
// to LDS by the DMA and read straight back.  With VARW_BREG the 2 * S B fragments of the NEXT step are loaded global -> VGPR
// (global_load_dwordx4, 4 runs of 256 contiguous bytes per instruction) into a second register set while the current step
// multiplies out of the first; steps alternate between the two sets (the step function is instantiated for both parities, no
// register copies; the sets are named registers, slice_gemm_w_kernel.h: OZ_BREG_FIRST).  LDS then holds the two shared A stages only (72 KiB at S = 9), a step writes a third of the bytes to LDS
// and reads two thirds of the fragments from it.  The loads are inline asm like the LDS-DMA copies and ordered by the same
// counted vmcnt; registers: 16 * WA * S accumulators + 2 * (2 * S * 4) for B: fits WA = 2 up to S = 9.
#pragma once

namespace ozhip {

// MFMA slots of one 64-k step: 16-row block a outermost, A slice i ascending, B slice j descending over the pairs with
// i + j <= S - 1, column block b innermost.  A "group" is the run of slots that share one A fragment.
template <int S, int WA> // S here: staged slices = diagonals of the pass
struct YSched {
  static constexpr int MA = 2 * WA;
  static constexpr int MAXG = MA * S, MAXS = MA * S * S * 2;
  int ns = 0, ng = 0;
  int sa[MAXS] = {}, si[MAXS] = {}, sj[MAXS] = {}, sb[MAXS] = {}, sg[MAXS] = {};
  int gfirst[MAXG] = {}, g_a[MAXG] = {}, g_i[MAXG] = {};
  bool first[MAXS] = {}; // the slot is the step's first MFMA on its accumulator tuple (block, column block, diagonal)
  constexpr YSched() {
    for (int a = 0; a < MA; a++)
      for (int i = 0; i < S; i++) {
        bool any = false;
        for (int j = S - 1; j >= 0; j--) {
          if (i + j > S - 1) continue;
          for (int b = 0; b < 2; b++) {
            if (!any) {
              gfirst[ng] = ns;
              g_a[ng] = a;
              g_i[ng] = i;
              any = true;
            }
            sa[ns] = a; si[ns] = i; sj[ns] = j; sb[ns] = b; sg[ns] = ng;
            ns++;
          }
        }
        if (any) ng++;
      }
    for (int q = 0; q < ns; q++) {
      first[q] = true;
      for (int r = 0; r < q; r++)
        if (sa[r] == sa[q] && sb[r] == sb[q] && si[r] + sj[r] == si[q] + sj[q]) first[q] = false;
    }
  }
  constexpr int max_j_from(int s0) const {
    int m = 0;
    for (int s = s0; s < ns; s++) m = sj[s] > m ? sj[s] : m;
    return m;
  }
  constexpr int last_use(int j, int b) const { // last slot of the step that reads B fragment (j, b)
    int l = -1;
    for (int s = 0; s < ns; s++)
      if (sj[s] == j && sb[s] == b) l = s;
    return l;
  }
};

template <int S, int WA>
inline constexpr YSched<S, WA> kYSched{};

// One output tile of (32*WA) x 128: rows start at A row-block rb0, columns at B row-block 4*tn.  Step structure, ring and
// copy schedule as w_tile / x_tile; a step covers the k-blocks 2 k, 2 k + 1 of the pass.
#ifndef OZ_Y_RING
#define OZ_Y_RING 4 // A fragment ring entries (tools/gemm_ablate.hip -DOZ_Y_RING=n: A/B)
#endif
// `prologue_hook(phase)`: phase 1 runs between the issue of the first stage's copies and the wait for them: whatever round trip it
// makes (w_persistent: the NEXT tile's claim ticket) hides under the latency this tile waits out anyway; phase 0 runs in FRONT of
// the copies (round 6 tried the ticket's atomic there, collected in phase 1: neutral to negative, slice_gemm_w_kernel.h; the phase stays
// for the next idea).
struct NoHook {
  template <class P>
  __device__ __forceinline__ void operator()(P) const {}
};
template <int S, int D0, int ND, int WA, int VARW, int STAG, int DMA0_, int DMAE_, int TAIL_, int RING = OZ_Y_RING, class HOOK = NoHook>
__device__ __forceinline__ void y_tile(const SliceGemmArgs &p, char *smem, const uint32_t rb0, const uint32_t tn,
                                       const uint32_t xcd, HOOK &&prologue_hook = HOOK()) {
  static_assert(D0 == 0 && ND <= S, "k64 tile: the diagonals 0 .. ND-1 (single pass, or the first pass of a two-pass mode)");
#define YC (kYSched<ND, WA>)
  constexpr int SL = ND, MA = YC.MA, KSL = 2 * SL; // staged slices; staged blocks per row-block and step
  constexpr int NA = 2, PD = 1;
  constexpr bool BREG = (VARW & VARW_BREG) != 0;
  constexpr int NB = BREG ? 0 : (VARW & VARW_B1) ? 1 : 2;
  static_assert((VARW & VARW_NA3) == 0, "k64 tile: two A buffers (prefetch distance 1)");
  // VARW_ACCN: the accumulators are named registers (slice_gemm_w_kernel.h).  VARW_BHI (with VARW_ACCN and one B buffer): the
  // NHI highest B slices do not pass through LDS - 11 staged slices need 88 KiB for two A stages and 88 KiB for one B stage of a
  // 64 x 128 tile, 16 KiB more than a CU has.  The fragments of the slices j >= 9 are loaded global -> v[144:159] instead and
  // refilled IN PLACE: slice j is read by the groups i <= SL - 1 - j only, so in the step's last 16-row block its last reader is
  // that block's group i = SL - 1 - j - for j = 10 / 9 the block's first / second group, 110 / 90 MFMA slots (~1 us) before the
  // next step's first group asks for the new fragment.  (The low slices are needed until the step's last slots: in place they
  // would have a few slots; they keep the LDS round trip, whose reads are issued behind the barrier.)
  constexpr bool ACCN = (VARW & VARW_ACCN) != 0;
  constexpr int NHI = (VARW & VARW_BHI) ? (SL == 12 ? 4 : SL > 9 ? SL - 9 : 0) : 0;
  constexpr int SLB = SL - NHI, KSLB = 2 * SLB; // B slices staged through LDS; their blocks per row-block and step
  static_assert(NHI == 0 || (ACCN && NB == 1 && !BREG), "in-place B slices: named accumulators, one B buffer");
  // clobber sets of the tile's asm statements (slice_gemm_w_kernel.h): inside the k loop / behind it
  constexpr int CLK = BREG ? OZ_CL_P9 : SL == 12 ? OZ_CL_P12 : OZ_CL_P11, CLE = BREG ? OZ_CL_P9_ACC : SL == 12 ? OZ_CL_P12 : OZ_CL_P11_ACC;
  constexpr int BHI0 = accn_bhi_first(CLK); // first register of the in-place B fragments
  static_assert(NHI * 8 <= accn_v_first(CLK) - BHI0, "the in-place fragments sit below the accumulators");
  constexpr int A_STAGE = WA * KSL * FRAG_BYTES;
  constexpr int B_STAGE = 4 * KSLB * FRAG_BYTES;
  constexpr int OFF_B = NA * A_STAGE;
  constexpr int NQA = (WA * KSL + 3) / 4;
  constexpr int NDMA = NQA + (BREG ? KSL : KSLB);
  constexpr int R = (SL == 12 && (VARW & VARW_ACCN)) ? 2 : RING; // (fp64_int8_12: 480 of 512 registers are operands)
  constexpr int NG = MA * SL;
  constexpr bool NO_GLOBAL = (VARW & (VARW_NO_GLOBAL | VARW_MFMA_ONLY)) != 0;
  constexpr bool MFMA_ONLY = (VARW & VARW_MFMA_ONLY) != 0;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  // ---- VARW_TRACE (tools/gemm_ablate.hip -DABLATE_TILE_TRACE only): wall-clock stamps (s_memrealtime, 100 MHz) of the phases of
  // every tile a workgroup runs -> p.acc, which a single-pass launch does not use otherwise.  0: tile entry (ticket in hand),
  // 1: first stage's copies issued + the next ticket drawn, 2: first stage landed and the barrier passed (k loop starts),
  // 3: in front of the last step, 4: behind it, 5: epilogue issued (p.dump_only & 1: and its stores drained), + the shader
  // cycles (s_memtime) at 0 and 5.  A stamp waits for the scalar memory path (lgkmcnt(0)): a few tens of cycles each.
  constexpr bool TTRACE = (VARW & VARW_TRACE) != 0;
  unsigned long long tt[TTRACE ? 8 : 1] = {}, tcyc[TTRACE ? 2 : 1] = {}; // (6, 7: inside the prologue - set-up done, copies issued)
  auto tstamp = [&](auto ic) {
    if constexpr (TTRACE) {
      unsigned long long t;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      tt[decltype(ic)::value] = t;
    }
  };
  auto tcycles = [&](auto ic) {
    if constexpr (TTRACE) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      tcyc[decltype(ic)::value] = t;
    }
  };
  auto tflush = [&](bool overlapped) {
    if constexpr (TTRACE) {
      if (p.dump_only & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tstamp(std::integral_constant<int, 5>{});
      tcycles(std::integral_constant<int, 1>{});
      if (threadIdx.x == 0) {
        unsigned long long *base = reinterpret_cast<unsigned long long *>(p.acc);
        const unsigned long long seq = atomicAdd(base + blockIdx.x, 1ull);
        if (seq < 40) {
          unsigned long long *r = base + 16384 + ((size_t)blockIdx.x * 40 + seq) * 16;
#pragma unroll
          for (int i = 0; i < 6; i++) r[i] = tt[i];
          r[6] = tcyc[0];
          r[7] = tcyc[1];
          r[8] = ((unsigned long long)rb0 << 32) | tn;
          r[9] = (overlapped ? 1ull : 0ull) | ((unsigned long long)xcd << 8);
          r[10] = tt[6];
          r[11] = tt[7];
        }
      }
    }
  };
  tstamp(std::integral_constant<int, 0>{});
  tcycles(std::integral_constant<int, 0>{});

  // ---- staging: w_tile's, with runs of 2 * S blocks ------------------------------------------------------------
  const size_t rb_stride = (size_t)p.KB * (size_t)(S * FRAG_BYTES);
  const uint32_t rba_last = p.rba - 1u;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const size_t pass0 = (size_t)p.kb0 * (S * FRAG_BYTES);
  const int8_t *a_src[NQA];
  uint32_t a_lds[NQA];
#pragma unroll
  for (int t = 0; t < NQA; t++) {
    uint32_t q = (uint32_t)(wave * NQA + t);
    if (q > (uint32_t)(WA * KSL - 1)) q = WA * KSL - 1;
    const uint32_t a = q / KSL, c = q - a * KSL; // staged block c = (k-block c / SL of the step, slice c % SL)
    uint32_t rb = rb0 + a;
    if (rb > rba_last) rb = rba_last;
    a_src[t] = uniform_ptr(p.a_planes + rb * rb_stride + ((c / SL) * S + c % SL) * FRAG_BYTES + pass0);
    a_lds[t] = q * FRAG_BYTES;
  }
  const int8_t *b_src = uniform_ptr(p.b_planes + (size_t)(4u * tn + wave) * rb_stride + pass0);
  // LDS pointers stay in address space 3 (a generic pointer derived from the dynamic LDS symbol makes hipcc emit a null check
  // against src_shared_base in some instantiations, which its own verifier then rejects: "Operand has incorrect register class")
  OZ_AS3 char *const lds = (OZ_AS3 char *)smem;
  const uint32_t lds0 = (uint32_t)(size_t)lds;
  const uint32_t ldsb0 = lds0 + OFF_B + wave * (KSLB * FRAG_BYTES);
  auto copy_a = [&](int t, uint32_t voff, uint32_t lds_a) {
    if constexpr (NO_GLOBAL) return;
    glds16<0>(a_src[t], voff, lds_a, a_lds[t]);
  };
  auto copy_b = [&](auto sc, uint32_t voff, uint32_t lds_b) {
    if constexpr (NO_GLOBAL) return;
    constexpr int c = decltype(sc)::value;          // staged block c of the wave's run
    constexpr int kbl = c / SLB, s = c % SLB;       // k-block of the step, slice
    constexpr int G = 4;                            // blocks per immediate-offset group (inside one k-block's run)
    constexpr int g0 = s / G * G;
    glds16<(s % G) * FRAG_BYTES>(b_src + (kbl * S + g0) * FRAG_BYTES, voff, lds_b, (uint32_t)((kbl * SLB + g0) * FRAG_BYTES));
  };
  // VARW_BREG: B fragment (b, j) of step `kstep`, k-group g = lane >> 4, row r = lane & 15: byte
  //   ((2 kstep + (g >> 1)) * S + j) KiB + (g & 1) * 512 + (16 b + r) * 16   of the wave's row-block
  const uint32_t vG = (uint32_t)(((lane >> 4) >> 1) * (S * FRAG_BYTES) + ((lane >> 4) & 1) * 512 + (lane & 15) * 16);
  static_assert(!BREG || OZ_BREG_FIRST + 16 * SL <= (ACCN ? 224 : 256), "two register sets of 2 * SL fragments in v[OZ_BREG_FIRST : 255]");
  auto load_b = [&](auto cc, uint32_t kstep, auto set) {
    constexpr int c = decltype(cc)::value;          // in the order the next step needs them: j descending, b
    constexpr int j = SL - 1 - c / 2, b = c & 1;
    constexpr int G = 4, g0 = j / G * G;            // slices per immediate-offset group (offset < 4096)
    constexpr int REG = OZ_BREG_FIRST + ((decltype(set)::value * 2 + b) * SL + j) * 4;
    if constexpr (!NO_GLOBAL) {
      const uint32_t voff = vG + kstep * (uint32_t)(2 * S * FRAG_BYTES);
      if constexpr (ACCN)
        gload16_named_accn<CLK, REG, (j - g0) * FRAG_BYTES + b * 256>(b_src + g0 * FRAG_BYTES, voff);
      else
        gload16_named<REG, (j - g0) * FRAG_BYTES + b * 256>(b_src + g0 * FRAG_BYTES, voff);
    }
  };
  // VARW_BHI: fragment (b, j >= SLB) of step `kstep` -> v[OZ_BHI_FIRST + ((b * NHI) + j - SLB) * 4 ...], same addressing as load_b
  auto load_hi = [&](auto jc, auto bc, uint32_t kstep) {
    constexpr int j = decltype(jc)::value, b = decltype(bc)::value;
    constexpr int G = 4, g0 = j / G * G;
    constexpr int REG = BHI0 + (b * NHI + (j - SLB)) * 4;
    if constexpr (!NO_GLOBAL) {
      const uint32_t voff = vG + kstep * (uint32_t)(2 * S * FRAG_BYTES);
      gload16_named_accn<CLK, REG, (j - g0) * FRAG_BYTES + b * 256>(b_src + g0 * FRAG_BYTES, voff);
    }
  };
  auto copy_n = [&](auto cc, int abuf, int bbuf, uint32_t kstep, auto bnext) { // bnext: register set (VARW_BREG)
    constexpr int c = decltype(cc)::value;
    const uint32_t voff = lane_off + kstep * (uint32_t)(2 * S * FRAG_BYTES);
    if constexpr (c < NQA) {
      copy_a(c, voff, lds0 + abuf * A_STAGE);
    } else if constexpr (BREG) {
      load_b(std::integral_constant<int, c - NQA>{}, kstep, bnext);
    } else {
      // one B buffer: the copy overwrites what this step's B fragments were read from; those reads were issued a step's
      // tail ago, the wait makes "they have returned" a guarantee
      if constexpr (NB == 1 && c == NQA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      copy_b(std::integral_constant<int, c - NQA>{}, voff, ldsb0 + bbuf * B_STAGE);
    }
  };

  // ---- accumulators: MA x 2 x S tuples of 4 registers; the first 64 in the AGPR half ----------------------------
  constexpr int NACC = MA * 2 * ND, NACC_A = ACCN ? 1 : (NACC < 64 ? NACC : 64), NACC_V = (!ACCN && NACC > 64) ? NACC - 64 : 1;
  static_assert(!ACCN || NACC <= (BREG ? 72 : SL == 12 ? 96 : 88), "named accumulators: a[0:255] + v[160:255] (register form: + v[80:111])");
  v4i accA[NACC_A], accV[NACC_V];
  // (the register kernel never zeroes: the first MFMA of its first step on every tuple takes the constant 0 as C - 288
  // v_accvgpr_write / v_mov in front of every tile were 0.6 us of a 2.4 us prologue that is all instruction issue,
  // profiles/r6_ablate/r6d_tile_trace_*)
  // (not in the multi-product kernels, VARW_ZFILL: with the peeled first steps next to its product loop hipcc's block placement turned
  // their k loops irreducible and its wait insertion gave up on counted lgkmcnt waits there - a full LDS drain per fragment group;
  // a ZGEMM's products never take the overlapped path either: they keep the round-5 shape of the tile function)
  constexpr bool ZERO_BY_FIRST_STEP = BREG && ACCN && !MFMA_ONLY && (VARW & VARW_ZFILL) == 0;
  if constexpr (ZERO_BY_FIRST_STEP) {
  } else if constexpr (ACCN) {
    static_for<NACC>([&](auto xc) { zero_accn<CLK, decltype(xc)::value>(); });
  } else {
#pragma unroll
    for (int x = 0; x < NACC_A; x++)
#pragma unroll
      for (int r = 0; r < 4; r++) accA[x][r] = 0;
#pragma unroll
    for (int x = 0; x < NACC_V; x++)
#pragma unroll
      for (int r = 0; r < 4; r++) accV[x][r] = 0;
  }
  auto mfma = [&](auto xc, const v4i &b, const v4i &a) {
    constexpr int X = decltype(xc)::value;
    if constexpr (ACCN)
      mfma16_accn<CLK, X, -1>(b, a);
    else if constexpr (X < 64)
      mfma16_agpr(accA[X], b, a);
    else
      mfma16_vgpr(accV[X - 64], b, a);
  };
  auto mfma_named = [&](auto xc, auto reg, const v4i &a, auto cl, auto zc) { // B operand in the named registers (VARW_BREG); zc: C = 0
    constexpr int X = decltype(xc)::value, REG = decltype(reg)::value;
    if constexpr (ACCN)
      mfma16_accn<decltype(cl)::value, X, REG, decltype(zc)::value>(a, a);
    else if constexpr (X < 64)
      mfma16_agpr_named<REG>(accA[X], a);
    else
      mfma16_vgpr_named<REG>(accV[X - 64], a);
  };

  // ---- circular K (in 64-k steps) with the per-XCD phase hint -----------------------------------------------------
  const uint32_t nk = (p.kb1 - p.kb0) >> 1;
  uint32_t *phase = p.phase ? p.phase + (uint32_t)PHASE_LINE_WORDS * xcd : nullptr;
  uint32_t koff = 0;
  if (phase && 2u * nk > p.phase_min_kb) {
    if (threadIdx.x == 0)
      *(volatile OZ_AS3 uint32_t *)lds = __hip_atomic_load(phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    koff = (*(volatile OZ_AS3 uint32_t *)lds + 1u) % nk;
    __syncthreads();
  } else {
    phase = nullptr;
  }
  koff = __builtin_amdgcn_readfirstlane(koff);
  auto koff_next = [&](uint32_t k) { return k + 1 == nk ? 0u : k + 1; };

  v4i cf[MFMA_ONLY ? SL : 1]; // MFMA-only ablation: full-entropy operands in registers
  if constexpr (MFMA_ONLY) {
#pragma unroll
    for (int s = 0; s < SL; s++) {
      uint32_t x = (uint32_t)(lane * SL + s) * 2654435761u + blockIdx.x;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        cf[s][c] = (int)x;
      }
      asm volatile("" : "+v"(cf[s]));
    }
  }

  // ---- schedule of one step (compile time; see w_tile for the event model) ----------------------------------------
  constexpr int NS = YC.ns;
  constexpr int NRP = (NG - 1 + R - 1) / R * R;
  constexpr int TAIL = TAIL_ < NS ? TAIL_ : NS - 1;
  constexpr int XS = NS - TAIL;
  constexpr int DMA0 = DMA0_ >= 0 ? DMA0_ : YC.gfirst[1 < NG ? 1 : 0];
  constexpr int DMAE_FIT = NDMA > 1 ? (XS - 1 - DMA0) / (NDMA - 1) : 1;
  constexpr int DMAE = DMAE_ < DMAE_FIT ? DMAE_ : (DMAE_FIT > 1 ? DMAE_FIT : 1);
  constexpr int JT = YC.max_j_from(XS) + 1; // B slices j < JT are still needed after the barrier slot
  static_assert(YC.ng == NG, "empty (a, i) groups are not supported by the ring arithmetic");
  static_assert(NRP + 2 - R > NG - 1 || YC.gfirst[NRP + 2 - R <= NG - 1 && NRP + 2 - R >= 0 ? NRP + 2 - R : 0] >= XS,
                "the first ring read of the next stage must come after the barrier");
  static_assert(NG >= 2, "at least two groups per step");
  static_assert(DMA0 + (NDMA - 1) * DMAE < XS, "every copy of a stage is issued before the barrier slot");
  static_assert(NB != 1 || DMA0 + NQA * DMAE >= YC.gfirst[1], "one B buffer: its refill starts behind the step's first group");
  static_assert(NHI == 0 || (JT <= SLB && YC.last_use(SLB, 1) + 2 < XS), "in-place B slices: released, and their loads issued, before the barrier slot");

  const int g4 = lane >> 4;
  const uint32_t vF = (uint32_t)((g4 >> 1) * (SL * FRAG_BYTES) + (g4 & 1) * 512 + (lane & 15) * 16);
  const OZ_AS3 char *la0 = lds + vF;
  const uint32_t vFB = (uint32_t)((g4 >> 1) * (SLB * FRAG_BYTES) + (g4 & 1) * 512 + (lane & 15) * 16);
  const OZ_AS3 char *lb0 = lds + OFF_B + wave * (KSLB * FRAG_BYTES) + vFB;
  int abuf = 0, bbuf = 0;
  v4i bj[2][SLB], af[R], af0; // B fragments [column block][slice] (VARW_BREG: named registers instead)
  auto read_a = [&](auto gc, const OZ_AS3 char *la) { // A fragment of group g -> af0 (g == 0) or ring slot (g - 1) % R
    constexpr int g = decltype(gc)::value;
    const v4i f = *(const OZ_AS3 v4i *)(la + ((YC.g_a[g] >> 1) * KSL + YC.g_i[g]) * FRAG_BYTES + (YC.g_a[g] & 1) * 256);
    if constexpr (g == 0)
      af0 = f;
    else
      af[(g - 1) % R] = f;
  };
  auto read_b = [&](int b, int j, const OZ_AS3 char *lb) { bj[b][j] = *(const OZ_AS3 v4i *)(lb + j * FRAG_BYTES + b * 256); };

  // ---- prologue -----------------------------------------------------------------------------------------------
  uint32_t k_issue = koff;
  tstamp(std::integral_constant<int, 6>{});
  prologue_hook(std::integral_constant<int, 0>{});
  if (0u < nk) {
    static_for<NDMA>([&](auto cc) { copy_n(cc, 0, 0, k_issue, std::integral_constant<int, 0>{}); });
    static_for<2 * NHI>([&](auto cc) {
      load_hi(std::integral_constant<int, SLB + decltype(cc)::value / 2>{}, std::integral_constant<int, decltype(cc)::value & 1>{}, k_issue);
    });
    k_issue = koff_next(k_issue);
  }
  tstamp(std::integral_constant<int, 7>{});
  prologue_hook(std::integral_constant<int, 1>{});
  tstamp(std::integral_constant<int, 1>{});
  if constexpr (!NO_GLOBAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (!MFMA_ONLY) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (!BREG) {
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int j = 0; j < SLB; j++) read_b(b, j, lb0);
    }
    static_for<(R < NG ? R : NG)>([&](auto gc) { read_a(gc, la0); });
  }
  tstamp(std::integral_constant<int, 2>{});
  asm volatile("s_nop 7" ::: "memory"); // zero-fill -> first MFMA reading it as C

  uint32_t it = 0;

  // ---- the recombination in the shadow of the LAST step's MFMAs (register kernels with named accumulators) -----------------------
  // The tile boundary of the register kernel IS the FP64 recombination (profiles/r4_ablate/r4l_tile_boundary_...: 3.8 - 4.0 us per
  // 64 x 128 tile whatever K is: 9 x (v_accvgpr_read + v_cvt_f64_i32 + v_fma_f64) per output with the matrix pipe idle; 11 % of the
  // kernel at K = 512).  A step walks the 16-row blocks a = 0 .. MA-1 one after the other, so in the last step the accumulators of
  // block a are final once its MFMAs have issued: their recombination, cut into micro-operations (one diagonal of one column
  // pair: 2 reads, 2 conversions, 2 fma), is issued BETWEEN the MFMAs of block a + 1 - a wave issues VALU work while the matrix
  // pipe works off a 16-cycle MFMA - and only the last block's chain is left behind the k loop.  Same operations on every
  // element in the same order as recombine_and_store16's 16-byte store form: bit-identical.  Interior tiles of a real, final,
  // single-chunk product with an even ldc and a 16-byte aligned C; every other tile takes the plain last step and epilogue.  The
  // condition is uniform over the WORKGROUP (the overlapped step has no barrier: the waves must agree on how many they pass).
  // Round 5 first built this with the accumulators as compiler values: hipcc re-assigned 13 tuples between the k loop and this
  // step THROUGH SCRATCH (wrong sums, -10 %: profiles/r5_ablate/r5c_*).  With named accumulators there is nothing to re-assign; the
  // step lists only the second B set as clobbered (OZ_CL_P9_SET1), so the chain's values live in the first set's registers.
  // Round 6 (profiles/r6_ablate/r6b_tile_trace_*): stamped per tile, the overlapped step took 7.0 us against 3.2 us of MFMA issue
  // time - the chain hid almost nothing.  Its micro-operation was read -> convert -> fma on ONE temporary, twice: six
  // instructions each waiting for the one before it (the compiler reuses the register), and an in-order wave that stalls on
  // a VALU dependency does not issue its next MFMA either.  The chain is now a software pipeline over the block's 8 outputs
  // per lane x ND diagonals = 8 ND elements e = (d, output): stream step t reads the accumulator of element t, converts
  // element t - 2 and accumulates element t - 4 - three independent instructions per MFMA slot, every dependency two slots
  // (>= 32 cycles) old, the fma of an output 8 slots behind its predecessor.  Same operations on every element in the same
  // order (x = fma(double(c_d), 2^.., x), d ascending from fma(.., 0.0)): bit-identical.
  constexpr int EPE = 8 * ND;                 // elements of a 16-row block per lane: (diagonal d, output o = 2 * unit + w), o fastest
  constexpr int SPA = NS / MA;                // MFMA slots per 16-row block
  constexpr int EPI_T0 = 3;                   // first stream step this many slots into the next block (XDL write -> VALU read)
  // unit u (column pair (b, vp) of the block) is complete behind stream step 8 (ND - 1) + 2 u + 1 + 4: scaled / stored at
  constexpr int EPI_TAIL0 = EPI_T0 + 8 * (ND - 1) + 6, EPI_TAILD = 4; // slots TAIL0 + TAILD u (scale) and + 2 behind it (store)
  constexpr bool OVERLAP_BUILT = BREG && ACCN && SL == 9 && !NO_GLOBAL && !MFMA_ONLY && ((VARW >> 8) & 3) == 0 &&
                                 (VARW & VARW_NO_EPILOGUE) == 0 && SPA * MA == NS && SPA > EPI_TAIL0 + 3 * EPI_TAILD + 2 &&
                                 (VARW & VARW_ZFILL) == 0; // (the multi-product kernels run complex products only: never this form)
  const uint32_t e_mu = rb0 * 32u, e_nu = tn * 128u + (uint32_t)wave * 32u;
  bool overlap = false;
  if constexpr (OVERLAP_BUILT) {
    overlap = p.epi_overlap && p.L == 7 && p.final && !p.cplx && !p.acc_in && e_mu + 16u * MA <= p.M && tn * 128u + 128u <= p.N &&
              (p.ldc & 1u) == 0 && p.ldc < (1u << 25) && (reinterpret_cast<uintptr_t>(p.c) & 15u) == 0 && nk >= 2;
#ifdef OZIMMU_HIP_TEST_HOOKS
    overlap = overlap && !p.dump;
#endif
  }
  double e_ea[OVERLAP_BUILT ? MA : 1], e_eb[OVERLAP_BUILT ? 8 : 1];
  double e_xx[OVERLAP_BUILT ? 8 : 1] = {}, e_cv[3] = {0, 0, 0}, e_v0 = 0, e_v1 = 0;
  int e_rd[3] = {0, 0, 0};
  double2 e_old[OVERLAP_BUILT ? 4 : 1];
  uint32_t e_boff = 0;
  bool e_odd = false, e_rmw = false;
  // Addresses of C: a wave-UNIFORM 64-bit base (SGPR pair) + the lane's 32-bit byte offset e_boff, and the stores are asm
  // statements in that form.  Written as plain pointer arithmetic, hipcc copied p.c into a VGPR pair at kernel entry, hoisted
  // the lane's offset there too (both are invariant over the tile loop), SPILLED both - they live across k loops that leave it
  // 80 registers - and reloaded them from scratch in front of every store of the overlapped step, each reload behind an
  // `s_waitcnt vmcnt(0)` that also waits for the previous unit's store to retire: 16 memory round trips per tile, the whole
  // 3.9 us by which the step exceeded its MFMA time in profiles/r6_ablate/r6b_tile_trace_* (rounds 5 and 6 alike).
  // (and with the base spelled as a pointer expression it still did: the final add became a VALU add on a spilled VGPR copy of
  // p.c.  The four column bases of a tile - unit u = column pair 16 (u >> 1) + 2 (u & 1) of the wave's 32 columns, block 0 - are
  // therefore added up by SALU instructions in an asm statement, once per tile in epi_setup; a block is an immediate offset.)
  uint32_t e_blo[4] = {0, 0, 0, 0}, e_bhi[4] = {0, 0, 0, 0};
  auto e_colp = [&](int u) { // wave-uniform: first row of block 0 of unit u's first column (+ nl columns per lane, in e_boff)
    return reinterpret_cast<const int8_t *>(((uint64_t)e_bhi[u] << 32) | e_blo[u]);
  };
  auto epi_setup = [&]() { // issued a block's worth of MFMAs ahead of the first use: exponents, the old C of block 0
    // the lane id from the hardware, HERE: `lane` itself lives across the k loops in a scratch slot (its reload cost this step a
    // full vmcnt drain), and what is computed from it must not be a loop invariant that is hoisted to the kernel's entry and spilled
    uint32_t ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ln));
    const uint32_t nl = 4u * (ln >> 4);
    e_odd = (ln & 1u) != 0;
    e_rmw = p.beta != 0.0;
    e_boff = ((nl + (ln & 1u)) * (uint32_t)p.ldc + (ln & 14u)) * 8u; // (< 2^32: the overlapped form runs for ldc < 2^25)
#pragma unroll
    for (int a = 0; a < MA; a++) e_ea[a] = 0x1p-44 * p.ea[e_mu + (ln & 15u) + 16 * a];
    const double *eb_lane = p.eb + e_nu + nl;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int v = 0; v < 4; v++) e_eb[4 * b + v] = eb_lane[16 * b + v];
    {
      const uint64_t cb = (uint64_t)reinterpret_cast<uintptr_t>(p.c);
      const uint32_t clo = __builtin_amdgcn_readfirstlane((uint32_t)cb), chi = __builtin_amdgcn_readfirstlane((uint32_t)(cb >> 32));
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint64_t off = ((uint64_t)(e_nu + 16u * (u >> 1) + 2u * (u & 1)) * p.ldc + e_mu) * 8u;
        const uint32_t olo = __builtin_amdgcn_readfirstlane((uint32_t)off), ohi = __builtin_amdgcn_readfirstlane((uint32_t)(off >> 32));
        asm volatile("s_add_u32 %0, %2, %4\n\ts_addc_u32 %1, %3, %5" : "=&s"(e_blo[u]), "=&s"(e_bhi[u]) : "s"(clo), "s"(chi), "s"(olo), "s"(ohi) : "scc");
      }
    }
    if (e_rmw) {
#pragma unroll
      for (int u = 0; u < 4; u++) e_old[u] = *reinterpret_cast<const double2 *>(e_colp(u) + e_boff);
    }
  };
  // stream step t of block A (compile-time indices); CLR: the clobber set of its statements.  The read of element t is issued by
  // the caller (fused with the slot's MFMA in the overlapped step: epi_read_reg names the register); here: convert element t - 2,
  // accumulate element t - 4.  The scales 2^(46 - L (d + 2)) are literals: the overlapped form runs for L = 7 only (K <= 2^17).
  auto epi_read_reg = [](auto a_tag, auto t_tag) constexpr { // accumulator register of element t of block A: AGPR index, or 256 + VGPR index
    constexpr int A = decltype(a_tag)::value, t = decltype(t_tag)::value;
    constexpr int d = t / 8, o = t % 8, u = o >> 1, b = u >> 1, vp = u & 1, x = (A * 2 + b) * ND + d, v = 2 * vp + (o & 1);
    return x < 64 ? 4 * x + v : 256 + accn_v_first(OZ_CL_P9) + 4 * (x - 64) + v;
  };
  // slot t of the block behind block A (t - EPI_T0 = the stream step): [MFMA X with the B fragment in v[BR : BR + 3]] + the step's
  // read / accumulate / convert, one statement (X < 0: the serial part behind the loop)
  auto epi_stream = [&](auto a_tag, auto t_tag, auto cl_tag, auto x_tag, auto br_tag, const v4i &af_) {
    constexpr int t = decltype(t_tag)::value - EPI_T0, CLR = decltype(cl_tag)::value;
    constexpr int X = decltype(x_tag)::value, BR = decltype(br_tag)::value;
    constexpr bool RD = t >= 0 && t < EPE, CVT = t >= 2 && t - 2 < EPE, ACCU = t >= 4 && t - 4 < EPE;
    constexpr int e = ACCU ? t - 4 : 0, d = e / 8, o = e % 8;
    constexpr int E = 46 - 7 * (D0 + d + 2);
    constexpr int RS = RD ? epi_read_reg(a_tag, std::integral_constant<int, (RD ? t : 0)>{}) : -1;
    epi_slot_asm<CLR, X, BR, RS, CVT, ACCU ? (d == 0 ? 1 : 2) : 0, E>(af_, e_rd[RD ? t % 3 : 0], e_cv[CVT ? (t - 2) % 3 : 0], e_rd[CVT ? (t - 2) % 3 : 0],
                                                                      e_xx[o], e_cv[e % 3]);
  };
  // unit u of block A: scale (q = 0), then pair the lanes, alpha / beta, store (q = 1)
  auto epi_tail = [&](auto a_tag, auto u_tag, auto q_tag) {
    constexpr int A = decltype(a_tag)::value, u = decltype(u_tag)::value, q = decltype(q_tag)::value;
    constexpr int b = u >> 1, vp = u & 1;
    if constexpr (q == 0) {
      // reference: x_ptr[tid] / (1l << 44) * a_max_exp[mi] * b_max_exp[ni]  (src/gemm.cu:140-141).  e_ea holds 2^-44 * a_max_exp
      // (epi_setup): x * 2^-44 is exact (|x| >= 2^-24 or 0) and so is 2^-44 * 2^(e+1) (>= 2^-1065, a representable power of
      // two), so (x * 2^-44) * ea and x * (2^-44 * ea) are the SAME real number rounded once - one multiplication less per output
      e_v0 = e_xx[2 * u] * e_ea[A] * e_eb[4 * b + 2 * vp];
      e_v1 = e_xx[2 * u + 1] * e_ea[A] * e_eb[4 * b + 2 * vp + 1];
    } else {
      // (the neighbour's value by ONE v_mov_b32_dpp per word: lane_pair_swap's update_dpp form hands the builtin an `old` operand
      // and costs a register copy in front of every DPP move - all lanes are written here, there is no old value to keep)
      auto swap = [](double v) {
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
      };
      const double s0 = swap(e_v0), s1 = swap(e_v1);
      double2 y;
      y.x = e_odd ? s1 : e_v0; // row 2i:     (column n from this lane | column n+1 from the even neighbour)
      y.y = e_odd ? e_v1 : s0; // row 2i + 1: (column n from the odd neighbour | column n+1 from this lane)
      if (e_rmw) {
        y.x = fma(p.alpha, y.x, p.beta * e_old[u].x);
        y.y = fma(p.alpha, y.y, p.beta * e_old[u].y);
        if constexpr (A + 1 < MA) // the old values of the next block's unit u: a block's worth of MFMAs ahead of their use
          e_old[u] = *reinterpret_cast<const double2 *>(e_colp(u) + 128 * (A + 1) + e_boff);
      } else {
        y.x = p.alpha * y.x;
        y.y = p.alpha * y.y;
      }
      typedef double v2d_t __attribute__((ext_vector_type(2)));
      const v2d_t yv = {y.x, y.y};
      // (the wait state: a store of more than 8 bytes reads its data registers a cycle late, and the compiler does not
      // look inside the statement - without it 0.08 % of the elements came out with the NEXT unit's values)
      asm volatile("global_store_dwordx4 %0, %1, %2 offset:%c3\n\ts_nop 0" : : "v"(e_boff), "v"(yv), "s"(e_colp(u)), "i"(128 * A) : "memory");
    }
  };
  // the scale / store micro-operations of block A's units at slot t of the block behind it
  auto epi_slot = [&](auto a_tag, auto t_tag) {
    constexpr int t = decltype(t_tag)::value;
    static_for<4>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if constexpr (t == EPI_TAIL0 + EPI_TAILD * u) epi_tail(a_tag, uc, std::integral_constant<int, 0>{});
      if constexpr (t == EPI_TAIL0 + EPI_TAILD * u + 2) epi_tail(a_tag, uc, std::integral_constant<int, 1>{});
    });
  };

  // PAR (VARW_BREG): the register set this step multiplies out of; the next step's fragments are loaded into the other one
  // EPI: the overlapped last step (above)
  auto step = [&](auto pf_tag, auto nx_tag, auto par_tag, auto epi_tag, auto first_tag) {
    constexpr bool PF = decltype(pf_tag)::value, NX = decltype(nx_tag)::value, EPI = decltype(epi_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value; // the tile's first step of the register kernel: nothing has zeroed the accumulators
    static_assert(!FIRST || (ZERO_BY_FIRST_STEP && !EPI), "the zeroing first step belongs to the register kernel");
    static_assert(!EPI || (!PF && !NX), "the overlapped last step prefetches nothing and meets no barrier");
    constexpr int PC = BREG ? decltype(par_tag)::value : 0, PN = BREG ? (PC ^ 1) : 0;
    const int abuf_n = abuf ^ 1;
    const int abuf_pf = abuf_n, bbuf_pf = NB == 1 ? 0 : (bbuf ^ 1);
    const uint32_t kb_pf = k_issue;
    if constexpr (PF) k_issue = koff_next(k_issue);
    const OZ_AS3 char *la = la0 + abuf * A_STAGE;
    const OZ_AS3 char *la_n = la0 + abuf_n * A_STAGE;
    const OZ_AS3 char *lb_n = lb0 + (NB == 1 ? 0 : (bbuf ^ 1) * B_STAGE);
    static_for<NS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int g = YC.sg[s];
      if constexpr (s == XS && NX && !MFMA_ONLY) {
        // (STAG >= 100, tools/gemm_ablate.hip only, WRONG RESULTS: bit 0 drops the vmcnt wait, bit 1 the lgkmcnt wait, bit 2 the
        // barrier - what each of them costs a step)
        constexpr int ABL = STAG >= 100 ? STAG / 100 : 0;
        if constexpr (!NO_GLOBAL && !(ABL & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!(ABL & 2)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the buffer this barrier releases is refilled right behind it
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (STAG > 0 && STAG < 100)
          for (int q = 0; q < wave; q++) asm volatile("s_nop %0" ::"n"(STAG - 1));
        read_a(std::integral_constant<int, 0>{}, la_n);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (YC.gfirst[g] == s && g >= 1 && !MFMA_ONLY) {
        constexpr int gn = g + R - 1;
        if constexpr (gn < NG) {
          read_a(std::integral_constant<int, gn>{}, la);
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (gn - 1 >= NRP && NX) {
          read_a(std::integral_constant<int, gn - NRP>{}, la_n);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (PF && s >= DMA0 && (s - DMA0) % DMAE == 0 && (s - DMA0) / DMAE < NDMA) {
        copy_n(std::integral_constant<int, (s - DMA0) / DMAE>{}, abuf_pf, bbuf_pf, kb_pf, std::integral_constant<int, PN>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int a = YC.sa[s], i = YC.si[s], j = YC.sj[s], b = YC.sb[s];
      constexpr int X = (a * 2 + b) * ND + (i + j);
      if constexpr (MFMA_ONLY)
        mfma(std::integral_constant<int, X>{}, cf[j], cf[i]);
      else if constexpr (BREG && EPI && a >= 1) // + the recombination stream of block a - 1: read, accumulate, convert in the MFMA's statement
        epi_stream(std::integral_constant<int, (a >= 1 ? a - 1 : 0)>{}, std::integral_constant<int, s - a * SPA>{}, std::integral_constant<int, OZ_CL_P9_SET1>{},
                   std::integral_constant<int, X>{}, std::integral_constant<int, OZ_BREG_FIRST + ((PC * 2 + b) * SL + j) * 4>{},
                   g == 0 ? af0 : af[(g >= 1 ? g - 1 : 0) % R]);
      else if constexpr (BREG)
        mfma_named(std::integral_constant<int, X>{}, std::integral_constant<int, OZ_BREG_FIRST + ((PC * 2 + b) * SL + j) * 4>{},
                   g == 0 ? af0 : af[(g >= 1 ? g - 1 : 0) % R], std::integral_constant<int, EPI ? OZ_CL_P9_SET1 : OZ_CL_P9>{},
                   std::integral_constant<bool, FIRST && YC.first[s]>{});
      else if constexpr (j >= SLB) // VARW_BHI: the fragment in its named registers
        mfma16_accn<CLK, X, BHI0 + (b * NHI + (j - SLB)) * 4>(af0, g == 0 ? af0 : af[(g >= 1 ? g - 1 : 0) % R]);
      else
        mfma(std::integral_constant<int, X>{}, bj[b][j < SLB ? j : 0], g == 0 ? af0 : af[(g >= 1 ? g - 1 : 0) % R]);
      if constexpr (EPI) {
        static_assert(PC == 1, "the overlapped step multiplies out of the second register set");
        if constexpr (s == 1) {
          epi_setup();
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (a >= 1) { // block a - 1 is final: its recombination stream runs through this block's slots
          epi_slot(std::integral_constant<int, a - 1>{}, std::integral_constant<int, s - a * SPA>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (PF && NHI > 0) { // the next step's fragment into the registers whose last reader has issued two slots ago
        static_for<2 * NHI>([&](auto cc) {
          constexpr int jh = SLB + decltype(cc)::value / 2, bh = decltype(cc)::value & 1;
          if constexpr (YC.last_use(jh, bh) + 2 == s) {
            load_hi(std::integral_constant<int, jh>{}, std::integral_constant<int, bh>{}, kb_pf);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      }
      if constexpr (s >= XS && NX && !MFMA_ONLY) {
        if constexpr (!BREG) {
          constexpr int NREF = 2 * (SLB - JT);                // fragments refreshed behind the barrier
          constexpr int RPT = (NREF + TAIL - 1) / TAIL;
#pragma unroll
          for (int u = 0; u < RPT; u++) {
            const int idx = (s - XS) * RPT + u;               // (j descending from SLB-1, b)
            if (idx < NREF) read_b(idx & 1, SLB - 1 - (idx >> 1), lb_n);
          }
        }
        if constexpr (s == XS + 1 || (TAIL == 1 && s == XS))
          if (phase && (it & 1u) == 0 && threadIdx.x == 0)
            __hip_atomic_store(phase, koff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NX && !MFMA_ONLY) {
      if constexpr (!BREG) {
#pragma unroll
        for (int j = JT - 1; j >= 0; j--)
#pragma unroll
          for (int b = 0; b < 2; b++) read_b(b, j, lb_n);
      }
      static_for<R - 1>([&](auto qc) {
        constexpr int q = decltype(qc)::value + 1;
        if constexpr (NRP + q - R + 1 > NG - 1 && q < NG) read_a(std::integral_constant<int, q>{}, la_n);
      });
      __builtin_amdgcn_sched_barrier(0);
    }
    koff = koff_next(koff);
    abuf = abuf_n;
    bbuf ^= 1;
  };
  if constexpr (BREG) {
    // Two steps per iteration, one per register set, as straight-line code (a branch on the step's parity inside the loop makes
    // hipcc assign the accumulators differently on the two sides and shuffle ~300 registers at every join), and ONE form of
    // the step: the last step of a tile prefetches like any other - the circular k walk wraps to the tile's first stage, so the
    // addresses are valid and the data is dropped (one stage of extra L2 traffic per tile: 1 / 128 at K = 8192) - instead of
    // a second pair of step bodies whose entry costs another round of register shuffling.  The host launches this form for
    // an even number of steps only (slice_gemm_launch.h).
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using NO = std::false_type;
    using ZF = std::integral_constant<bool, ZERO_BY_FIRST_STEP>; // the tile's first step writes the accumulators (C = 0)
    if constexpr (OVERLAP_BUILT) {
      if (overlap) {
        step(std::true_type{}, std::true_type{}, P0{}, NO{}, ZF{});
        it++;
        for (; it + 2 < nk;) {
          step(std::true_type{}, std::true_type{}, P1{}, NO{}, NO{});
          it++;
          step(std::true_type{}, std::true_type{}, P0{}, NO{}, NO{});
          it++;
        }
        tstamp(std::integral_constant<int, 3>{});
        step(NO{}, NO{}, P1{}, std::true_type{}, NO{}); // nothing prefetched, no barrier; blocks 0 .. MA-2 recombined and stored
        it++;
        tstamp(std::integral_constant<int, 4>{});
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // last MFMA -> VALU reads of its accumulators
        static_for<SPA>([&](auto tc) { // the last block's stream: the same steps, back to back
          epi_stream(std::integral_constant<int, MA - 1>{}, tc, std::integral_constant<int, OZ_CL_P9_ACC>{}, std::integral_constant<int, -1>{},
                     std::integral_constant<int, -1>{}, af0);
          epi_slot(std::integral_constant<int, MA - 1>{}, tc);
          __builtin_amdgcn_sched_barrier(0);
        });
        tflush(true);
        return;
      }
    }
    if constexpr (ZERO_BY_FIRST_STEP) {
      step(std::true_type{}, std::true_type{}, P0{}, NO{}, ZF{});
      it++;
      step(std::true_type{}, std::true_type{}, P1{}, NO{}, NO{});
      it++;
    }
    for (; it < nk;) {
      step(std::true_type{}, std::true_type{}, P0{}, NO{}, NO{});
      it++;
      step(std::true_type{}, std::true_type{}, P1{}, NO{}, NO{});
      it++;
    }
  } else {
    using P0 = std::integral_constant<int, 0>;
    using NO = std::false_type;
    for (; it + PD < nk; it++) step(std::true_type{}, std::true_type{}, P0{}, NO{}, NO{});
    if constexpr (PD > 1)
      for (; it + 1 < nk; it++) step(std::false_type{}, std::true_type{}, P0{}, NO{}, NO{});
    for (; it < nk; it++) step(std::false_type{}, std::false_type{}, P0{}, NO{}, NO{});
  }

  tstamp(std::integral_constant<int, 3>{}); // (the plain form: the whole recombination lies behind the k loop, 3 = 4)
  tstamp(std::integral_constant<int, 4>{});
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // last MFMA -> VALU reads of its accumulator
  auto acc = [&](int a, int b, int d, int v) -> int {
    const int x = (a * 2 + b) * ND + d;
    if constexpr (ACCN) return read_accn<CLE>(x, v);
    if (x >= 64) return accV[x >= 64 ? x - 64 : 0][v];
    int r;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(accA[x < 64 ? x : 0][v]));
    return r;
  };
  if constexpr ((VARW & VARW_NO_EPILOGUE) != 0) {
    static_assert(!ACCN, "named accumulators need no keep-alive");
#pragma unroll
    for (int x = 0; x < NACC_A; x++) asm volatile("" ::"a"(accA[x]));
#pragma unroll
    for (int x = 0; x < (NACC > 64 ? NACC_V : 0); x++) asm volatile("" ::"v"(accV[x]));
    return;
  }
  recombine_and_store16<D0, ND, MA, (VARW >> 8) & 3>(p, acc, rb0 * 32, tn * 128 + wave * 32);
  tflush(false);
#undef YC
}

} // namespace ozhip
