// slice_gemm_x_tile.h — the "paired" tile of the wide slice GEMM: v_mfma_i32_16x16x64_i8 with TWO slice products of one
// diagonal per instruction.  Included by slice_gemm_w_kernel.h (same persistent workgroups, same staging, same LDS
// image; VARW_X16 selects this tile function instead of w_tile).
//
// Why (profiles/r3_ablate/r3a_mfma_shape_and_karatsuba_probe.txt): the part is power limited under full-entropy INT8
// MFMA, and the 16x16x64 shape does 4x fewer 32-bit accumulator updates per MAC than 32x32x32: 3.9 vs 3.3 POPS sustained
// on random operands (+18 %), accumulators pinned in place.  Its operand is "lane l: 16 k-bytes of row l & 15, k-group
// l >> 4" - 64 k per instruction, twice what a k-step of the staged image (32 k: 160 KiB of LDS hold two stages of
// a 96 x 128 tile at S = 9, not two of 64 k) offers.  So the two upper k-groups carry a SECOND slice pair of the same
// diagonal over the same 32 k:
//
//     A operand  PA_i = [ A_i | A_{i+1} ]      (k-groups 0,1 = slice i, k-groups 2,3 = slice i+1; 16 rows)
//     B operand  PB_q = [ B_{2q+1} | B_{2q} ]  (16 columns)
//     PA_i x PB_q  =  A_i B_{2q+1} + A_{i+1} B_{2q}            both on diagonal t = i + 2q + 1
//     ZA   x PB_q  =  A_0 B_{2q},  ZA = [ 0 | A_0 ]             the product (0, 2q) that has no partner, t = 2q
//
// Every product (i, j), i + j = t, is computed exactly once: j odd -> low half of PA_i x PB_(j-1)/2; j even, i >= 1 ->
// high half of PA_{i-1} x PB_{j/2}; j even, i = 0 -> ZA x PB_{j/2}.  With f = 0 for ZA and f = i + 1 for PA_i the
// diagonal of (f, q) is t = f + 2q.  S = 9: 20 full + 5 half-empty instructions per 16x16 block and k-step for 45
// products (the zero half costs an issue slot, little energy).  INT32 sums are exact, so the result is bit-identical to
// the other kernels'.
//
// Fragments come straight out of the unchanged 1 KiB blocks [k-half][row & 31][16 B] (layout.h) with per-lane addresses:
//     lane l, g = l >> 4, r = l & 15:   vA = (g >> 1) * 1024 + (g & 1) * 512 + r * 16     (A: slice i, then i + 1)
//                                       vB = (g < 2) * 1024  + (g & 1) * 512 + r * 16     (B: slice 2q + 1, then 2q)
// plus (row-block * SL + slice) KiB and 256 B for the upper 16 rows of a block.  512 B and 1 KiB are multiples of the
// 256 B bank period, so each 16-lane group of a ds_read_b128 still covers all 64 banks once.  A wave owns 2*WA x 2
// blocks of 16 x 16 (the same (32*WA) x 32 outputs as in w_tile); the B pair fragments of both column blocks live in
// registers for a k-step (2 * NQ * 4 registers, NQ = (SL + 1) / 2), the A fragments stream through a register ring.
#pragma once

namespace ozhip {

__device__ __forceinline__ void mfma16_agpr(v4i &c, const v4i &b, const v4i &a) {
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
}
__device__ __forceinline__ void mfma16_vgpr(v4i &c, const v4i &b, const v4i &a) {
  asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(b), "v"(a));
}
// B operand in the named registers v[REG : REG + 3] (VARW_BREG, slice_gemm_w_kernel.h)
template <int REG>
__device__ __forceinline__ void mfma16_agpr_named(v4i &c, const v4i &a) {
  asm volatile("v_mfma_i32_16x16x64_i8 %0, v[%c2:%c3], %1, %0" : "+a"(c) : "v"(a), "i"(REG), "i"(REG + 3) : OZ_BREG_CLOBBERS);
}
template <int REG>
__device__ __forceinline__ void mfma16_vgpr_named(v4i &c, const v4i &a) {
  asm volatile("v_mfma_i32_16x16x64_i8 %0, v[%c2:%c3], %1, %0" : "+v"(c) : "v"(a), "i"(REG), "i"(REG + 3) : OZ_BREG_CLOBBERS);
}
// the same behind a VALU write of an operand (ZA's v_cndmask): hipcc pads nothing in front of an asm statement, the two
// wait states a VALU result needs before an MFMA reads it as A/B are inside the string (guide 5.7 item 2)
__device__ __forceinline__ void mfma16_agpr_after_valu(v4i &c, const v4i &b, const v4i &a) {
  asm volatile("s_nop 1\n\tv_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
}
__device__ __forceinline__ void mfma16_vgpr_after_valu(v4i &c, const v4i &b, const v4i &a) {
  asm volatile("s_nop 1\n\tv_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(b), "v"(a));
}

// MFMA slots of one k-step: 16-row block a outermost, A fragment f ascending (0 = ZA, f = i + 1: PA_i), B pair q
// descending, column block b innermost.  A "group" is the run of slots that share one A fragment.
template <int S, int D0, int ND, int WA>
struct XSched {
  static constexpr int SL = (D0 + ND < S) ? (D0 + ND) : S;
  static constexpr int MA = 2 * WA;
  static constexpr int NQ = (SL - 1) / 2 + 1;
  static constexpr int MAXG = MA * SL, MAXS = MA * SL * NQ * 2;
  int ns = 0, ng = 0;
  int sa[MAXS] = {}, sf[MAXS] = {}, sq[MAXS] = {}, sb[MAXS] = {}, sg[MAXS] = {};
  int gfirst[MAXG] = {}, g_a[MAXG] = {}, g_f[MAXG] = {};
  constexpr XSched() {
    for (int a = 0; a < MA; a++)
      for (int f = 0; f < SL; f++) {
        bool any = false;
        for (int q = NQ - 1; q >= 0; q--) {
          const int t = f + 2 * q;
          if (t < D0 || t >= D0 + ND || t > S - 1) continue;
          for (int b = 0; b < 2; b++) {
            if (!any) {
              gfirst[ng] = ns;
              g_a[ng] = a;
              g_f[ng] = f;
              any = true;
            }
            sa[ns] = a; sf[ns] = f; sq[ns] = q; sb[ns] = b; sg[ns] = ng;
            ns++;
          }
        }
        if (any) ng++;
      }
  }
  constexpr int max_q_from(int s0) const {
    int m = 0;
    for (int s = s0; s < ns; s++) m = sq[s] > m ? sq[s] : m;
    return m;
  }
  constexpr int min_q_used() const {
    int m = NQ;
    for (int s = 0; s < ns; s++) m = sq[s] < m ? sq[s] : m;
    return m;
  }
};

template <int S, int D0, int ND, int WA>
inline constexpr XSched<S, D0, ND, WA> kXSched{};

// ---- epilogue for the 16x16 C/D layout: lane l holds column-major row m = l & 15 of a block, registers v = 0..3 are
// the columns n = 4 * (l >> 4) + v.  acc(a, b, d, v): register v of diagonal accumulator d of block (a, b), a = 16-row
// block 0..MA-1, b = 16-column block 0..1.  mu / nu: first row / column of the wave's (16*MA) x 32 outputs.  Same
// arithmetic, element by element, as recombine_and_store (slice_gemm_kernel.h): one fma per diagonal, t ascending, then
// x / 2^44 * eA[m] * eB[n], alpha, beta.  The row blocks are processed MA/2 at a time (the register budget of the
// 32x32 form: the accumulators are all still live).
template <int D0, int ND, int MA, int EPI = 0, class Acc>
__device__ __forceinline__ void recombine_and_store16(const SliceGemmArgs &p, const Acc &acc, uint32_t mu, uint32_t nu) {
  static_assert(MA % 2 == 0, "16-row blocks come in pairs (one 32-row fragment block)");
  constexpr int HA = MA / 2;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t m0 = mu + (lane & 15u);
  const uint32_t nl = 4u * (lane >> 4);
#ifdef OZIMMU_HIP_TEST_HOOKS
  if (p.dump) { // test hook: raw INT32 diagonal sums, [ND][N][M]
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const uint32_t n = nu + 16 * b + nl + v;
#pragma unroll
        for (int a = 0; a < MA; a++) {
          const uint32_t m = m0 + 16 * a;
          if (m < p.M && n < p.N)
#pragma unroll
            for (int d = 0; d < ND; d++) p.dump[((size_t)d * p.N + n) * p.M + m] = acc(a, b, d, v);
        }
      }
    if (p.dump_only) return;
  }
#endif
  double sc[ND];
#pragma unroll
  for (int d = 0; d < ND; d++) sc[d] = pow2d(46 - p.L * (D0 + d + 2));
  double ea[MA];
#pragma unroll
  for (int a = 0; a < MA; a++) ea[a] = (m0 + 16 * a < p.M) ? p.ea[m0 + 16 * a] : 0.0;

  // ---- interior tiles of a real GEMM: 16-byte stores through the lane-pair exchange (see recombine_and_store) ----
  if (p.final && !p.cplx && mu + 16u * MA <= p.M && nu + 32u <= p.N && (p.ldc & 1u) == 0 && p.ldc < (1u << 26) &&
      (reinterpret_cast<uintptr_t>(p.c) & 15u) == 0) {
    const bool odd = (lane & 1u) != 0;
    const uint32_t boff = ((nl + (lane & 1u)) * (uint32_t)p.ldc + (lane & 14u)) * 8u;
    const double *eb_lane = p.eb + nu + nl;
    const bool rmw = p.beta != 0.0;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int vp = 0; vp < 2; vp++)
#pragma unroll
        for (int ah = 0; ah < 2; ah++) {
          __builtin_amdgcn_sched_barrier(0);
          constexpr int dummy = 0;
          (void)dummy;
          const uint32_t cofs = 16 * b + 2 * vp; // column of register 2vp inside the wave's 32 columns (+ nl per lane)
          char *colp = reinterpret_cast<char *>(p.c + ((size_t)(nu + cofs) * p.ldc + mu)) + 128 * HA * ah; // wave-uniform
          double2 old[HA];
          if (rmw) {
#pragma unroll
            for (int a = 0; a < HA; a++) old[a] = *reinterpret_cast<const double2 *>(colp + (size_t)boff + 128 * a);
          }
          const double eb0 = eb_lane[cofs], eb1 = eb_lane[cofs + 1];
          double x0[HA], x1[HA];
#pragma unroll
          for (int a = 0; a < HA; a++) x0[a] = x1[a] = 0.0;
          if (p.acc_in) { // later diagonal pass / K chunk: continue the chain the previous launch left in the workspace
            const double *ap = p.acc + ((size_t)(nu + nl + cofs) * p.M + m0 + 16 * HA * ah);
#pragma unroll
            for (int a = 0; a < HA; a++) x0[a] = ap[16 * a], x1[a] = ap[p.M + 16 * a];
          }
#pragma unroll
          for (int d = 0; d < ND; d++)
#pragma unroll
            for (int a = 0; a < HA; a++) {
              x0[a] = fma((double)acc(HA * ah + a, b, d, 2 * vp), sc[d], x0[a]);
              x1[a] = fma((double)acc(HA * ah + a, b, d, 2 * vp + 1), sc[d], x1[a]);
            }
#pragma unroll
          for (int a = 0; a < HA; a++) {
            // reference: x_ptr[tid] / (1l << 44) * a_max_exp[mi] * b_max_exp[ni]  (src/gemm.cu:140-141)
            const double e = ea[HA * ah + a];
            const double v0 = x0[a] * 0x1p-44 * e * eb0, v1 = x1[a] * 0x1p-44 * e * eb1;
            const double s0 = lane_pair_swap(v0), s1 = lane_pair_swap(v1);
            double2 y;
            y.x = odd ? s1 : v0; // row 2i:     (column n from this lane | column n+1 from the even neighbour)
            y.y = odd ? v1 : s0; // row 2i + 1: (column n from the odd neighbour | column n+1 from this lane)
            if (rmw) {
              y.x = fma(p.alpha, y.x, p.beta * old[a].x);
              y.y = fma(p.alpha, y.y, p.beta * old[a].y);
            } else {
              y.x = p.alpha * y.x;
              y.y = p.alpha * y.y;
            }
            if constexpr ((EPI & 1) != 0) {
              if (y.x != 0x1.23456789p-900) continue; // never true for real data; keeps the chain alive
            }
            *reinterpret_cast<double2 *>(colp + (size_t)boff + 128 * a) = y;
          }
        }
    return;
  }
  // ---- interior tiles of one of the four real products of a ZGEMM (interleaved complex C, 16 bytes per element) ----
  if (p.final && p.cplx && mu + 16u * MA <= p.M && nu + 32u <= p.N && p.ldc < (1u << 25)) {
    const uint32_t boff = (nl * (uint32_t)p.ldc + (lane & 15u)) * 16u;
    const double *eb_lane = p.eb + nu + nl;
    const size_t col_bytes = p.ldc * 16u;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int vp = 0; vp < 2; vp++)
#pragma unroll
        for (int ah = 0; ah < 2; ah++) {
          __builtin_amdgcn_sched_barrier(0);
          const uint32_t cofs = 16 * b + 2 * vp;
          char *colp = reinterpret_cast<char *>(reinterpret_cast<double2 *>(p.c) + ((size_t)(nu + cofs) * p.ldc + mu)) +
                       256 * HA * ah;
          double2 old0[HA], old1[HA];
#pragma unroll
          for (int a = 0; a < HA; a++) {
            old0[a] = *reinterpret_cast<const double2 *>(colp + (size_t)boff + 256 * a);
            old1[a] = *reinterpret_cast<const double2 *>(colp + col_bytes + (size_t)boff + 256 * a);
          }
          const double eb0 = eb_lane[cofs], eb1 = eb_lane[cofs + 1];
          double x0[HA], x1[HA];
#pragma unroll
          for (int a = 0; a < HA; a++) x0[a] = x1[a] = 0.0;
          if (p.acc_in) {
            const double *ap = p.acc + ((size_t)(nu + nl + cofs) * p.M + m0 + 16 * HA * ah);
#pragma unroll
            for (int a = 0; a < HA; a++) x0[a] = ap[16 * a], x1[a] = ap[p.M + 16 * a];
          }
#pragma unroll
          for (int d = 0; d < ND; d++)
#pragma unroll
            for (int a = 0; a < HA; a++) {
              x0[a] = fma((double)acc(HA * ah + a, b, d, 2 * vp), sc[d], x0[a]);
              x1[a] = fma((double)acc(HA * ah + a, b, d, 2 * vp + 1), sc[d], x1[a]);
            }
#pragma unroll
          for (int a = 0; a < HA; a++) {
            const double e = ea[HA * ah + a];
            const double v0 = x0[a] * 0x1p-44 * e * eb0, v1 = x1[a] * 0x1p-44 * e * eb1;
            // C += (alpha_re + i alpha_im) * v  (axy_complex_kernel, src/gemm.cu:160-186)
            double2 y0 = old0[a], y1 = old1[a];
            y0.x = fma(p.alpha, v0, y0.x);
            y0.y = fma(p.alpha_im, v0, y0.y);
            y1.x = fma(p.alpha, v1, y1.x);
            y1.y = fma(p.alpha_im, v1, y1.y);
            *reinterpret_cast<double2 *>(colp + (size_t)boff + 256 * a) = y0;
            *reinterpret_cast<double2 *>(colp + col_bytes + (size_t)boff + 256 * a) = y1;
          }
        }
    return;
  }
  // ---- edges, odd ldc, misaligned C, non-final passes: element by element -------------------------------------
  constexpr int CG = 2;
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int vp = 0; vp < 2; vp++)
#pragma unroll
      for (int ah = 0; ah < 2; ah++) {
        __builtin_amdgcn_sched_barrier(0);
        uint32_t n[CG];
        bool ok[CG][HA];
        double x[CG][HA], ebn[CG];
#pragma unroll
        for (int rr = 0; rr < CG; rr++) {
          n[rr] = nu + 16 * b + nl + 2 * vp + rr;
          ebn[rr] = (p.final && n[rr] < p.N) ? p.eb[n[rr]] : 0.0;
#pragma unroll
          for (int a = 0; a < HA; a++) {
            const uint32_t m = m0 + 16 * (HA * ah + a);
            ok[rr][a] = n[rr] < p.N && m < p.M;
            x[rr][a] = (ok[rr][a] && p.acc_in) ? p.acc[(size_t)n[rr] * p.M + m] : 0.0;
          }
        }
#pragma unroll
        for (int d = 0; d < ND; d++)
#pragma unroll
          for (int rr = 0; rr < CG; rr++)
#pragma unroll
            for (int a = 0; a < HA; a++) {
              if constexpr ((EPI & 2) != 0)
                x[rr][a] = __hiloint2double(acc(HA * ah + a, b, d, 2 * vp + rr), __double2hiint(x[rr][a]) ^ __double2loint(x[rr][a]));
              else
                x[rr][a] = fma((double)acc(HA * ah + a, b, d, 2 * vp + rr), sc[d], x[rr][a]);
            }
#pragma unroll
        for (int rr = 0; rr < CG; rr++)
#pragma unroll
          for (int a = 0; a < HA; a++) {
            if (!ok[rr][a]) continue;
            if constexpr ((EPI & 1) != 0) {
              if (x[rr][a] != 0x1.23456789p-900) continue;
            }
            const uint32_t m = m0 + 16 * (HA * ah + a);
            if (!p.final) {
              p.acc[(size_t)n[rr] * p.M + m] = x[rr][a];
              continue;
            }
            const double v = x[rr][a] * 0x1p-44 * ea[HA * ah + a] * ebn[rr];
            if (p.cplx) {
              double2 *zp = reinterpret_cast<double2 *>(p.c) + ((size_t)n[rr] * p.ldc + m);
              double2 y = *zp;
              y.x = fma(p.alpha, v, y.x);
              y.y = fma(p.alpha_im, v, y.y);
              *zp = y;
            } else if (p.beta != 0.0) {
              double *cp = p.c + (size_t)n[rr] * p.ldc + m;
              *cp = fma(p.alpha, v, p.beta * *cp);
            } else {
              p.c[(size_t)n[rr] * p.ldc + m] = p.alpha * v;
            }
          }
      }
}

constexpr int X_PAD = FRAG_BYTES; // LDS: one unused KiB in front of the A stages (ZA reads "slice -1" in its masked lanes)
                                  // and one behind the B buffers (PB_(NQ-1) reads "slice SL" when SL is odd)

// One output tile of (32*WA) x 128: rows start at A row-block rb0, columns at B row-block 4*tn.  Same staging, buffers,
// copy schedule and step structure as w_tile; RING = A fragment ring entries.
template <int S, int D0, int ND, int WA, int VARW, int STAG, int DMA0_, int DMAE_, int TAIL_, int RING = 4>
__device__ __forceinline__ void x_tile(const SliceGemmArgs &p, char *smem, const uint32_t rb0, const uint32_t tn,
                                       const uint32_t xcd) {
#define XC (kXSched<S, D0, ND, WA>)
  constexpr int SL = XC.SL, MA = XC.MA, NQ = XC.NQ;
  // two A buffers (prefetch distance 1); two wave-private B buffers, or ONE (VARW_B1): a wave holds ALL B pair fragments of
  // a k-step in registers from the end of the previous step on, so the next stage's B may land in the buffer the current
  // one came from (w_tile has the same form): the second pass of S = 14..18 stages 14-18 slices
  constexpr int NA = 2, PD = 1;
  constexpr int NB = (VARW & VARW_B1) ? 1 : 2;
  static_assert((VARW & VARW_NA3) == 0, "paired tile: two A buffers (prefetch distance 1)");
  constexpr int A_STAGE = WA * SL * FRAG_BYTES;
  constexpr int B_STAGE = 4 * SL * FRAG_BYTES;
  constexpr int OFF_B = NA * A_STAGE;
  constexpr int NQA = (WA * SL + 3) / 4;
  constexpr int NDMA = NQA + SL;
  constexpr int R = RING;
  constexpr int NG = MA * SL;
  constexpr bool NO_GLOBAL = (VARW & (VARW_NO_GLOBAL | VARW_MFMA_ONLY)) != 0;
  constexpr bool MFMA_ONLY = (VARW & VARW_MFMA_ONLY) != 0;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  // ---- staging: identical to w_tile, the stages start X_PAD bytes into the allocation ----------------------------
  const size_t rb_stride = (size_t)p.KB * (size_t)(S * FRAG_BYTES);
  const uint32_t rba_last = p.rba - 1u;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const size_t pass0 = (size_t)p.kb0 * (S * FRAG_BYTES);
  const int8_t *a_src[NQA];
  uint32_t a_lds[NQA];
#pragma unroll
  for (int t = 0; t < NQA; t++) {
    uint32_t q = (uint32_t)(wave * NQA + t);
    if (q > (uint32_t)(WA * SL - 1)) q = WA * SL - 1;
    const uint32_t a = q / SL, s = q - a * SL;
    uint32_t rb = rb0 + a;
    if (rb > rba_last) rb = rba_last;
    a_src[t] = uniform_ptr(p.a_planes + rb * rb_stride + s * FRAG_BYTES + pass0);
    a_lds[t] = q * FRAG_BYTES;
  }
  const int8_t *b_src = uniform_ptr(p.b_planes + (size_t)(4u * tn + wave) * rb_stride + pass0);
  const uint32_t lds0 = (uint32_t)(size_t)((OZ_AS3 char *)smem) + X_PAD;
  const uint32_t ldsb0 = lds0 + OFF_B + wave * (SL * FRAG_BYTES);
  auto copy_a = [&](int t, uint32_t voff, uint32_t lds_a) {
    if constexpr (NO_GLOBAL) return;
    glds16<0>(a_src[t], voff, lds_a, a_lds[t]);
  };
  auto copy_b = [&](auto sc, uint32_t voff, uint32_t lds_b) {
    if constexpr (NO_GLOBAL) return;
    constexpr int s = decltype(sc)::value;
    constexpr int G = 4;
    constexpr int g0 = s / G * G;
    glds16<(s % G) * FRAG_BYTES>(b_src + g0 * FRAG_BYTES, voff, lds_b, (uint32_t)(g0 * FRAG_BYTES));
  };
  auto copy_n = [&](auto cc, int abuf, int bbuf, uint32_t kstep) {
    constexpr int c = decltype(cc)::value;
    const uint32_t voff = lane_off + kstep * (uint32_t)(S * FRAG_BYTES);
    if constexpr (c < NQA) {
      copy_a(c, voff, lds0 + abuf * A_STAGE);
    } else {
      // one B buffer: the copy overwrites what this step's B fragments were read from; those reads were issued a step's
      // tail ago, the wait makes "they have returned" a guarantee
      if constexpr (NB == 1 && c == NQA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      copy_b(std::integral_constant<int, c - NQA>{}, voff, ldsb0 + bbuf * B_STAGE);
    }
  };

  // ---- accumulators: MA x 2 x ND tuples of 4 registers; the first 64 in the AGPR half ---------------------------
  constexpr int NACC = MA * 2 * ND, NACC_A = NACC < 64 ? NACC : 64, NACC_V = NACC > 64 ? NACC - 64 : 1;
  v4i accA[NACC_A], accV[NACC_V];
#pragma unroll
  for (int x = 0; x < NACC_A; x++)
#pragma unroll
    for (int r = 0; r < 4; r++) accA[x][r] = 0;
#pragma unroll
  for (int x = 0; x < NACC_V; x++)
#pragma unroll
    for (int r = 0; r < 4; r++) accV[x][r] = 0;
  auto mfma = [&](auto xc, auto after_valu, const v4i &b, const v4i &a) {
    constexpr int X = decltype(xc)::value;
    constexpr bool AV = decltype(after_valu)::value;
    if constexpr (X < 64) {
      if constexpr (AV) mfma16_agpr_after_valu(accA[X], b, a);
      else mfma16_agpr(accA[X], b, a);
    } else {
      if constexpr (AV) mfma16_vgpr_after_valu(accV[X - 64], b, a);
      else mfma16_vgpr(accV[X - 64], b, a);
    }
  };

  // ---- circular K with the per-XCD phase hint (as w_tile) ---------------------------------------------------------
  const uint32_t nk = p.kb1 - p.kb0;
  uint32_t *phase = p.phase ? p.phase + (uint32_t)PHASE_LINE_WORDS * xcd : nullptr;
  uint32_t koff = 0;
  if (phase && nk > p.phase_min_kb) { // (see w_tile)
    if (threadIdx.x == 0)
      *(volatile uint32_t *)smem = __hip_atomic_load(phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    koff = (*(volatile uint32_t *)smem + 2u) % nk;
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

  // ---- schedule of one k-step (compile time; see w_tile for the event model) --------------------------------------
  constexpr int NS = XC.ns;
  constexpr int NRP = (NG - 1 + R - 1) / R * R;
  constexpr int TAIL = TAIL_ < NS ? TAIL_ : NS - 1;
  constexpr int XS = NS - TAIL;
  constexpr int DMA0 = DMA0_ >= 0 ? DMA0_ : XC.gfirst[1 < NG ? 1 : 0];
  constexpr int DMAE_FIT = NDMA > 1 ? (XS - 1 - DMA0) / (NDMA - 1) : 1;
  constexpr int DMAE = DMAE_ < DMAE_FIT ? DMAE_ : (DMAE_FIT > 1 ? DMAE_FIT : 1);
  constexpr int JT = XC.max_q_from(XS) + 1; // B pairs q < JT are still needed after the barrier slot
  static_assert(XC.ng == NG, "empty (a, f) groups are not supported by the ring arithmetic");
  static_assert(NRP + 2 - R > NG - 1 || XC.gfirst[NRP + 2 - R <= NG - 1 && NRP + 2 - R >= 0 ? NRP + 2 - R : 0] >= XS,
                "the first ring read of the next stage must come after the barrier");
  static_assert(NG >= 2, "at least two groups per k-step");
  static_assert(DMA0 + (NDMA - 1) * DMAE < XS, "every copy of a stage is issued before the barrier slot");

  const int g4 = lane >> 4;
  const uint32_t vA = (uint32_t)((g4 >> 1) * FRAG_BYTES + (g4 & 1) * 512 + (lane & 15) * 16);
  const uint32_t vB = (uint32_t)((g4 < 2 ? FRAG_BYTES : 0) + (g4 & 1) * 512 + (lane & 15) * 16);
  // la0 + (row-block * SL + f) KiB + 256 * (upper 16 rows): fragment f of a block (f = 0: ZA, its lanes 32..63 read slice 0)
  const char *la0 = smem + X_PAD - FRAG_BYTES + vA;
  const char *lb0 = smem + X_PAD + OFF_B + wave * (SL * FRAG_BYTES) + vB;
  const bool upper = lane >= 32; // the lanes of ZA that carry A_0
  int abuf = 0, bbuf = 0;
  v4i bq[2][NQ], af[R], af0;
  auto read_a = [&](auto gc, const char *la) { // A fragment of group g -> af0 (g == 0) or ring slot (g - 1) % R
    constexpr int g = decltype(gc)::value;
    const v4i f = *(const v4i *)(la + ((XC.g_a[g] >> 1) * SL + XC.g_f[g]) * FRAG_BYTES + (XC.g_a[g] & 1) * 256);
    if constexpr (g == 0)
      af0 = f;
    else
      af[(g - 1) % R] = f;
  };
  auto zero_low_half = [&](v4i &f) { // ZA = [0 | A_0]: lanes 0..31 addressed "slice -1"
#pragma unroll
    for (int c = 0; c < 4; c++) f[c] = upper ? f[c] : 0;
  };
  auto read_b = [&](int b, int q, const char *lb) { bq[b][q] = *(const v4i *)(lb + q * (2 * FRAG_BYTES) + b * 256); };

  // ---- prologue -----------------------------------------------------------------------------------------------
  uint32_t k_issue = koff;
  if (0u < nk) {
    static_for<NDMA>([&](auto cc) { copy_n(cc, 0, 0, k_issue); });
    k_issue = koff_next(k_issue);
  }
  if constexpr (!NO_GLOBAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (!MFMA_ONLY) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int q = 0; q < NQ; q++) read_b(b, q, lb0);
    static_for<(R < NG ? R : NG)>([&](auto gc) { read_a(gc, la0); });
  }
  asm volatile("s_nop 7" ::: "memory"); // zero-fill -> first MFMA reading it as C

  uint32_t it = 0;
  auto stamp = [&]() -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0" : "=s"(t));
    return t;
  };
  auto step = [&](auto pf_tag, auto nx_tag) {
    constexpr bool PF = decltype(pf_tag)::value, NX = decltype(nx_tag)::value;
    constexpr bool TRACE = (VARW & VARW_TRACE) != 0;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (TRACE) ts[5] = stamp();
    const int abuf_n = abuf ^ 1;
    const int abuf_pf = abuf_n, bbuf_pf = NB == 1 ? 0 : (bbuf ^ 1);
    const uint32_t kb_pf = k_issue;
    if constexpr (PF) k_issue = koff_next(k_issue);
    const char *la = la0 + abuf * A_STAGE;
    const char *la_n = la0 + abuf_n * A_STAGE;
    const char *lb_n = lb0 + (NB == 1 ? 0 : (bbuf ^ 1) * B_STAGE);
    static_for<NS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int g = XC.sg[s];
      if constexpr (s == XS && NX && !MFMA_ONLY) {
        if constexpr (TRACE) ts[0] = stamp();
        if constexpr (!NO_GLOBAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (TRACE) ts[1] = stamp();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the buffer this barrier releases is refilled right behind it
        if constexpr (TRACE) ts[2] = stamp();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (TRACE) ts[3] = stamp();
        if constexpr (STAG > 0)
          for (int q = 0; q < wave; q++) asm volatile("s_nop %0" ::"n"(STAG - 1));
        read_a(std::integral_constant<int, 0>{}, la_n);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (XC.gfirst[g] == s && g >= 1 && !MFMA_ONLY) {
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
        if constexpr (TRACE && s == DMA0 + 4 * DMAE) ts[6] = stamp();
        copy_n(std::integral_constant<int, (s - DMA0) / DMAE>{}, abuf_pf, bbuf_pf, kb_pf);
        if constexpr (TRACE && s == DMA0 + 4 * DMAE) ts[7] = stamp();
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int a = XC.sa[s], f = XC.sf[s], q = XC.sq[s], b = XC.sb[s];
      constexpr int X = (a * 2 + b) * ND + (f + 2 * q - D0);
      if constexpr (MFMA_ONLY) {
        mfma(std::integral_constant<int, X>{}, std::false_type{}, cf[(2 * q) % SL], cf[f]);
      } else {
        if constexpr (f == 0 && XC.gfirst[g] == s) { // ZA: clear the half that pairs with B_{2q+1}
          if constexpr (g == 0)
            zero_low_half(af0);
          else
            zero_low_half(af[(g >= 1 ? g - 1 : 0) % R]);
        }
        mfma(std::integral_constant<int, X>{}, std::integral_constant<bool, (f == 0 && XC.gfirst[g] == s)>{}, bq[b][q],
             g == 0 ? af0 : af[(g >= 1 ? g - 1 : 0) % R]);
      }
      if constexpr (s >= XS && NX && !MFMA_ONLY) {
        constexpr int NREF = 2 * (NQ - JT);                 // fragments refreshed behind the barrier
        constexpr int RPT = (NREF + TAIL - 1) / TAIL;
#pragma unroll
        for (int u = 0; u < RPT; u++) {
          const int idx = (s - XS) * RPT + u;               // (q descending from NQ-1, b)
          if (idx < NREF) read_b(idx & 1, NQ - 1 - (idx >> 1), lb_n);
        }
        if constexpr (s == XS + 1 || (TAIL == 1 && s == XS))
          if (phase && (it & 3u) == 0 && threadIdx.x == 0)
            __hip_atomic_store(phase, koff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NX && !MFMA_ONLY) {
#pragma unroll
      for (int q = JT - 1; q >= 0; q--)
#pragma unroll
        for (int b = 0; b < 2; b++) read_b(b, q, lb_n);
      static_for<R - 1>([&](auto qc) {
        constexpr int q = decltype(qc)::value + 1;
        if constexpr (NRP + q - R + 1 > NG - 1 && q < NG) read_a(std::integral_constant<int, q>{}, la_n);
      });
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (TRACE) {
      ts[4] = stamp();
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+s"(ts[0]), "+s"(ts[1]), "+s"(ts[2]), "+s"(ts[3]), "+s"(ts[4]), "+s"(ts[5]), "+s"(ts[6]), "+s"(ts[7]));
      if (blockIdx.x < 32 && it >= 100 && it < 108 && lane == 0) {
        uint32_t *tr = reinterpret_cast<uint32_t *>(p.acc) + ((blockIdx.x * 4 + wave) * 8 + (it - 100)) * 8;
#pragma unroll
        for (int q = 0; q < 8; q++) tr[q] = (uint32_t)ts[q];
      }
    }
    koff = koff_next(koff);
    abuf = abuf_n;
    bbuf ^= 1;
  };
  for (; it + PD < nk; it++) step(std::true_type{}, std::true_type{});
  for (; it + 1 < nk; it++) step(std::false_type{}, std::true_type{});
  for (; it < nk; it++) step(std::false_type{}, std::false_type{});

  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // last MFMA -> VALU reads of its accumulator
  auto acc = [&](int a, int b, int d, int v) -> int {
    const int x = (a * 2 + b) * ND + d;
    if (x >= 64) return accV[x >= 64 ? x - 64 : 0][v];
    int r;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(accA[x < 64 ? x : 0][v]));
    return r;
  };
  if constexpr ((VARW & VARW_NO_EPILOGUE) != 0) {
#pragma unroll
    for (int x = 0; x < NACC_A; x++) asm volatile("" ::"a"(accA[x]));
#pragma unroll
    for (int x = 0; x < (NACC > 64 ? NACC_V : 0); x++) asm volatile("" ::"v"(accV[x]));
    return;
  }
  recombine_and_store16<D0, ND, MA, (VARW >> 8) & 3>(p, acc, rb0 * 32, tn * 128 + wave * 32);
#undef XC
}

} // namespace ozhip
