// split_resident.h — device side of the resident split (one read of an operand strip, the strip held in the registers of one
// workgroup) and of the cut itself, shared by split.hip (split_resident_kernel, the streaming cut kernels) and by
// slice_gemm_one_launch.hip (split + slice GEMM of a small problem in ONE launch).
// Restates cut_int8_core / split_int8_kernel, /root/reference/src/split.cu:154-242.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.h"
#include "layout.h"

namespace ozhip {

__device__ __forceinline__ unsigned exp_field(double x) {
  return (unsigned)((unsigned long long)__double_as_longlong(x) >> 52) & 0x7FFu;
}

// ---- the cut itself: 16 k values of one row -> S x 16 slice bytes (src/split.cu:154-185) ------------------------------
// e = the row's maximum exponent field.  e == 0: zero/subnormal row -> max_exp 0, zero slices (reference: 0*2 = 0,
// src/split.cu:191); e >= 0x7FE: Inf/NaN in the row, or 2^(e+1) not representable -> poisoned row (max_exp = NaN).
// exponent field of a tagged exponent word (SplitJobs::tag); a word left by an earlier epoch reads as "nothing seen"
__device__ __forceinline__ unsigned tagged_exp(uint32_t w, uint32_t tag) { return (w & ~0x7FFu) == tag ? (w & 0x7FFu) : 0u; }

__device__ __forceinline__ double max_exp_of(unsigned e) {
  const bool live = e != 0u && e < 0x7FEu;
  const unsigned long long bits = live ? ((unsigned long long)(e + 1) << 52) : (e == 0u ? 0ull : 0x7FF8000000000000ull);
  return __longlong_as_double((long long)bits);
}

// out: this lane's 16 bytes of slice 0 inside the fragment block run (slice s follows FRAG_BYTES * s later)
// LOW = false: the caller guarantees S * L <= 64, i.e. every slice lies in the upper two words of the 128-bit value (all
// modes up to fp64_int8_9 at L = 7): the lower words are neither built nor kept (32 registers and a third of the selects)
// WT = true: the stores are write-through at system scope (sc0 sc1: the data is in memory, not in this XCD's L2, when the
// store is acknowledged) - for consumers on other XCDs INSIDE the same kernel (slice_gemm_one_launch.hip)
// FPBUILD (with LOW = false): the two words by FP64 scaling instead of integer shifts; false = the round-5 form (A/B, parity arm)
template <bool LOW = true, bool WT = false, bool FPBUILD = true>
__device__ __forceinline__ void cut_and_store(const double (&v)[16], unsigned e, int S, int L, int8_t *out) {
  const bool live = e != 0u && e < 0x7FEu;
  // 128-bit shifted mantissa W = (m53 << 75) >> off per element (src/split.cu:163-175), built as four 32-bit words
  // without branches: V = m53 << 11 (top-aligned 64 bits) is shifted right by off % 32 with funnel shifts and then moved
  // down by off / 32 whole words with selects.  (The first form used 64-bit variable shifts under per-element
  // `if (off < 64) ... else if (off < 128)`: exec-mask branches and quarter-rate shifts, a third of the kernel's VALU time.)
  constexpr int NW = LOW ? 4 : 2; // words kept: the top NW of the four; W[q][i] = word 4 - NW + i
  unsigned W[16][NW];
  unsigned negmask[4] = {0, 0, 0, 0}; // per packed word: 0xFF in the bytes of negative elements
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const unsigned long long bq = (unsigned long long)__double_as_longlong(v[q]);
    const unsigned bh = (unsigned)(bq >> 32), bl = (unsigned)bq;
    if constexpr (!LOW && FPBUILD) {
      // The upper 64 bits of the shifted mantissa by FP64 arithmetic (round 6): they are floor(m53 * 2^(11 - off)) =
      // floor(|x| * 2^(1085 - e)) - |x| < 2^(e - 1022), so y = |x| * 2^(1053 - e) < 2^31 is an exact scaling (v_ldexp_f64; a result
      // below the normal range only loses bits far below 2^-64 of the value), its integer part the upper word, and the next 32 bits
      // the integer part of (y - upper) * 2^32 (the difference is exact).  Subnormal elements scale by their true value, which is the
      // exponent-field-1 rule above.  6 instructions instead of ~20 selects and funnel shifts per element: the cut is VALU bound.
      const double y = live ? __builtin_ldexp(__builtin_fabs(v[q]), 1053 - (int)e) : 0.0;
      const unsigned hi = (unsigned)y;
      W[q][1] = hi;
      W[q][0] = (unsigned)__builtin_ldexp(y - (double)hi, 32);
      if (bh >> 31) negmask[q >> 2] |= 0xFFu << (8 * (q & 3));
      continue;
    }
    const unsigned f = (bh >> 20) & 0x7FFu;
    const unsigned mh = (bh & 0xFFFFFu) | (f ? 0x100000u : 0u); // m53 = (mh:bl), 21 + 32 bits
    const unsigned ef = f ? f : 1u; // subnormal: exponent of field 1 (fix of SURVEY §8a quirk 7)
    const unsigned off = e + 1u - ef; // >= 1 for live rows
    const unsigned vh = live ? ((mh << 11) | (bl >> 21)) : 0u, vl = live ? (bl << 11) : 0u; // V = m53 << 11
    const unsigned wa = off >> 5, wb = off & 31u;
    const unsigned t3 = vh >> wb, t2 = __builtin_amdgcn_alignbit(vh, vl, wb), t1 = __builtin_amdgcn_alignbit(vl, 0u, wb);
    W[q][NW - 1] = wa == 0u ? t3 : 0u;
    W[q][NW - 2] = wa == 0u ? t2 : wa == 1u ? t3 : 0u;
    if constexpr (LOW) {
      W[q][1] = wa == 0u ? t1 : wa == 1u ? t2 : wa == 2u ? t3 : 0u;
      W[q][0] = wa == 1u ? t1 : wa == 2u ? t2 : wa == 3u ? t3 : 0u;
    }
    if (bh >> 31) negmask[q >> 2] |= 0xFFu << (8 * (q & 3)); // sign_flag = a > 0 (src/split.cu:159)
  }

  // Slice s = bits [p, p + L) of the 128-bit value, p = 128 - (s + 1) * L >= 2.  The kernel is VALU bound (rocprofv3:
  // 1 770 VALU instructions per block in the first version, mostly 64-bit shifts at a fraction of the 32-bit rate),
  // so the field extraction works on the four 32-bit words of (hi:lo): p is wave-uniform, the word index selects one of
  // four unrolled bodies with a scalar branch, and a field costs one v_bfe_u32 (inside a word) or v_alignbit_b32 +
  // v_and (across two words), plus one v_lshl_or_b32 to place its byte.
  const unsigned mask = (1u << L) - 1u;
  auto word = [&](int q, int i) -> unsigned { // word i (0 = least significant) of element q's 128-bit value
    return (i >= 4 || i < 4 - NW) ? 0u : W[q][(i >= 4 || i < 4 - NW) ? 0 : i - (4 - NW)];
  };
  for (int s = 0; s < S; s++) {
    const int p = 128 - (s + 1) * L;
    const int wi = p >> 5, r = p & 31;
    unsigned w[4] = {0, 0, 0, 0};
    auto extract = [&](auto wic) {
      constexpr int WI = decltype(wic)::value;
      if (r + L <= 32) {
#pragma unroll
        for (int q = 0; q < 16; q++) w[q >> 2] |= ((word(q, WI) >> r) & mask) << (8 * (q & 3));
      } else {
#pragma unroll
        for (int q = 0; q < 16; q++)
          w[q >> 2] |= (__builtin_amdgcn_alignbit(word(q, WI + 1), word(q, WI), (unsigned)r) & mask) << (8 * (q & 3));
      }
    };
    if constexpr (LOW) {
      switch (wi) { // wave-uniform
        case 3: extract(std::integral_constant<int, 3>{}); break;
        case 2: extract(std::integral_constant<int, 2>{}); break;
        case 1: extract(std::integral_constant<int, 1>{}); break;
        default: extract(std::integral_constant<int, 0>{}); break;
      }
    } else {
      if (wi == 3) extract(std::integral_constant<int, 3>{});
      else extract(std::integral_constant<int, 2>{}); // p >= 64 by the caller's guarantee
    }
    uint4 o;
    unsigned *op = &o.x;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      // per-byte two's complement of the bytes selected by negmask (values <= 127, SWAR, no carries)
      const unsigned m = negmask[i], tt = w[i] ^ m;
      op[i] = ((tt & 0x7F7F7F7Fu) + (m & 0x01010101u)) ^ (tt & 0x80808080u);
    }
    if constexpr (WT) {
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      const v4u ov = {o.x, o.y, o.z, o.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(out + (size_t)s * FRAG_BYTES), "v"(ov) : "memory");
    } else {
      *(uint4 *)(out + (size_t)s * FRAG_BYTES) = o;
    }
  }
}

// ---- resident split: ONE read of the operand, the strip held in registers (round 4) ------------------------------------
// The two-pass form reads every element twice (16 + S bytes per element) because the cut needs the row maximum of the WHOLE
// row; the one-pass form above re-reads its strip from L2 with four waves.  Here a strip of R rows x K lives in the registers
// of ONE workgroup of up to 8 waves: every wave loads up to 4 unit blocks of 1024 elements (R rows x 1024 / R k: 16 doubles =
// 32 registers per lane and block, the same lane = (row, 16 consecutive k) assignment the cut works on), the waves combine
// their row maxima through R words of LDS, and the cut runs on the registers.  HBM sees 8 + S bytes per element, the
// exponent-word buffer, its epoch tags and atomics are not involved, and a call needs one split launch instead of two.
//   R = 32: K <= 1024, unit block = one 32 x 32 fragment block;   R = 16: K <= 2048, 16 rows x 64 k;   R = 8: K <= 4096, 8 x 128.
// The host takes the tallest strip that still gives every CU a workgroup (small problems are bound by the cut's VALU work and
// by latency, not by bandwidth: 1024 x 1024 x 2 operands in 32-row strips would keep 64 of 256 CUs busy).
// Loads are coalesced in both layouts: row-contiguous operands directly (R * 8 bytes per k), k-contiguous ones with the
// lanes along k (runs of 256 bytes and more) and a transpose through the wave's LDS tile.
template <int R, bool KCONTIG>
__device__ __forceinline__ void fetch_unit(const double *__restrict__ in, size_t rows, size_t K, size_t sr, size_t sk,
                                           size_t row0, size_t k0, int lane, double t[16]) {
  constexpr int KU = 1024 / R; // k per unit block
  if constexpr (!KCONTIG) {
    const size_t rg = row0 + (size_t)(lane % R);
    const size_t kbase = k0 + (size_t)(lane / R) * 16;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const size_t k = kbase + q;
      t[q] = (rg < rows && k < K) ? in[k * sk + rg * sr] : 0.0;
    }
  } else {
#pragma unroll
    for (int it = 0; it < 16; it++) { // instruction `it` covers elements it * 64 .. + 63 of the unit in (row, k) order
      const int el = it * 64 + lane;
      const size_t rg = row0 + (size_t)(el / KU), k = k0 + (size_t)(el % KU);
      t[it] = (rg < rows && k < K) ? in[rg * sr + k * sk] : 0.0;
    }
  }
}
// k-contiguous: t[] (lanes along k) -> v[] (16 consecutive k of row lane % R) through the wave's tile [R][1024 / R + 1]
template <int R>
__device__ __forceinline__ void transpose_unit(const double t[16], int lane, double *tile, double v[16]) {
  constexpr int KU = 1024 / R, LD = KU + 1;
  __builtin_amdgcn_wave_barrier(); // the previous block's reads of the tile are done
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const int el = it * 64 + lane;
    tile[(el / KU) * LD + el % KU] = t[it];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int r = lane % R, g = lane / R;
#pragma unroll
  for (int q = 0; q < 16; q++) v[q] = tile[r * LD + g * 16 + q];
}

constexpr int RES_MAX_WAVES = 8, RES_UNITS = 4;
constexpr int RES_TILE_DOUBLES = 1056; // >= R * (1024 / R + 1) for R = 32, 16, 8

// UNITS: unit blocks a wave may hold (RES_UNITS; the one-launch kernel of slice_gemm_one_launch.hip stops at K = 2048: 2)
template <int R, bool KCONTIG, bool LOW, int UNITS = RES_UNITS, bool WT = false>
__device__ __forceinline__ void split_resident_strip(const SplitJob &j, int S, int L, size_t strip, double *tile,
                                                     unsigned *row_e) {
  constexpr int KU = 1024 / R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const size_t rows = j.v.rows, K = j.v.K, sr = j.v.stride_r, sk = j.v.stride_k;
  const size_t KB = k_blocks(K), NU = (KB * 32 + KU - 1) / KU; // unit blocks along k (the last one may be partial: KB * 32 % KU)
  const size_t row0 = strip * R;
  if (threadIdx.x < R) row_e[threadIdx.x] = 0u;
  __syncthreads();
  // unit block u of this wave: number wave + nwaves * u along k (neighbouring waves read neighbouring memory)
  double v[UNITS][16];
  unsigned e = 0;
  if (row0 < rows) {
#pragma unroll
    for (int u = 0; u < UNITS; u++) {
      const size_t ub = (size_t)wave + (size_t)nwaves * u;
      if (ub < NU) {
        if constexpr (KCONTIG) {
          double t[16];
          fetch_unit<R, true>(j.v.in, rows, K, sr, sk, row0, ub * KU, lane, t);
          transpose_unit<R>(t, lane, tile, v[u]);
        } else {
          fetch_unit<R, false>(j.v.in, rows, K, sr, sk, row0, ub * KU, lane, v[u]);
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const unsigned x = exp_field(v[u][q]);
          e = x > e ? x : e;
        }
      }
    }
    if (e) atomicMax(&row_e[lane % R], e); // LDS
  }
  __syncthreads();
  e = row_e[lane % R];
  const size_t rg = row0 + (size_t)(lane % R);
  if (wave == 0 && lane < R && rg < rows) {
    if constexpr (WT) {
      asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(j.max_exp + rg), "v"(max_exp_of(e)) : "memory");
    } else {
      j.max_exp[rg] = max_exp_of(e);
    }
  }
  // rows beyond `rows` (the planes are padded to TILE_ROWS) and k beyond K were read as +0.0: e = 0 / zero slices
  const size_t rb = row0 / 32;
  const int g = lane / R; // k-group of 16 inside the unit: k-block g >> 1 of the unit's KU / 32, k-half g & 1
#pragma unroll
  for (int u = 0; u < UNITS; u++) {
    const size_t ub = (size_t)wave + (size_t)nwaves * u;
    if (ub < NU) {
      if (row0 >= rows) {
#pragma unroll
        for (int q = 0; q < 16; q++) v[u][q] = 0.0;
      }
      const size_t kb = ub * (KU / 32) + (size_t)(g >> 1);
      if (kb >= KB) continue; // the last unit of an operand whose KB is no multiple of KU / 32 reaches beyond the planes
      int8_t *out = j.planes + ((rb * KB + kb) * (size_t)S) * FRAG_BYTES + (size_t)(g & 1) * 512 +
                    (size_t)((row0 & 31) + lane % R) * 16;
      cut_and_store<LOW, WT>(v[u], row0 < rows ? e : 0u, S, L, out);
    }
  }
}

} // namespace ozhip
