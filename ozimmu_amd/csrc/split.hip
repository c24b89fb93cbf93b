// split.hip — FP64 -> S x INT8 mantissa slices (+ per-row exponent), and the auto-mode statistic.
//
// Restates, for gfx950, the reference's split path:
//   get_exp_max_element   /root/reference/src/split.cu:13-67   (row max of the exponent field)
//   cut_int8_core         src/split.cu:154-185                 (truncating bit-slice of the mantissa)
//   split_int8_kernel     src/split.cu:193-242                 (layout, zero padding, max_exp store)
//   calculate_mantissa_loss_core/kernel  src/split.cu:317-380  (auto mode statistic)
//
// MI355X design (HBM-bound byte work, no MFMA):
//  * both memory layouts of an operand are read with the lanes of a half-wave on 32 consecutive
//    doubles (256 B runs): row-contiguous operands directly, k-contiguous operands through an LDS
//    transpose -- never the reference's stride-ld per-thread walk (src/split.cu:208-210);
//  * the row maximum is a separate streaming pass (u32 atomicMax per row, 64-lane DPP/LDS reduction
//    instead of the reference's hard-coded 32-lane shuffles, src/split.cu:35-57);
//  * one wave cuts one 32x32 block and writes S fragment blocks of 1 KiB, 16 B per lane, straight in
//    MFMA operand order (layout.h) -- S contiguous KiB per wave instead of S strided byte stores per
//    thread (src/split.cu:176-184);
//  * no device synchronisation (the reference calls cudaDeviceSynchronize per operand, :261).
//
// Algorithmic HBM bytes: (8 + S) per element (one read, S slice bytes).  Operands that stay cache resident are split
// in one pass (split_fused_kernel: 8 + S from HBM); larger ones in two streaming passes (row-max pass + cut pass:
// 16 + S).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>

#include <type_traits>

#include "config.h"
#include "kernels.h"
#include "layout.h"
#include "split_resident.h"

namespace ozhip {



__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// ---- pass 1: per-row maximum exponent field ---------------------------------------------------------
// k-contiguous operand: element (r,k) at in[r*ld + k].  One wave per row segment, lanes along k.
__device__ __forceinline__ void row_max_kcontig(const double *__restrict__ in, size_t rows, size_t K, size_t sr,
                                                size_t sk, uint32_t *exps, unsigned kchunk, size_t bx, size_t by,
                                                uint32_t tag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t r = bx * 4 + wave;
  if (r >= rows) return;
  const size_t k0 = by * kchunk;
  const size_t k1 = k0 + kchunk < K ? k0 + kchunk : K;
  const double *p = in + r * sr;
  unsigned e = 0;
  size_t k = k0 + lane;
  for (; k + 7 * 64 < k1; k += 8 * 64) {
    unsigned t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) t[u] = exp_field(p[(k + u * 64) * sk]);
#pragma unroll
    for (int u = 0; u < 8; u++) e = t[u] > e ? t[u] : e;
  }
  for (; k < k1; k += 64) {
    const unsigned t = exp_field(p[k * sk]);
    e = t > e ? t : e;
  }
  e = wave_max_u32(e);
  if (lane == 0 && e) atomicMax(exps + r, tag | e);
}

// row-contiguous operand: element (r,k) at in[k*ld + r].  Lanes along r, the 4 waves interleave k.
__device__ __forceinline__ void row_max_rcontig(const double *__restrict__ in, size_t rows, size_t K, size_t sr,
                                                size_t sk, uint32_t *exps, unsigned kchunk, size_t bx, size_t by,
                                                unsigned (*red)[64], uint32_t tag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t r = bx * 64 + lane;
  const size_t k0 = by * kchunk;
  const size_t k1 = k0 + kchunk < K ? k0 + kchunk : K;
  unsigned e = 0;
  if (r < rows) {
    const double *p = in + r * sr;
    size_t k = k0 + wave;
    for (; k + 7 * 4 < k1; k += 8 * 4) {
      unsigned t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = exp_field(p[(k + u * 4) * sk]);
#pragma unroll
      for (int u = 0; u < 8; u++) e = t[u] > e ? t[u] : e;
    }
    for (; k < k1; k += 4) {
      const unsigned t = exp_field(p[k * sk]);
      e = t > e ? t : e;
    }
  }
  red[wave][lane] = e;
  __syncthreads();
  if (wave == 0) {
    e = red[0][lane];
#pragma unroll
    for (int w = 1; w < 4; w++) e = red[w][lane] > e ? red[w][lane] : e;
    if (r < rows && e) atomicMax(exps + r, tag | e);
  }
}

// Up to 4 operand views per launch (A and B of a product, or Re/Im of both): blockIdx.x enumerates the (row group,
// k chunk) pairs of all views; blockIdx.z = matrix of a batch.  Small problems are bounded by launch gaps, not
// bandwidth: one launch per pass instead of one per operand view.
__global__ __launch_bounds__(256) void row_max_kernel(const SplitJobs jobs) {
  __shared__ unsigned red[4][64];
  int ji = 0;
  uint32_t blk = blockIdx.x;
#pragma unroll
  for (int q = 0; q < 3; q++)
    if (ji + 1 < jobs.count && blk >= jobs.nblk[ji]) {
      blk -= jobs.nblk[ji];
      ji++;
    }
  if (jobs.zero_words && blockIdx.x == 0 && blockIdx.z == 0)
    for (uint32_t i = threadIdx.x; i < jobs.zero_words; i += 256u) jobs.zero_ptr[i] = 0u;
  SplitJob j = jobs.job[0];
  uint32_t nx = jobs.nx[0], kchunk = jobs.kchunk[0];
  if (ji == 1) j = jobs.job[1], nx = jobs.nx[1], kchunk = jobs.kchunk[1];
  if (ji == 2) j = jobs.job[2], nx = jobs.nx[2], kchunk = jobs.kchunk[2];
  if (ji == 3) j = jobs.job[3], nx = jobs.nx[3], kchunk = jobs.kchunk[3];
  const double *in = j.v.in + (long long)blockIdx.z * j.in_stride;
  uint32_t *exps = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(j.exps) + (size_t)blockIdx.z * jobs.exps_stride);
  const uint32_t bx = blk % nx, by = blk / nx;
  if (j.v.stride_k < j.v.stride_r)
    row_max_kcontig(in, j.v.rows, j.v.K, j.v.stride_r, j.v.stride_k, exps, kchunk, bx, by, jobs.tag);
  else
    row_max_rcontig(in, j.v.rows, j.v.K, j.v.stride_r, j.v.stride_k, exps, kchunk, bx, by, red, jobs.tag);
}

hipError_t launch_row_max_multi(const SplitJob *job, int count, hipStream_t stream, uint32_t batch, size_t ws_stride,
                                size_t exps_stride, uint32_t tag, uint32_t *zero_ptr, uint32_t zero_words) {
  if (count < 1 || count > 4) return hipErrorInvalidValue;
  SplitJobs jobs{};
  jobs.zero_ptr = zero_ptr;
  jobs.zero_words = zero_ptr ? zero_words : 0;
  jobs.count = count;
  jobs.ws_stride = ws_stride;
  jobs.exps_stride = exps_stride;
  jobs.tag = tag;
  uint64_t total = 0;
  for (int i = 0; i < count; i++) {
    const OperandView &v = job[i].v;
    jobs.job[i] = job[i];
    const bool kc = v.stride_k < v.stride_r;
    jobs.nx[i] = (uint32_t)(kc ? (v.rows + 3) / 4 : (v.rows + 63) / 64);
    // k values per workgroup: 8192 (one wave per row) / 512 (64 rows per workgroup) for operands that fill the chip;
    // small operands get shorter chunks until ~1024 workgroups exist (1024 x 1024 row-contiguous: 32 -> 256 workgroups)
    unsigned kchunk = kc ? 8192u : 512u;
    const unsigned min_chunk = kc ? 1024u : 64u;
    while (kchunk > min_chunk && (uint64_t)jobs.nx[i] * ((v.K + kchunk - 1) / kchunk) * batch < 1024u) kchunk /= 2;
    jobs.kchunk[i] = kchunk;
    jobs.nblk[i] = (v.rows && v.K) ? jobs.nx[i] * (uint32_t)((v.K + kchunk - 1) / kchunk) : 0;
    if (jobs.nx[i] == 0) jobs.nx[i] = 1;
    total += jobs.nblk[i];
  }
  if (total == 0 || batch == 0) // nothing to scan: the side job still has to happen
    return jobs.zero_words ? launch_zero_words(zero_ptr, (size_t)zero_words * 4, 0, 1, stream) : hipSuccess;
  if (total > 0x7FFFFFFFull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(row_max_kernel, dim3((unsigned)total, 1, batch), dim3(256), 0, stream, jobs);
  return hipGetLastError();
}

// Zero `bytes` (a multiple of 4) at the head of `count` workspace slots `pitch` bytes apart.  A plain kernel instead
// of hipMemsetAsync / hipMemset2DAsync: the runtime's fill costs ~4 us plus a ~6 us dispatch gap per call (rocprofv3
// kernel trace at 1024^3, where the whole call is ~70 us); this one queues back-to-back with the split kernels.
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t *base, uint32_t words, size_t pitch) {
  uint32_t *p = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(base) + (size_t)blockIdx.y * pitch);
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < words) p[i] = 0u;
}

hipError_t launch_zero_words(void *base, size_t bytes, size_t pitch, uint32_t count, hipStream_t stream) {
  if (bytes == 0 || count == 0) return hipSuccess;
  if ((bytes & 3u) != 0 || (reinterpret_cast<uintptr_t>(base) & 3u) != 0 || bytes > ((size_t)1 << 33) || count > 65535u)
    return hipErrorInvalidValue;
  const uint32_t words = (uint32_t)(bytes / 4);
  hipLaunchKernelGGL(zero_words_kernel, dim3((words + 255u) / 256u, count), dim3(256), 0, stream,
                     reinterpret_cast<uint32_t *>(base), words, pitch);
  return hipGetLastError();
}

hipError_t launch_row_max_exp(const OperandView &v, uint32_t *exps, hipStream_t stream, const Batch &b, uint32_t *zero_ptr,
                              uint32_t zero_words) {
  const SplitJob job{v, nullptr, nullptr, b.in_stride, exps};
  return launch_row_max_multi(&job, 1, stream, b.count, b.ws_stride, b.exps_stride, b.tag, zero_ptr, zero_words);
}

// ---- shared: load one 32 rows x 32 k block, lane (r = lane&31, kh = lane>>5) gets its 16 k values ---
// Out-of-range rows / k read as +0.0 (-> zero slices: the padding the GEMM relies on).
// Two steps so that a wave can have the NEXT block's loads in flight while it cuts the current one:
//   fetch_block      global -> registers, coalesced in both layouts (half-wave = 32 consecutive doubles)
//   arrange_block    row-contiguous: already in place; k-contiguous: transpose through the wave's LDS tile
template <bool KCONTIG>
__device__ __forceinline__ void fetch_block(const double *__restrict__ in, size_t rows, size_t K, size_t sr, size_t sk,
                                            size_t rb, size_t kb, int lane, double t[16]) {
  if constexpr (!KCONTIG) {
    const size_t rg = rb * 32 + (lane & 31);
    const size_t kbase = kb * 32 + (lane >> 5) * 16;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const size_t k = kbase + q;
      t[q] = (rg < rows && k < K) ? in[k * sk + rg * sr] : 0.0;
    }
  } else {
    const size_t k = kb * 32 + (lane & 31);
#pragma unroll
    for (int it = 0; it < 16; it++) { // 2 rows per instruction
      const size_t rg = rb * 32 + it * 2 + (lane >> 5);
      t[it] = (rg < rows && k < K) ? in[rg * sr + k * sk] : 0.0;
    }
  }
}

template <bool KCONTIG>
__device__ __forceinline__ void arrange_block(const double t[16], int lane, double (*tile)[33], double v[16]) {
  if constexpr (!KCONTIG) {
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = t[q];
  } else {
    __builtin_amdgcn_wave_barrier(); // the previous block's reads of the tile are done
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int it = 0; it < 16; it++) tile[it * 2 + (lane >> 5)][lane & 31] = t[it];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int r = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = tile[r][kh * 16 + q];
  }
}

template <bool KCONTIG>
__device__ __forceinline__ void load_block(const double *__restrict__ in, size_t rows, size_t K, size_t sr, size_t sk,
                                           size_t rb, size_t kb, int lane, double (*tile)[33], double v[16]) {
  double t[16];
  fetch_block<KCONTIG>(in, rows, K, sr, sk, rb, kb, lane, t);
  arrange_block<KCONTIG>(t, lane, tile, v);
}


// ---- pass 2: cut ---------------------------------------------------------------------------------------
// One wave cuts `strip` consecutive blocks along the memory-contiguous axis of the operand.  k-contiguous operands
// (PREFETCH): the loads of the next block are issued before the current one is cut -- with the LDS transpose the
// kernel sits at 2 waves/SIMD, too few to hide HBM latency by occupancy alone (8192^2: 0.53 -> 0.31 ms).
// Row-contiguous operands keep one block per wave at 3 waves/SIMD (prefetching there only costs registers).
template <bool KCONTIG, bool PREFETCH = KCONTIG, bool LOW = true>
__device__ __forceinline__ void cut_body(const double *__restrict__ in, size_t rows, size_t K, size_t sr, size_t sk,
                                         const uint32_t *__restrict__ exps, int S, int L, int8_t *__restrict__ planes,
                                         double *__restrict__ max_exp, size_t RB, size_t KB, int strip, size_t block,
                                         double (*tiles)[32][33], uint32_t tag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int CUT_STRIP = PREFETCH ? strip : 1;
  // strips along the contiguous axis: k-contiguous -> CUT_STRIP k-blocks of one row-block, else CUT_STRIP row-blocks
  const size_t strips_fast = ((KCONTIG ? KB : RB) + CUT_STRIP - 1) / CUT_STRIP;
  const size_t gw = block * 4 + wave;
  if (gw >= strips_fast * (KCONTIG ? RB : KB)) return;
  const size_t slow = gw / strips_fast, fast0 = (gw % strips_fast) * CUT_STRIP;
  const size_t nfast = KCONTIG ? KB : RB;
  auto block_of = [&](int b, size_t &rb, size_t &kb) {
    rb = KCONTIG ? slow : fast0 + b;
    kb = KCONTIG ? fast0 + b : slow;
  };

  double t[16];
  if constexpr (PREFETCH) {
    size_t rb, kb;
    block_of(0, rb, kb);
    fetch_block<KCONTIG>(in, rows, K, sr, sk, rb, kb, lane, t);
  }
#pragma unroll 1
  for (int b = 0; b < CUT_STRIP && fast0 + b < nfast; b++) {
    size_t rb, kb;
    block_of(b, rb, kb);
    double v[16];
    if constexpr (PREFETCH) {
      arrange_block<KCONTIG>(t, lane, tiles[KCONTIG ? wave : 0], v);
      if (b + 1 < CUT_STRIP && fast0 + b + 1 < nfast) { // next block's loads fly while this one is cut
        size_t rbn, kbn;
        block_of(b + 1, rbn, kbn);
        fetch_block<KCONTIG>(in, rows, K, sr, sk, rbn, kbn, lane, t);
      }
    } else {
      load_block<KCONTIG>(in, rows, K, sr, sk, rb, kb, lane, tiles[KCONTIG ? wave : 0], v);
    }

    const int r = lane & 31;
    const size_t rg = rb * 32 + r;
    const unsigned e = rg < rows ? tagged_exp(exps[rg], tag) : 0u;
    if (kb == 0 && lane < 32 && rg < rows) max_exp[rg] = max_exp_of(e);
    cut_and_store<LOW>(v, e, S, L, planes + ((rb * KB + kb) * (size_t)S) * FRAG_BYTES + (size_t)lane * 16);
  }
}

// one operand view per launch: each layout gets its own register budget (k-contiguous: 2 waves/SIMD with the LDS
// transpose and prefetch; row-contiguous: 3 waves/SIMD) -- the form for operands that are bandwidth bound
template <bool KCONTIG, bool PREFETCH = KCONTIG, bool LOW = true>
__global__ __launch_bounds__(256) void cut_kernel(const double *__restrict__ in, size_t rows, size_t K,
                                                  size_t sr, size_t sk, const uint32_t *__restrict__ exps, int S, int L,
                                                  int8_t *__restrict__ planes, double *__restrict__ max_exp,
                                                  size_t RB, size_t KB, int strip, long long in_stride,
                                                  size_t ws_stride, size_t exps_stride, uint32_t tag, int reverse) {
  __shared__ double tiles[KCONTIG ? 4 : 1][32][33];
  in += (long long)blockIdx.z * in_stride; // batch (kernels.h: Batch)
  exps = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(exps) + (size_t)blockIdx.z * exps_stride);
  planes += (size_t)blockIdx.z * ws_stride;
  max_exp = reinterpret_cast<double *>(reinterpret_cast<char *>(max_exp) + (size_t)blockIdx.z * ws_stride);
  // reverse: walk the operand from its END.  Both passes walk memory in ascending order by workgroup index; the row-maximum
  // pass has just streamed the operand through the 256 MiB memory-side cache, so what it read LAST is what is still there.
  const size_t block = reverse ? (size_t)(gridDim.x - 1u - blockIdx.x) : (size_t)blockIdx.x;
  cut_body<KCONTIG, PREFETCH, LOW>(in, rows, K, sr, sk, exps, S, L, planes, max_exp, RB, KB, strip, block, tiles, tag);
}

// up to 4 operand views per launch (see row_max_kernel): the form for small problems
// PF: the k-contiguous path prefetches along its strip (only useful for strips of more than one block; without it the
// kernel needs ~100 instead of 186 registers, i.e. 4 instead of 2 waves per SIMD for BOTH layouts of the launch)
template <bool PF, bool LOW = true>
__global__ __launch_bounds__(256) void cut_multi_kernel(const SplitJobs jobs) {
  // the transpose tiles of the k-contiguous path: dynamic, so that a launch with row-contiguous views only (N/T products:
  // panel updates) allocates none and is not held to four workgroups per CU by 33 KiB of LDS it never touches
  extern __shared__ __attribute__((aligned(16))) double tiles_dyn[];
  double(*tiles)[32][33] = reinterpret_cast<double(*)[32][33]>(tiles_dyn);
  int ji = 0;
  uint32_t blk = blockIdx.x;
#pragma unroll
  for (int q = 0; q < 3; q++)
    if (ji + 1 < jobs.count && blk >= jobs.nblk[ji]) {
      blk -= jobs.nblk[ji];
      ji++;
    }
  SplitJob j = jobs.job[0];
  uint32_t strip = jobs.nx[0];
  if (ji == 1) j = jobs.job[1], strip = jobs.nx[1];
  if (ji == 2) j = jobs.job[2], strip = jobs.nx[2];
  if (ji == 3) j = jobs.job[3], strip = jobs.nx[3];
  const size_t off = (size_t)blockIdx.z * jobs.ws_stride;
  const double *in = j.v.in + (long long)blockIdx.z * j.in_stride;
  const uint32_t *exps = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(j.exps) + (size_t)blockIdx.z * jobs.exps_stride);
  double *max_exp = reinterpret_cast<double *>(reinterpret_cast<char *>(j.max_exp) + off);
  const size_t RB = (j.v.rows + TILE_ROWS - 1) / TILE_ROWS * (TILE_ROWS / FRAG_ROWS), KB = k_blocks(j.v.K);
  if (j.v.stride_k < j.v.stride_r)
    cut_body<true, PF, LOW>(in, j.v.rows, j.v.K, j.v.stride_r, j.v.stride_k, exps, jobs.S, jobs.L, j.planes + off, max_exp, RB,
                       KB, (int)strip, blk, tiles, jobs.tag);
  else
    cut_body<false, false, LOW>(in, j.v.rows, j.v.K, j.v.stride_r, j.v.stride_k, exps, jobs.S, jobs.L, j.planes + off, max_exp,
                           RB, KB, 1, blk, tiles, jobs.tag);
}

static int cut_strip_for(bool kcontig, size_t RB, size_t KB) {
  // One block per wave in both layouts.  Strips of 2..4 blocks with a register prefetch along the strip (PREFETCH) were
  // the faster form of the k-contiguous cut while it sat at 2 waves/SIMD; since the 32-bit field extraction the
  // one-block form needs 108 registers (4 waves/SIMD) and wins at every size (tools/ab_split_strip.py: 4096^3 -1.6 %,
  // 8192^3 -0.4..1 % of the call).  OZIMMU_HIP_SPLIT_STRIP=n keeps the strip form for A/B runs.
  (void)RB, (void)KB;
  if (const int e = config().split_strip) return kcontig ? std::max(1, e) : 1;
  return 1;
}

hipError_t launch_cut_multi(const SplitJob *job, int count, int S, int L, hipStream_t stream, uint32_t batch,
                            size_t ws_stride, size_t exps_stride, uint32_t tag) {
  if (count < 1 || count > 4) return hipErrorInvalidValue;
  SplitJobs jobs{};
  jobs.count = count;
  jobs.S = S;
  jobs.L = L;
  jobs.ws_stride = ws_stride;
  jobs.exps_stride = exps_stride;
  jobs.tag = tag;
  uint64_t total = 0;
  for (int i = 0; i < count; i++) {
    const OperandView &v = job[i].v;
    jobs.job[i] = job[i];
    const size_t RB = row_blocks_padded(v.rows), KB = k_blocks(v.K);
    const bool kc = v.stride_k < v.stride_r;
    const int strip = cut_strip_for(kc, RB, KB);
    const size_t strips = ((kc ? KB : RB) + strip - 1) / strip * (kc ? RB : KB);
    jobs.nx[i] = (uint32_t)strip;
    jobs.nblk[i] = (uint32_t)((strips + 3) / 4);
    total += jobs.nblk[i];
  }
  if (total == 0 || batch == 0) return hipSuccess;
  if (total > 0x7FFFFFFFull) return hipErrorInvalidValue;
  bool prefetch = false, any_kcontig = false;
  for (int i = 0; i < count; i++) {
    prefetch = prefetch || (jobs.nblk[i] && jobs.nx[i] > 1);
    any_kcontig = any_kcontig || (jobs.nblk[i] && job[i].v.stride_k < job[i].v.stride_r);
  }
  const size_t lds = any_kcontig ? sizeof(double) * 4 * 32 * 33 : 0;
  if (prefetch)
    hipLaunchKernelGGL(cut_multi_kernel<true>, dim3((unsigned)total, 1, batch), dim3(256), lds, stream, jobs);
  else if (S * L > 64)
    hipLaunchKernelGGL((cut_multi_kernel<false, true>), dim3((unsigned)total, 1, batch), dim3(256), lds, stream, jobs);
  else // every slice in the upper 64 bits of the shifted mantissa (up to fp64_int8_9 at L = 7): the lean form
    hipLaunchKernelGGL((cut_multi_kernel<false, false>), dim3((unsigned)total, 1, batch), dim3(256), lds, stream, jobs);
  return hipGetLastError();
}

// OZIMMU_HIP_SPLIT_REVERSE (development switch, read like those of config.h): 0 = the cut pass walks the operand from its start
static bool cut_reverse() {
  auto read = []() {
    const char *e = counted_getenv("OZIMMU_HIP_SPLIT_REVERSE");
    return !(e && e[0] == '0');
  };
  if (config().env_per_call) return read();
  static const bool r = read();
  return r;
}

hipError_t launch_cut(const OperandView &v, const uint32_t *exps, int S, int L, int8_t *planes,
                      double *max_exp, hipStream_t stream, const Batch &b) {
  const size_t RB = row_blocks_padded(v.rows), KB = k_blocks(v.K);
  if (RB * KB == 0 || b.count == 0) return hipSuccess;
  const bool kcontig = v.stride_k < v.stride_r;
  const int strip = cut_strip_for(kcontig, RB, KB);
  const size_t strips = ((kcontig ? KB : RB) + strip - 1) / strip * (kcontig ? RB : KB);
  const dim3 grid((unsigned)((strips + 3) / 4), 1, b.count);
  const bool low = S * L > 64; // some slice reaches into the lower 64 bits of the shifted mantissa
  const int rev = cut_reverse() ? 1 : 0;
  if (kcontig && strip == 1 && low) // no strip to prefetch along: the register-lean form (4 waves per SIMD instead of 2)
    hipLaunchKernelGGL((cut_kernel<true, false, true>), grid, dim3(256), 0, stream, v.in, v.rows, v.K, v.stride_r, v.stride_k,
                       exps, S, L, planes, max_exp, RB, KB, strip, b.in_stride, b.ws_stride, b.exps_stride, b.tag, rev);
  else if (kcontig && strip == 1)
    hipLaunchKernelGGL((cut_kernel<true, false, false>), grid, dim3(256), 0, stream, v.in, v.rows, v.K, v.stride_r, v.stride_k,
                       exps, S, L, planes, max_exp, RB, KB, strip, b.in_stride, b.ws_stride, b.exps_stride, b.tag, rev);
  else if (kcontig)
    hipLaunchKernelGGL(cut_kernel<true>, grid, dim3(256), 0, stream, v.in, v.rows, v.K, v.stride_r, v.stride_k,
                       exps, S, L, planes, max_exp, RB, KB, strip, b.in_stride, b.ws_stride, b.exps_stride, b.tag, rev);
  else if (low)
    hipLaunchKernelGGL((cut_kernel<false, false, true>), grid, dim3(256), 0, stream, v.in, v.rows, v.K, v.stride_r, v.stride_k,
                       exps, S, L, planes, max_exp, RB, KB, strip, b.in_stride, b.ws_stride, b.exps_stride, b.tag, rev);
  else
    hipLaunchKernelGGL((cut_kernel<false, false, false>), grid, dim3(256), 0, stream, v.in, v.rows, v.K, v.stride_r, v.stride_k,
                       exps, S, L, planes, max_exp, RB, KB, strip, b.in_stride, b.ws_stride, b.exps_stride, b.tag, rev);
  return hipGetLastError();
}

// ---- one-pass split for operands that stay cache resident ------------------------------------------------------------
// One workgroup owns a 32-row block for ALL of K: phase 1 finds the 32 row maxima (kept in LDS), phase 2 re-reads the
// strip -- 32 x K x 8 bytes, L2 / Infinity-Cache hits as long as the operand is small against the 256 MiB cache -- and
// cuts it.  HBM sees the operand once (algorithmic 8 + S bytes per element instead of 16 + S), and the exponent
// words, their memset and the atomics of the two-pass form disappear: up to four operand views (A and B of a real
// product, Re/Im of both for a complex one) x the matrices of a batch are split by ONE launch, which is what small
// problems -- bounded by launch gaps, not bandwidth -- need.
template <bool KCONTIG>
__device__ __forceinline__ void split_strip(const SplitJob &j, int S, int L, size_t rb, double (*tile)[33],
                                            unsigned (*red)[32]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t KB = k_blocks(j.v.K);
  const double *in = j.v.in;
  const size_t rows = j.v.rows, K = j.v.K, sr = j.v.stride_r, sk = j.v.stride_k;
  // phase 1: exponent maxima.  Row-contiguous: a lane sees 16 k of its own row (lane & 31); k-contiguous: element `it`
  // of a lane belongs to row 2*it + (lane >> 5), reduced across lanes once at the end.
  unsigned e_rc = 0, e_kc[16];
#pragma unroll
  for (int q = 0; q < 16; q++) e_kc[q] = 0;
  if (rb * 32 < rows) {
    // a strip is walked by only 4 waves: keep DEPTH blocks of loads in flight per wave (the pass is latency bound)
    constexpr int DEPTH = 4;
    for (size_t kb0 = wave; kb0 < KB; kb0 += 4 * DEPTH) {
      double t[DEPTH][16];
#pragma unroll
      for (int d = 0; d < DEPTH; d++)
        if (kb0 + 4 * d < KB) fetch_block<KCONTIG>(in, rows, K, sr, sk, rb, kb0 + 4 * d, lane, t[d]);
#pragma unroll
      for (int d = 0; d < DEPTH; d++)
        if (kb0 + 4 * d < KB) {
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const unsigned x = exp_field(t[d][q]);
            if constexpr (KCONTIG)
              e_kc[q] = x > e_kc[q] ? x : e_kc[q];
            else
              e_rc = x > e_rc ? x : e_rc;
          }
        }
    }
  }
  if constexpr (KCONTIG) {
#pragma unroll
    for (int q = 0; q < 16; q++) {
      unsigned x = e_kc[q];
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) { // within the half-wave that shares (lane >> 5)
        const unsigned o = __shfl_xor(x, off, 64);
        x = o > x ? o : x;
      }
      if ((lane & 31) == 0) red[wave][2 * q + (lane >> 5)] = x;
    }
  } else {
    const unsigned o = __shfl_xor(e_rc, 32, 64);
    e_rc = o > e_rc ? o : e_rc;
    if (lane < 32) red[wave][lane] = e_rc;
  }
  __syncthreads();
  const int r = lane & 31;
  unsigned e = red[0][r];
#pragma unroll
  for (int w = 1; w < 4; w++) e = red[w][r] > e ? red[w][r] : e;
  const size_t rg = rb * 32 + r;
  if (wave == 0 && lane < 32 && rg < rows) j.max_exp[rg] = max_exp_of(e);
  // phase 2: cut (rows >= `rows` of the padded planes and k >= K read as zeros -> zero slices); the next block's
  // loads are in flight while the current one is cut
  double t[16];
  if ((size_t)wave < KB) fetch_block<KCONTIG>(in, rows, K, sr, sk, rb, wave, lane, t);
#pragma unroll 1
  for (size_t kb = wave; kb < KB; kb += 4) {
    double v[16];
    arrange_block<KCONTIG>(t, lane, tile, v);
    if (kb + 4 < KB) fetch_block<KCONTIG>(in, rows, K, sr, sk, rb, kb + 4, lane, t);
    cut_and_store(v, e, S, L, j.planes + ((rb * KB + kb) * (size_t)S) * FRAG_BYTES + (size_t)lane * 16);
  }
}

__global__ __launch_bounds__(256) void split_fused_kernel(const SplitJobs jobs) {
  __shared__ double tiles[4][32][33];
  __shared__ unsigned red[4][32];
  // blockIdx.x: row-block over all views (each padded to TILE_ROWS), blockIdx.z: matrix of the batch
  int ji = 0;
  uint32_t rb = blockIdx.x;
#pragma unroll
  for (int q = 0; q < 3; q++)
    if (ji + 1 < jobs.count && rb >= jobs.nblk[ji]) {
      rb -= jobs.nblk[ji];
      ji++;
    }
  SplitJob j = jobs.job[0];
  if (ji == 1) j = jobs.job[1];
  if (ji == 2) j = jobs.job[2];
  if (ji == 3) j = jobs.job[3];
  j.v.in += (long long)blockIdx.z * j.in_stride;
  j.planes += (size_t)blockIdx.z * jobs.ws_stride;
  j.max_exp = reinterpret_cast<double *>(reinterpret_cast<char *>(j.max_exp) + (size_t)blockIdx.z * jobs.ws_stride);
  const int wave = threadIdx.x >> 6;
  if (j.v.stride_k < j.v.stride_r)
    split_strip<true>(j, jobs.S, jobs.L, rb, tiles[wave], red);
  else
    split_strip<false>(j, jobs.S, jobs.L, rb, tiles[wave], red);
}

hipError_t launch_split_fused(const SplitJob *job, int count, int S, int L, hipStream_t stream, uint32_t batch,
                              size_t ws_stride) {
  if (count < 1 || count > 4) return hipErrorInvalidValue;
  SplitJobs jobs{};
  jobs.count = count;
  jobs.S = S;
  jobs.L = L;
  jobs.ws_stride = ws_stride;
  uint32_t total = 0;
  for (int i = 0; i < count; i++) {
    jobs.job[i] = job[i];
    jobs.nblk[i] = (uint32_t)row_blocks_padded(job[i].v.rows);
    total += jobs.nblk[i];
  }
  if (total == 0 || batch == 0) return hipSuccess;
  hipLaunchKernelGGL(split_fused_kernel, dim3(total, 1, batch), dim3(256), 0, stream, jobs);
  return hipGetLastError();
}


// UNITS: unit blocks a wave holds at most (1, 2 or RES_UNITS = 4).  The strip of a short K (K <= 256 / 512 at 32 rows) is one or two
// blocks per wave, and the kernel compiled for four keeps 167 registers: ONE workgroup of 8 waves per CU, whose load, row maximum, cut
// (VALU bound: ~860 instructions per block) and stores run one after the other with nothing else on the CU.  Compiled for the blocks it
// really holds it needs half the registers, two or three workgroups share a CU and one's loads run under another's cut
// (8192 x 256 x 2 operands: profiles/r6_ablate/README.md 7).
template <int R, bool LOW, int UNITS>
__global__ __launch_bounds__(64 * RES_MAX_WAVES) void split_resident_kernel(const SplitJobs jobs) {
  extern __shared__ __attribute__((aligned(16))) double res_tiles[]; // [waves][RES_TILE_DOUBLES]: k-contiguous views only
  __shared__ unsigned row_e[32];
  int ji = 0;
  uint32_t strip = blockIdx.x;
#pragma unroll
  for (int q = 0; q < 3; q++)
    if (ji + 1 < jobs.count && strip >= jobs.nblk[ji]) {
      strip -= jobs.nblk[ji];
      ji++;
    }
  if (jobs.zero_words && blockIdx.x == 0 && blockIdx.z == 0)
    for (uint32_t i = threadIdx.x; i < jobs.zero_words; i += blockDim.x) jobs.zero_ptr[i] = 0u;
  SplitJob j = jobs.job[0];
  if (ji == 1) j = jobs.job[1];
  if (ji == 2) j = jobs.job[2];
  if (ji == 3) j = jobs.job[3];
  j.v.in += (long long)blockIdx.z * j.in_stride;
  j.planes += (size_t)blockIdx.z * jobs.ws_stride;
  j.max_exp = reinterpret_cast<double *>(reinterpret_cast<char *>(j.max_exp) + (size_t)blockIdx.z * jobs.ws_stride);
  double *tile = res_tiles + (size_t)(threadIdx.x >> 6) * RES_TILE_DOUBLES;
  if (j.v.stride_k < j.v.stride_r) {
    split_resident_strip<R, true, LOW, UNITS>(j, jobs.S, jobs.L, strip, tile, row_e);
  } else {
    if constexpr (R == 8 && UNITS >= 2) {
      // Strips of 8 rows of a ROW-contiguous operand share every 128-byte line (16 rows of one k) with a neighbour, and workgroups go
      // to the XCDs round robin: neighbours on different XCDs fetch the operand twice, once into each L2.  Inside every aligned group
      // of 16 of an operand's workgroups the pair (2x, 2x + 1) therefore goes to the two workgroups 8 apart - the same XCD on an
      // 8-XCD part, a harmless permutation anywhere else.  Measured (profiles/r6_ablate/r6y_*): 1024^2 x 2048 +2...3 % (N/N), +5.8 % (N/T);
      // at K <= 1024 (one unit block per wave: a latency chain, not bandwidth) the two workgroups queue on the same L2 channel
      // instead and N/N loses 1 %: not applied there; k-contiguous operands share nothing and are left alone.
      const uint32_t nb = ji == 0 ? jobs.nblk[0] : ji == 1 ? jobs.nblk[1] : ji == 2 ? jobs.nblk[2] : jobs.nblk[3];
      if ((strip | 15u) < nb) strip = (strip & ~15u) + ((strip & 7u) << 1) + ((strip >> 3) & 1u);
    }
    split_resident_strip<R, false, LOW, UNITS>(j, jobs.S, jobs.L, strip, tile, row_e);
  }
}

// OZIMMU_HIP_SPLIT_RESIDENT_UNITS=1 / 2 / 4: the smallest instantiation taken (4 = always the general one, the round-5 form: A/B, tests);
// read once per process like the switches of csrc/config.h, per call under OZIMMU_HIP_ENV_PER_CALL
static int resident_units_switch() {
  static std::atomic<int> once{-2};
  int v = once.load(std::memory_order_relaxed);
  if (config().env_per_call || v == -2) {
    const char *e = std::getenv("OZIMMU_HIP_SPLIT_RESIDENT_UNITS");
    v = e ? std::atoi(e) : -1;
    once.store(v, std::memory_order_relaxed);
  }
  return v;
}

// the longest K a resident strip (8 rows) can hold: 8 waves x 4 unit blocks x 128 k
size_t resident_split_max_k() { return (size_t)RES_MAX_WAVES * RES_UNITS * 128; }

hipError_t launch_split_resident(const SplitJob *job, int count, int S, int L, hipStream_t stream, uint32_t batch,
                                 size_t ws_stride, uint32_t *zero_ptr, uint32_t zero_words, int cus) {
  if (count < 1 || count > 4) return hipErrorInvalidValue;
  SplitJobs jobs{};
  jobs.count = count;
  jobs.S = S;
  jobs.L = L;
  jobs.ws_stride = ws_stride;
  jobs.zero_ptr = zero_ptr;
  jobs.zero_words = zero_ptr ? zero_words : 0;
  size_t kmax = 0, blocks32 = 0;
  bool any_kcontig = false;
  for (int i = 0; i < count; i++) {
    if (job[i].v.K > resident_split_max_k()) return hipErrorInvalidValue;
    kmax = std::max(kmax, k_blocks(job[i].v.K) * 32);
    blocks32 += row_blocks_padded(job[i].v.rows);
    any_kcontig = any_kcontig || job[i].v.stride_k < job[i].v.stride_r;
  }
  // strip height: the tallest of 32 / 16 / 8 rows whose K capacity (1024 / 2048 / 4096) covers the operands and that still
  // gives every CU a workgroup; OZIMMU_HIP_SPLIT_RESIDENT=8 / 16 / 32 forces it where K allows
  auto capacity = [](int r) { return (size_t)RES_MAX_WAVES * RES_UNITS * (size_t)(1024 / r); };
  const int forced = config().split_resident;
  int R = 32;
  if (forced == 8 || forced == 16 || forced == 32) {
    R = forced;
    while (R > 8 && kmax > capacity(R)) R /= 2;
  } else {
    // (tools/ab.py, profiles/r4_ablate/r4f_resident_split_strip_height.txt: 1024^3 and 1024 x 1024 x 2048 want 256 strips of 8
    // rows rather than 128 of 16: +12 / +9 % of the call; 1536^3 is better off with 192 strips of 16 than with 384 of 8)
    while (R > 8 && (kmax > capacity(R) || 8 * blocks32 * (size_t)(32 / R) * batch < 5 * (size_t)cus)) R /= 2;
  }
  const size_t ku = 1024 / R;
  const size_t nu = (kmax + ku - 1) / ku;                                  // unit blocks of the longest view
  // waves per workgroup: <= 4 unit blocks per wave, and no wave idling through the last round of blocks (12 blocks: 6 waves x 2)
  const size_t per_wave = (std::max<size_t>(1, nu) + RES_MAX_WAVES - 1) / RES_MAX_WAVES;
  const unsigned waves = (unsigned)((std::max<size_t>(1, nu) + per_wave - 1) / per_wave);
  uint64_t total = 0;
  for (int i = 0; i < count; i++) {
    jobs.job[i] = job[i];
    jobs.nblk[i] = (uint32_t)(row_blocks_padded(job[i].v.rows) * (32 / R));
    total += jobs.nblk[i];
  }
  if (total == 0 || batch == 0)
    return jobs.zero_words ? launch_zero_words(zero_ptr, (size_t)zero_words * 4, 0, 1, stream) : hipSuccess;
  if (total > 0x7FFFFFFFull) return hipErrorInvalidValue;
  const size_t lds = any_kcontig ? sizeof(double) * waves * RES_TILE_DOUBLES : 0;
  const dim3 grid((unsigned)total, 1, batch), block(64 * waves);
  const bool low = S * L > 64; // some slice reaches into the lower 64 bits of the shifted mantissa
  // (the LOW form - S * L > 64: fp64_int8_10 and up - keeps the one instantiation: its 233..242 registers are the 128-bit values)
  const int forced_units = resident_units_switch();
  const size_t units = forced_units == 1 || forced_units == 2 || forced_units == 4 ? std::max<size_t>(per_wave, (size_t)forced_units) : per_wave;
#define OZ_RES_LAUNCH(RR)                                                                                        \
  do {                                                                                                           \
    if (low) hipLaunchKernelGGL((split_resident_kernel<RR, true, RES_UNITS>), grid, block, lds, stream, jobs);    \
    else if (units <= 1) hipLaunchKernelGGL((split_resident_kernel<RR, false, 1>), grid, block, lds, stream, jobs); \
    else if (units <= 2) hipLaunchKernelGGL((split_resident_kernel<RR, false, 2>), grid, block, lds, stream, jobs); \
    else hipLaunchKernelGGL((split_resident_kernel<RR, false, RES_UNITS>), grid, block, lds, stream, jobs);       \
  } while (0)
  if (R == 32) OZ_RES_LAUNCH(32);
  else if (R == 16) OZ_RES_LAUNCH(16);
  else OZ_RES_LAUNCH(8);
#undef OZ_RES_LAUNCH
  return hipGetLastError();
}

// ---- test hook: tiled planes -> reference layout [S][rows][ldo] --------------------------------------
__global__ void untile_kernel(const int8_t *__restrict__ planes, size_t rows, size_t K, int S,
                              int8_t *__restrict__ out, size_t ldo, size_t KB) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)S * rows * ldo;
  if (idx >= total) return;
  const size_t k = idx % ldo, r = (idx / ldo) % rows, s = idx / (ldo * rows);
  int8_t val = 0;
  if (k < K) {
    const size_t off = (((r / 32) * KB + k / 32) * (size_t)S + s) * FRAG_BYTES + ((k / 16) & 1) * 512 +
                       (r & 31) * 16 + (k & 15);
    val = planes[off];
  }
  out[idx] = val;
}

hipError_t launch_untile(const int8_t *planes, size_t rows, size_t K, int S, int8_t *out, size_t ldo,
                         hipStream_t stream) {
  const size_t total = (size_t)S * rows * ldo;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(untile_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, planes,
                     rows, K, S, out, ldo, k_blocks(K));
  return hipGetLastError();
}

// ---- auto mode: mantissa-loss statistic (src/split.cu:317-380) ---------------------------------------
// Per non-zero element of a live row: req = (e_max + 1 - e_in) + 53; loss_S = max(0, req - S L) for S = 3 .. 18; the 16 sums
// over the operand.  Round 6: the first form gave every wave ONE 32 x 32 block - ~50 VALU instructions per element and mode
// sweep, then 96 shuffles, 16 LDS atomics and (per workgroup) 16 global atomics per 8 KiB read - and took 231 / 253 us per 8192^2
// operand where its own row-maximum pass reads the same bytes in 89 / 99 us (profiles/r6_ablate/r6i_trace_auto_8192.txt): the
// statistic cost fp64_int8_auto 5 % of an 8192^3 call.  Now: persistent waves walk the blocks with a grid stride and reduce ONCE;
// the 16 modes are 8 pairs of packed 16-bit lanes (req <= 2100, a block's 16 elements per lane sum to <= 33 600 < 2^16):
// v_pk_sub_i16, v_pk_max_i16, v_pk_add_u16 per pair and element, widened to 32 bits per block; k-contiguous operands are not
// transposed through LDS any more (an element needs its row's exponent, not its row's neighbours).  Same integer sums.
typedef short pk_i16 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));

template <bool KCONTIG>
__global__ __launch_bounds__(256) void mantissa_loss_kernel(const double *__restrict__ in, size_t rows,
                                                            size_t K, size_t sr, size_t sk,
                                                            const uint32_t *__restrict__ exps, int L,
                                                            unsigned long long *counters, size_t RB,
                                                            size_t KB, uint32_t tag) {
  __shared__ unsigned long long blk[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 16) blk[threadIdx.x] = 0;
  __syncthreads();
  pk_i16 thr[8];
#pragma unroll
  for (int i = 0; i < 8; i++) thr[i] = pk_i16{(short)((3 + 2 * i) * L), (short)((4 + 2 * i) * L)};
  unsigned long long loss[16];
#pragma unroll
  for (int i = 0; i < 16; i++) loss[i] = 0;
  const size_t nblk = RB * KB, stride = (size_t)gridDim.x * 4;
  for (size_t gw = (size_t)blockIdx.x * 4 + wave; gw < nblk; gw += stride) {
    const size_t rb = KCONTIG ? gw / KB : gw % RB;
    const size_t kb = KCONTIG ? gw % KB : gw / RB;
    double t[16];
    fetch_block<KCONTIG>(in, rows, K, sr, sk, rb, kb, lane, t);
    // the row of element q: row-contiguous operands: the lane's one row; k-contiguous: row 2 q + (lane >> 5) of the block
    unsigned e[KCONTIG ? 16 : 1];
    if constexpr (KCONTIG) {
      // one load of the block's 32 row words (lane & 31 -> row), handed to the lanes that need them through the LDS crossbar: 16
      // dependent global loads per lane and block made this form 40 % slower than the row-contiguous one
      const size_t rl = rb * 32 + (lane & 31);
      const unsigned mine = rl < rows ? tagged_exp(exps[rl], tag) : 0u;
#pragma unroll
      for (int q = 0; q < 16; q++) e[q] = (unsigned)__shfl((int)mine, q * 2 + (lane >> 5), 64);
    } else {
      const size_t rg = rb * 32 + (lane & 31);
      e[0] = rg < rows ? tagged_exp(exps[rg], tag) : 0u;
    }
    pk_u16 part[8];
#pragma unroll
    for (int i = 0; i < 8; i++) part[i] = pk_u16{0, 0};
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const unsigned eq = e[KCONTIG ? q : 0];
      // max_exp != 0 (src/split.cu:322); non-finite rows carry no statistic; zero elements are skipped (:322)
      const bool live = eq != 0u && eq != 0x7FFu && t[q] != 0.0;
      const short req = live ? (short)((eq + 1u - exp_field(t[q])) + 53u) : (short)0; // :325-329 (<= 2100)
      const pk_i16 r2 = pk_i16{req, req};
#pragma unroll
      for (int i = 0; i < 8; i++) { // :330-337, two modes per instruction
        const pk_i16 d = __builtin_elementwise_max(r2 - thr[i], pk_i16{0, 0});
        part[i] += __builtin_bit_cast(pk_u16, d);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      loss[2 * i] += part[i].x;
      loss[2 * i + 1] += part[i].y;
    }
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
    unsigned long long x = loss[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    if (lane == 0 && x) atomicAdd(&blk[i], x);
  }
  __syncthreads();
  if (threadIdx.x < 16 && blk[threadIdx.x]) atomicAdd(counters + threadIdx.x, blk[threadIdx.x]);
}

hipError_t launch_mantissa_loss(const OperandView &v, const uint32_t *exps, int L,
                                unsigned long long *counters, hipStream_t stream, uint32_t tag) {
  const size_t RB = (v.rows + 31) / 32, KB = k_blocks(v.K);
  if (RB * KB == 0) return hipSuccess;
  // persistent waves: eight workgroups of four waves per CU of a 256-CU part keep the memory system busy; fewer for small operands
  const unsigned grid = (unsigned)std::min<size_t>((RB * KB + 3) / 4, 2048);
  if (v.stride_k < v.stride_r)
    hipLaunchKernelGGL(mantissa_loss_kernel<true>, dim3(grid), dim3(256), 0, stream, v.in, v.rows, v.K,
                       v.stride_r, v.stride_k, exps, L, counters, RB, KB, tag);
  else
    hipLaunchKernelGGL(mantissa_loss_kernel<false>, dim3(grid), dim3(256), 0, stream, v.in, v.rows, v.K,
                       v.stride_r, v.stride_k, exps, L, counters, RB, KB, tag);
  return hipGetLastError();
}

} // namespace ozhip
