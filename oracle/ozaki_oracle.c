/*
 * ozaki_oracle.c — CPU restatement of the ozIMMU INT8 Ozaki-scheme DGEMM.
 * TEST INFRASTRUCTURE ONLY (see ozaki_oracle.h).  Plain C11 + OpenMP, built by oracle/Makefile with
 * -ffp-contract=off so every FP64 operation below rounds exactly where it is written.
 *
 * Citations are file:line relative to /root/reference.
 */
#include "ozaki_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

static const uint64_t EXP_MASK = 0x7FF0000000000000ull;  /* cutf fp::mask_exponent  (src/split.cu:27) */
static const uint64_t MANT_MASK = 0x000FFFFFFFFFFFFFull; /* cutf fp::mask_mantissa  (src/split.cu:165) */

static inline uint64_t d2u(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return u;
}
static inline double u2d(uint64_t u) {
  double x;
  memcpy(&x, &u, 8);
  return x;
}

int oz_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* src/split.cu:520-536 */
uint32_t oz_oracle_bits_per_int8(uint32_t k) {
  if (k == 0) return 0;
  uint32_t log2_k = 0;
  while ((1u << (log2_k + 1)) <= k && log2_k < 30) log2_k++;
  if ((1u << log2_k) != k) log2_k++;
  uint32_t l = (31 - log2_k) / 2;
  return l < 7 ? l : 7;
}

/* src/handle.cu:146-192 (names), src/cublas.cu:18-48 (unknown -> dgemm is the caller's policy) */
int oz_oracle_num_split_from_mode(const char *mode) {
  if (!mode) return -3;
  if (strcmp(mode, "dgemm") == 0) return -1;
  if (strcmp(mode, "sgemm") == 0) return -2;
  if (strcmp(mode, "fp64_int8_auto") == 0) return 0;
  const char *p = "fp64_int8_";
  size_t lp = strlen(p);
  if (strncmp(mode, p, lp) != 0) return -3;
  const char *q = mode + lp;
  if (*q == 0) return -3;
  int v = 0;
  for (; *q; ++q) {
    if (*q < '0' || *q > '9') return -3;
    v = v * 10 + (*q - '0');
    if (v > 99) return -3;
  }
  /* "fp64_int8_03" is not a reference mode name */
  if (mode[lp] == '0') return -3;
  return (v >= 3 && v <= 18) ? v : -3;
}

/* src/config.cu:85-93 */
int oz_oracle_pair_list(int S, int *a_ids, int *b_ids) {
  int p = 0;
  for (int sum = 2; sum <= S + 1; sum++) {
    for (int j = 1; j < sum; j++) {
      if (j > S || sum - j > S) continue;
      if (a_ids) a_ids[p] = j;
      if (b_ids) b_ids[p] = sum - j;
      p++;
    }
  }
  return p;
}

/* src/utils.hpp:30-39 */
size_t oz_oracle_pad4(size_t k) { return ((k + 3) / 4) * 4; }

/*
 * src/split.cu:154-185 cut_int8_core<double, __uint128_t>.
 * `max_exp_bits` = bits of 2^(e_max+1) (src/split.cu:191 x2 of the masked row maximum).
 */
static inline void cut_int8(int8_t *out, size_t inc, double a, uint64_t max_exp_bits, int S, int L,
                            int quirks) {
  const uint64_t ab = d2u(a);
  const int sign_flag = a > 0;                       /* :159 */
  uint64_t efield = ab & EXP_MASK;                   /* :162,:172 */
  const uint64_t implicit = efield ? 1ull : 0ull;    /* :161-162 */
  if (!efield && !(quirks & OZ_QUIRK_REF_SUBNORMAL)) /* fix of SURVEY §8a quirk 7: a subnormal has */
    efield = 1ull << 52;                             /* the exponent of field 1, not of field 0    */
  const u128 mantissa = (u128)((ab & MANT_MASK) | (implicit << 52)) << 75; /* :163-169 */
  const uint64_t off = (max_exp_bits - efield) >> 52;                      /* :170-173 */
  u128 shifted = off >= 128 ? (u128)0 : (mantissa >> off); /* :175; >=128 is UB there: clamp to 0 */
  for (int s = 0; s < S; s++) {                            /* :176-184 */
    const int8_t v = (int8_t)(shifted >> (128 - L));
    out[(size_t)s * inc] = (int8_t)(sign_flag ? v : -v);
    shifted <<= L;
  }
}

/* src/split.cu:13-67 + :191 + :193-242 */
void oz_oracle_split(const double *in, size_t rows, size_t K, size_t stride_r, size_t stride_k,
                     int S, int L, int8_t *planes, size_t ldo, double *max_exp, int quirks) {
  const size_t plane_elems = rows * ldo; /* :206 N = m*ldo */
#pragma omp parallel for schedule(static)
  for (long long r = 0; r < (long long)rows; r++) {
    const double *row = in + (size_t)r * stride_r;
    uint64_t emax = 0;
    for (size_t k = 0; k < K; k++) { /* :21-32: max over mask_exponent, compared as FP64 == as u64 */
      const uint64_t e = d2u(row[k * stride_k]) & EXP_MASK;
      if (e > emax) emax = e;
    }
    int8_t *orow = planes + (size_t)r * ldo;
    if (emax == EXP_MASK) {
      /* Inf/NaN in the row.  Reference: max_exp = Inf, garbage slices.
       * Product policy: the row is poisoned -- max_exp = NaN, slices 0 -> that row of C is NaN. */
      for (int s = 0; s < S; s++) memset(orow + (size_t)s * plane_elems, 0, ldo);
      max_exp[r] = u2d(0x7FF8000000000000ull);
      continue;
    }
    if (emax == 0) {
      /* all-zero / all-subnormal row: reference max_exp = 0*2 = 0 -> the row of C is exactly 0
       * whatever the slices hold; the product writes zero slices. */
      for (int s = 0; s < S; s++) memset(orow + (size_t)s * plane_elems, 0, ldo);
      max_exp[r] = 0.0;
      continue;
    }
    const double me = u2d(emax) * 2; /* :191 x2 */
    const uint64_t meb = d2u(me);    /* emax <= 0x7FE -> 2^1024 overflows to Inf only for 0x7FE */
    if (meb == EXP_MASK) {
      /* row max >= 2^1023: 2^(e+1) is not representable.  Treated like a non-finite row. */
      for (int s = 0; s < S; s++) memset(orow + (size_t)s * plane_elems, 0, ldo);
      max_exp[r] = u2d(0x7FF8000000000000ull);
      continue;
    }
    for (size_t k = 0; k < K; k++) cut_int8(orow + k, plane_elems, row[k * stride_k], meb, S, L, quirks);
    for (size_t k = K; k < ldo; k++) /* :222-232 */
      for (int s = 0; s < S; s++) orow[(size_t)s * plane_elems + k] = 0;
    max_exp[r] = me; /* :234-241 */
  }
}

/* src/split.cu:244-262 (col_major = op_n) and :266-283 (B: op flipped, m<->n swapped) */
void oz_oracle_split_A(int op_a, size_t m, size_t k, const double *a, size_t lda, int S, int L,
                       int8_t *planes, size_t ldo, double *max_exp, int quirks) {
  if (op_a == OZ_OP_N) /* A is m x k col-major: (r,kk) at a[kk*lda + r] */
    oz_oracle_split(a, m, k, 1, lda, S, L, planes, ldo, max_exp, quirks);
  else /* A stored k x m: (r,kk) at a[r*lda + kk] */
    oz_oracle_split(a, m, k, lda, 1, S, L, planes, ldo, max_exp, quirks);
}
void oz_oracle_split_B(int op_b, size_t k, size_t n, const double *b, size_t ldb, int S, int L,
                       int8_t *planes, size_t ldo, double *max_exp, int quirks) {
  if (op_b == OZ_OP_N) /* B is k x n col-major: column j contiguous in k */
    oz_oracle_split(b, n, k, ldb, 1, S, L, planes, ldo, max_exp, quirks);
  else /* B stored n x k: (kk, j) at b[kk*ldb + j] */
    oz_oracle_split(b, n, k, 1, ldb, S, L, planes, ldo, max_exp, quirks);
}

/* src/gemm.cu:315-329: OP_T x OP_N, both K-contiguous, alpha=1 beta=0, int32 exact */
void oz_oracle_int8_gemm(const int8_t *a_plane, const int8_t *b_plane, size_t m, size_t n,
                         size_t kp, int32_t *c32) {
#pragma omp parallel for schedule(static)
  for (long long j = 0; j < (long long)n; j++) {
    const int8_t *bj = b_plane + (size_t)j * kp;
    for (size_t i = 0; i < m; i++) {
      const int8_t *ai = a_plane + i * kp;
      int32_t acc = 0;
      for (size_t kk = 0; kk < kp; kk++) acc += (int32_t)ai[kk] * (int32_t)bj[kk];
      c32[(size_t)j * m + i] = acc;
    }
  }
}

void oz_oracle_diagonal_sums(const int8_t *a_planes, const int8_t *b_planes, size_t m, size_t n,
                             size_t ldo, int S, size_t k0, size_t k1, int64_t *d) {
  const size_t pa = m * ldo, pb = n * ldo;
  memset(d, 0, sizeof(int64_t) * (size_t)S * m * n);
#pragma omp parallel for schedule(static)
  for (long long j = 0; j < (long long)n; j++) {
    for (size_t i = 0; i < m; i++) {
      for (int t = 2; t <= S + 1; t++) {
        int64_t acc = 0;
        for (int ia = 1; ia < t; ia++) {
          const int ib = t - ia;
          if (ia > S || ib > S) continue;
          const int8_t *ai = a_planes + (size_t)(ia - 1) * pa + i * ldo;
          const int8_t *bj = b_planes + (size_t)(ib - 1) * pb + (size_t)j * ldo;
          int32_t s32 = 0;
          for (size_t kk = k0; kk < k1; kk++) s32 += (int32_t)ai[kk] * (int32_t)bj[kk];
          acc += s32;
        }
        d[((size_t)(t - 2) * n + (size_t)j) * m + i] = acc;
      }
    }
  }
}

/* src/utils.hpp:143-154 */
static int bad_shape(int op, size_t m, size_t n, size_t ld) { return (op == OZ_OP_N ? m : n) > ld; }

/* scale of src/gemm.cu:96-99 for rshift of :394-400 : 2^-rshift */
static double accumulate_scale(int L, int a_id, int b_id) {
  const int rshift = L * (a_id + b_id - 2) - (7 - L) * 2;
  return u2d((uint64_t)(0x3FF - rshift) << 52);
}

/* the pair loop of src/gemm.cu:385-403 (OZ_ORDER_REFERENCE) or its diagonal-grouped form: fills acc[m*n]
 * (column-major, ld = m) from K-contiguous planes [S][rows][ldo].  Returns 0, 2 on allocation failure. */
static int accumulate_products(const int8_t *ap, const int8_t *bp, size_t m, size_t n, size_t ldo, int S, int L,
                               int order, size_t kchunk, double *acc) {
  const size_t pa = m * ldo, pb = n * ldo;
  memset(acc, 0, sizeof(double) * m * n); /* :367 / :481 init_accumulator_buffer */
  if (order == OZ_ORDER_REFERENCE) {
    int32_t *c32 = (int32_t *)malloc(sizeof(int32_t) * m * n);
    if (!c32) return 2;
    int ai[200], bi[200];
    const int P = oz_oracle_pair_list(S, ai, bi);
    for (int p = 0; p < P; p++) { /* :385-403 */
      oz_oracle_int8_gemm(ap + (size_t)(ai[p] - 1) * pa, bp + (size_t)(bi[p] - 1) * pb, m, n, ldo, c32);
      const double scale = accumulate_scale(L, ai[p], bi[p]);
#pragma omp parallel for schedule(static)
      for (long long t = 0; t < (long long)(m * n); t++) /* :86-88 */
        acc[t] += (double)((int64_t)c32[t] << 32) * scale;
    }
    free(c32);
    return 0;
  }
  /* diagonal grouping: per K-chunk, D_t = sum_{i+j=t} A_i B_j^T exactly (integers), then
   * acc = fma((double)D_t, 2^(46-L*t), acc), t ascending, chunks outer. */
  if (kchunk == 0 || kchunk > ldo) kchunk = ldo ? ldo : 1;
  int64_t *dsum = (int64_t *)malloc(sizeof(int64_t) * (size_t)S * m * n);
  if (!dsum) return 2;
  for (size_t k0 = 0; k0 < ldo; k0 += kchunk) {
    const size_t k1 = k0 + kchunk < ldo ? k0 + kchunk : ldo;
    oz_oracle_diagonal_sums(ap, bp, m, n, ldo, S, k0, k1, dsum);
    for (int t = 2; t <= S + 1; t++) {
      /* same power of two the reference applies to a pair on this diagonal: 2^32 * 2^-rshift */
      const double sc = 4294967296.0 * accumulate_scale(L, 1, t - 1);
      const int64_t *dt = dsum + (size_t)(t - 2) * m * n;
#pragma omp parallel for schedule(static)
      for (long long e = 0; e < (long long)(m * n); e++) acc[e] = fma((double)dt[e], sc, acc[e]);
    }
  }
  free(dsum);
  return 0;
}

int oz_oracle_gemm(int op_a, int op_b, size_t m, size_t n, size_t k, double alpha, const double *a,
                   size_t lda, const double *b, size_t ldb, double beta, double *c, size_t ldc, int S,
                   int order, size_t kchunk, int quirks) {
  /* src/gemm.cu:535-556 */
  if (bad_shape(op_a, m, k, lda) || bad_shape(op_b, k, n, ldb) || bad_shape(OZ_OP_N, m, n, ldc))
    return 1;
  if (S < 3 || S > 18) return 1;
  if (m == 0 || n == 0) return 0;

  const int L = (int)oz_oracle_bits_per_int8((uint32_t)k); /* :357 */
  const size_t ldo = oz_oracle_pad4(k);                    /* :369-372 */
  int8_t *ap = (int8_t *)malloc((size_t)S * m * (ldo ? ldo : 1));
  int8_t *bp = (int8_t *)malloc((size_t)S * n * (ldo ? ldo : 1));
  double *ea = (double *)malloc(sizeof(double) * m);
  double *eb = (double *)malloc(sizeof(double) * n);
  double *acc = (double *)malloc(sizeof(double) * m * n);
  if (!ap || !bp || !ea || !eb || !acc) {
    free(ap), free(bp), free(ea), free(eb), free(acc);
    return 2;
  }

  oz_oracle_split_A(op_a, m, k, a, lda, S, L, ap, ldo, ea, quirks); /* :381-383 */
  oz_oracle_split_B(op_b, k, n, b, ldb, S, L, bp, ldo, eb, quirks);
  if (accumulate_products(ap, bp, m, n, ldo, S, L, order, kchunk, acc)) {
    free(ap), free(bp), free(ea), free(eb), free(acc);
    return 2;
  }

  /* src/gemm.cu:124-148 axby_kernel */
#pragma omp parallel for schedule(static)
  for (long long j = 0; j < (long long)n; j++) {
    for (size_t i = 0; i < m; i++) {
      const double x = acc[(size_t)j * m + i] / (double)(1ll << 44) * ea[i] * eb[j]; /* :140-141 */
      double *y = c + (size_t)j * ldc + i;
      if (beta != 0) /* :143-147; `a*x + b*y` is contracted by nvcc: defined here as fma(a,x,b*y) */
        *y = fma(alpha, x, beta * *y);
      else
        *y = alpha * x;
    }
  }
  free(ap), free(bp), free(ea), free(eb), free(acc);
  return 0;
}

/*
 * src/gemm.cu:412-521 gemm_int8<cuDoubleComplex>; complex branches of the split: src/split.cu:69-152, :211-240;
 * init_c_complex :199-239 (with its read-after-write bug at :218-219 fixed: both components use the original c);
 * axy_complex :160-186.  Operands are interleaved (re, im) doubles; alpha, beta: {re, im}.
 */
int oz_oracle_zgemm(int op_a, int op_b, size_t m, size_t n, size_t k, const double alpha[2], const double *a,
                    size_t lda, const double *b, size_t ldb, const double beta[2], double *c, size_t ldc, int S,
                    int order, size_t kchunk, int quirks) {
  if (bad_shape(op_a, m, k, lda) || bad_shape(op_b, k, n, ldb) || bad_shape(OZ_OP_N, m, n, ldc)) return 1;
  if (S < 3 || S > 18) return 1;
  if (m == 0 || n == 0) return 0;
  /* OZ_OP_C (ozaki_oracle.h): a copy of the operand with Im negated, then the op_t path */
  double *a_conj = NULL, *b_conj = NULL;
  if (op_a == OZ_OP_C) {
    a_conj = (double *)malloc(sizeof(double) * 2 * lda * (m ? m : 1));
    for (size_t i = 0; i < lda * m; i++) a_conj[2 * i] = a[2 * i], a_conj[2 * i + 1] = -a[2 * i + 1];
    a = a_conj, op_a = OZ_OP_T;
  }
  if (op_b == OZ_OP_C) {
    b_conj = (double *)malloc(sizeof(double) * 2 * ldb * (k ? k : 1));
    for (size_t i = 0; i < ldb * k; i++) b_conj[2 * i] = b[2 * i], b_conj[2 * i + 1] = -b[2 * i + 1];
    b = b_conj, op_b = OZ_OP_T;
  }
  const int L = (int)oz_oracle_bits_per_int8((uint32_t)k); /* :425 */
  const size_t ldo = oz_oracle_pad4(k);
  const size_t pl = ldo ? ldo : 1;
  int8_t *ap[2], *bp[2];
  double *ea[2], *eb[2];
  for (int q = 0; q < 2; q++) {
    ap[q] = (int8_t *)malloc((size_t)S * m * pl);
    bp[q] = (int8_t *)malloc((size_t)S * n * pl);
    ea[q] = (double *)malloc(sizeof(double) * m);
    eb[q] = (double *)malloc(sizeof(double) * n);
  }
  double *acc = (double *)malloc(sizeof(double) * m * n);
  /* split Re and Im separately, each with its own row maxima (src/split.cu:69-152, :211-216) */
  for (int q = 0; q < 2; q++) {
    if (op_a == OZ_OP_N)
      oz_oracle_split(a + q, m, k, 2, 2 * lda, S, L, ap[q], ldo, ea[q], quirks);
    else
      oz_oracle_split(a + q, m, k, 2 * lda, 2, S, L, ap[q], ldo, ea[q], quirks);
    if (op_b == OZ_OP_N)
      oz_oracle_split(b + q, n, k, 2 * ldb, 2, S, L, bp[q], ldo, eb[q], quirks);
    else
      oz_oracle_split(b + q, n, k, 2, 2 * ldb, S, L, bp[q], ldo, eb[q], quirks);
  }
  /* :477 init_c_complex */
  const int beta_zero = beta[0] == 0 && beta[1] == 0; /* :230 */
#pragma omp parallel for schedule(static)
  for (long long j = 0; j < (long long)n; j++)
    for (size_t i = 0; i < m; i++) {
      double *y = c + 2 * ((size_t)j * ldc + i);
      if (beta_zero) {
        y[0] = 0, y[1] = 0;
      } else {
        const double cx = y[0], cy = y[1];
        y[0] = fma(cx, beta[0], -(cy * beta[1]));
        y[1] = fma(cy, beta[0], cx * beta[1]);
      }
    }
  static const int order4[4][2] = {{1, 1}, {0, 0}, {1, 0}, {0, 1}}; /* :479-480 */
  int rc = 0;
  for (int t = 0; t < 4 && !rc; t++) {
    const int pa = order4[t][0], pb = order4[t][1];
    rc = accumulate_products(ap[pa], bp[pb], m, n, ldo, S, L, order, kchunk, acc);
    double ar, ai; /* :501-512 */
    if (pa == 0 && pb == 0) {
      ar = alpha[0], ai = alpha[1];
    } else if (pa == 1 && pb == 1) {
      ar = -alpha[0], ai = -alpha[1];
    } else {
      ar = -alpha[1], ai = alpha[0];
    }
#pragma omp parallel for schedule(static)
    for (long long j = 0; j < (long long)n; j++)
      for (size_t i = 0; i < m; i++) { /* axy_complex_kernel :177-185, `a*x + y` contracted to fma */
        const double x = acc[(size_t)j * m + i] / (double)(1ll << 44) * ea[pa][i] * eb[pb][j];
        double *y = c + 2 * ((size_t)j * ldc + i);
        y[0] = fma(ar, x, y[0]);
        y[1] = fma(ai, x, y[1]);
      }
  }
  for (int q = 0; q < 2; q++) free(ap[q]), free(bp[q]), free(ea[q]), free(eb[q]);
  free(acc);
  free(a_conj), free(b_conj);
  return rc;
}

/* src/split.cu:317-350 (per element) + :352-380 (per row) */
void oz_oracle_mantissa_loss(const double *in, size_t rows, size_t K, size_t stride_r,
                             size_t stride_k, int L, uint64_t counters[16]) {
  uint64_t tot[16] = {0};
#pragma omp parallel
  {
    uint64_t loc[16] = {0};
#pragma omp for schedule(static)
    for (long long r = 0; r < (long long)rows; r++) {
      const double *row = in + (size_t)r * stride_r;
      uint64_t emax = 0;
      for (size_t k = 0; k < K; k++) {
        const uint64_t e = d2u(row[k * stride_k]) & EXP_MASK;
        if (e > emax) emax = e;
      }
      if (emax == EXP_MASK) continue; /* non-finite row: no statistic (product policy) */
      const double me = u2d(emax) * 2;
      if (me == 0) continue; /* :322 max_exp == 0 */
      const uint64_t meb = d2u(me) & EXP_MASK;
      for (size_t k = 0; k < K; k++) {
        const double v = row[k * stride_k];
        if (v == 0) continue; /* :322 */
        const uint64_t req = ((meb - (d2u(v) & EXP_MASK)) >> 52) + 53; /* :325-329 */
        for (int s = 3; s <= 18; s++) {                                  /* :330-337 */
          const uint64_t space = (uint64_t)s * (uint64_t)L;
          if (space < req) loc[s - 3] += req - space;
        }
      }
    }
#pragma omp critical
    for (int i = 0; i < 16; i++) tot[i] += loc[i];
  }
  for (int i = 0; i < 16; i++) counters[i] += tot[i];
}

/* src/split.cu:454-494 */
int oz_oracle_auto_select(int op_a, int op_b, size_t m, size_t n, size_t k, const double *a,
                          size_t lda, const double *b, size_t ldb, double threshold,
                          uint64_t counters_out[16]) {
  const int L = (int)oz_oracle_bits_per_int8((uint32_t)k); /* :461 */
  uint64_t cnt[16] = {0};                                   /* :462 (intended: all 16 zeroed) */
  if (op_a == OZ_OP_N)
    oz_oracle_mantissa_loss(a, m, k, 1, lda, L, cnt); /* :464-466 */
  else
    oz_oracle_mantissa_loss(a, m, k, lda, 1, L, cnt);
  if (op_b == OZ_OP_N)
    oz_oracle_mantissa_loss(b, n, k, ldb, 1, L, cnt); /* :468-471 (op flipped) */
  else
    oz_oracle_mantissa_loss(b, n, k, 1, ldb, L, cnt);
  if (counters_out) memcpy(counters_out, cnt, sizeof(cnt));
  const double denom = (double)(m * k + k * n);
  for (int s = 3; s <= 18; s++) /* :484-491 */
    if ((double)cnt[s - 3] / denom <= threshold) return s;
  return 0; /* :493 dgemm */
}

/* ---------------- truth ---------------- */

static inline double elemA(int op_a, const double *a, size_t lda, size_t i, size_t kk) {
  return op_a == OZ_OP_N ? a[kk * lda + i] : a[i * lda + kk];
}
static inline double elemB(int op_b, const double *b, size_t ldb, size_t kk, size_t j) {
  return op_b == OZ_OP_N ? b[j * ldb + kk] : b[kk * ldb + j];
}

void oz_oracle_gemm_ld(int op_a, int op_b, size_t m, size_t n, size_t k, double alpha,
                       const double *a, size_t lda, const double *b, size_t ldb, double beta,
                       const double *c0, size_t ldc0, long double *c_true) {
  /* pack op(A) rows and op(B) columns K-contiguous for speed */
  double *at = (double *)malloc(sizeof(double) * m * (k ? k : 1));
  double *bt = (double *)malloc(sizeof(double) * n * (k ? k : 1));
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)m; i++)
    for (size_t kk = 0; kk < k; kk++) at[(size_t)i * k + kk] = elemA(op_a, a, lda, (size_t)i, kk);
#pragma omp parallel for schedule(static)
  for (long long j = 0; j < (long long)n; j++)
    for (size_t kk = 0; kk < k; kk++) bt[(size_t)j * k + kk] = elemB(op_b, b, ldb, kk, (size_t)j);
#pragma omp parallel for schedule(static)
  for (long long j = 0; j < (long long)n; j++) {
    for (size_t i = 0; i < m; i++) {
      const double *ar = at + i * k, *bc = bt + (size_t)j * k;
      long double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      size_t kk = 0;
      for (; kk + 4 <= k; kk += 4) {
        s0 += (long double)ar[kk] * bc[kk];
        s1 += (long double)ar[kk + 1] * bc[kk + 1];
        s2 += (long double)ar[kk + 2] * bc[kk + 2];
        s3 += (long double)ar[kk + 3] * bc[kk + 3];
      }
      for (; kk < k; kk++) s0 += (long double)ar[kk] * bc[kk];
      long double v = (long double)alpha * ((s0 + s1) + (s2 + s3));
      if (beta != 0 && c0) v += (long double)beta * c0[(size_t)j * ldc0 + i];
      c_true[(size_t)j * m + i] = v;
    }
  }
  free(at), free(bt);
}

double oz_oracle_relative_residual(int op_a, int op_b, size_t m, size_t n, size_t k,
                                   const double *a, size_t lda, const double *b, size_t ldb,
                                   const double *c, size_t ldc) {
  long double *ct = (long double *)malloc(sizeof(long double) * m * n);
  oz_oracle_gemm_ld(op_a, op_b, m, n, k, 1.0, a, lda, b, ldb, 0.0, NULL, 0, ct);
  long double num = 0, den = 0;
  for (size_t j = 0; j < n; j++)
    for (size_t i = 0; i < m; i++) {
      const long double t = ct[j * m + i], d = (long double)c[j * ldc + i] - t;
      num += d * d;
      den += t * t;
    }
  free(ct);
  return (double)sqrtl(num / den);
}

double oz_oracle_relative_residual_sampled(int op_a, int op_b, size_t m, size_t n, size_t k,
                                           const double *a, size_t lda, const double *b,
                                           size_t ldb, const double *c, size_t ldc, size_t ns,
                                           const int64_t *rows, const int64_t *cols) {
  (void)m, (void)n;
  long double num = 0, den = 0;
#pragma omp parallel
  {
    long double lnum = 0, lden = 0;
#pragma omp for schedule(static)
    for (long long s = 0; s < (long long)ns; s++) {
      const size_t i = (size_t)rows[s], j = (size_t)cols[s];
      long double acc = 0;
      for (size_t kk = 0; kk < k; kk++)
        acc += (long double)elemA(op_a, a, lda, i, kk) * (long double)elemB(op_b, b, ldb, kk, j);
      const long double d = (long double)c[j * ldc + i] - acc;
      lnum += d * d;
      lden += acc * acc;
    }
#pragma omp critical
    {
      num += lnum;
      den += lden;
    }
  }
  return (double)sqrtl(num / den);
}

/* src/split.cu:454-494 with cuDoubleComplex (:367-374: Re and Im both counted; denominator m*k + k*n) */
int oz_oracle_auto_select_z(int op_a, int op_b, size_t m, size_t n, size_t k, const double *a, size_t lda,
                            const double *b, size_t ldb, double threshold, uint64_t counters_out[16]) {
  const int L = (int)oz_oracle_bits_per_int8((uint32_t)k);
  uint64_t cnt[16] = {0};
  for (int q = 0; q < 2; q++) {
    if (op_a == OZ_OP_N)
      oz_oracle_mantissa_loss(a + q, m, k, 2, 2 * lda, L, cnt);
    else
      oz_oracle_mantissa_loss(a + q, m, k, 2 * lda, 2, L, cnt);
    if (op_b == OZ_OP_N)
      oz_oracle_mantissa_loss(b + q, n, k, 2 * ldb, 2, L, cnt);
    else
      oz_oracle_mantissa_loss(b + q, n, k, 2, 2 * ldb, L, cnt);
  }
  if (counters_out) memcpy(counters_out, cnt, sizeof(cnt));
  const double denom = (double)(m * k + k * n);
  for (int s = 3; s <= 18; s++)
    if ((double)cnt[s - 3] / denom <= threshold) return s;
  return 0;
}

/* ||C - op(A)op(B)||_F / ||op(A)op(B)||_F for interleaved complex operands, on sampled entries */
double oz_oracle_relative_residual_sampled_z(int op_a, int op_b, size_t m, size_t n, size_t k, const double *a,
                                             size_t lda, const double *b, size_t ldb, const double *c, size_t ldc,
                                             size_t ns, const int64_t *rows, const int64_t *cols) {
  (void)m, (void)n;
  long double num = 0, den = 0;
#pragma omp parallel
  {
    long double lnum = 0, lden = 0;
#pragma omp for schedule(static)
    for (long long s = 0; s < (long long)ns; s++) {
      const size_t i = (size_t)rows[s], j = (size_t)cols[s];
      long double re = 0, im = 0;
      for (size_t kk = 0; kk < k; kk++) {
        const double *pa = a + 2 * (op_a == OZ_OP_N ? kk * lda + i : i * lda + kk);
        const double *pb = b + 2 * (op_b == OZ_OP_N ? j * ldb + kk : kk * ldb + j);
        const long double ai = op_a == OZ_OP_C ? -(long double)pa[1] : (long double)pa[1];
        const long double bi = op_b == OZ_OP_C ? -(long double)pb[1] : (long double)pb[1];
        re += (long double)pa[0] * pb[0] - ai * bi;
        im += (long double)pa[0] * bi + ai * pb[0];
      }
      const long double dr = (long double)c[2 * (j * ldc + i)] - re, di = (long double)c[2 * (j * ldc + i) + 1] - im;
      lnum += dr * dr + di * di;
      lden += re * re + im * im;
    }
#pragma omp critical
    {
      num += lnum;
      den += lden;
    }
  }
  return (double)sqrtl(num / den);
}
