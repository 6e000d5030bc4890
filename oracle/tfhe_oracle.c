/*
 * tfhe_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See tfhe_oracle.h for scope, parity status ("parity unpinned" at ciphertext
 * level) and the reference call sites this restates.
 *
 * Build: gcc -O2 -mfma -mavx2 -ffp-contract=off -fopenmp -shared -fPIC
 * (-ffp-contract=off is REQUIRED: every fused multiply-add below is an explicit
 *  fma() so that the HIP kernels can reproduce the same IEEE-754 op sequence.)
 */
#include "tfhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define ORACLE_HAVE_AVX2 1
#else
#define ORACLE_HAVE_AVX2 0
#endif

/* Round 4: every hot loop below exists twice -- the scalar statement (the definition; `oracle_set_simd(0)` selects it) and an
 * AVX2 form that applies THE SAME IEEE-754 operation sequence to two complex points / four words per instruction
 * (vfmadd231pd is fma() per lane, vroundpd is floor() / rint(), the integer paths are 64-bit lane arithmetic).  No
 * reassociation, no different rounding: tests/test_oracle.py compares the two paths word for word. */
/* level: 0 = scalar statement, 1 = AVX2, 2 = AVX-512 (F + DQ; used only when the CPU reports it).  -1 = best available. */
static int simd_best(void) {
#if ORACLE_HAVE_AVX2
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq")) return 2;
  return 1;
#else
  return 0;
#endif
}
static int g_simd = -1;
static inline int simd_level(void) {
  if (g_simd < 0) g_simd = simd_best();
  return g_simd;
}
void oracle_set_simd(int level) {
  const int best = simd_best();
  g_simd = level < 0 || level > best ? best : level;
}
int oracle_get_simd(void) { return simd_level(); }
#define AVX512_FN __attribute__((target("avx512f,avx512dq")))

typedef struct {
  double re, im;
} cplx;

struct oracle_ctx {
  oracle_params p;
  int N, M, k, n, L;  /* M = N/2 complex points */
  int big;            /* k*N */
  cplx *twist;        /* [M]   exp(+i*pi*j/N)   (untwist after the inverse transform) */
  cplx *tw;           /* [M/2] exp(-2*pi*i*t/M) (inverse transform)                   */
  cplx *itw;          /* [M]   the same entries stage by stage: itw[half + j] = tw[j * M/(2 half)] (contiguous for the AVX2 form) */
  cplx *psi;          /* [M]   forward transform: psi[2^l + b] = twiddle of block b at the stage with 2^l blocks */
  cplx *odd_fix;      /* [M]   odd slots: square of the last-stage twiddle (Fourier key material only) */
  cplx *bsk_f;        /* [n_ggsw][L][k+1][k+1][M]      */
  uint64_t *ksk;      /* [big][ks_level][n+1]          */
  int g;              /* grouping factor (1 = classic) */
  long n_ggsw;        /* n, or (n/g) * 2^g             */
  cplx *root;         /* [2N] exp(+i*pi*t/N)  (multi-bit: Fourier image of a monomial) */
  int32_t *slot_exp;  /* [M]  slot m of a spectrum is the evaluation at root^slot_exp[m] */
};

/* ---- canonical complex arithmetic ------------------------------- */
static inline cplx cmul(cplx a, cplx w) {
  cplx r;
  r.re = fma(a.re, w.re, -(a.im * w.im));
  r.im = fma(a.re, w.im, a.im * w.re);
  return r;
}
static inline cplx cmul_conj(cplx a, cplx w) { /* a * conj(w) */
  cplx r;
  r.re = fma(a.re, w.re, a.im * w.im);
  r.im = fma(a.im, w.re, -(a.re * w.im));
  return r;
}

/* Forward transform: the twist is merged into the twiddles.  With b_j = a_j + i a_{j+M} the
 * spectrum is the evaluation of b(Y) at y_f = exp(i pi (1 - 4 f) / N), roots of Y^M = i, slot m <-> f =
 * bitrev(m).  Cooley-Tukey stages from stride M/2 down to 1 on natural-order input: block b of the stage
 * with 2^l blocks reduces modulo (Y^half -+ t), t = psi[2^l + b] = y^half of the block's first root, by the
 * fused butterfly  s = u + t v (4 FMAs),  d = 2u - s (2 FMAs).
 * LAST stage (adjacent slots): the even slot gets s = u + t v; the odd slot holds  h = v + t' u  with
 * t' = -conj(t), i.e. the true value u - t v divided by -t (|t| = 1).  A lane pair of the device kernels
 * computes both with ONE instruction sequence (own + c * partner).  Products of two such spectra are put
 * right by multiplying ONE factor's odd slots by t^2 (odd_fix) -- done once, on the Fourier key material. */
static void fft_fwd_scalar(const oracle_ctx *c, cplx *x) {
  const int M = c->M;
  for (int half = M / 2, nb = 1; half >= 1; half >>= 1, nb <<= 1) {
    for (int b = 0; b < nb; b++) {
      const cplx t = c->psi[nb + b];
      cplx *blk = x + (size_t)b * 2 * half;
      for (int j = 0; j < half; j++) {
        const cplx u = blk[j], v = blk[j + half];
        cplx s, d;
        s.re = fma(t.re, v.re, u.re);
        s.re = fma(-t.im, v.im, s.re);
        s.im = fma(t.re, v.im, u.im);
        s.im = fma(t.im, v.re, s.im);
        if (half > 1) {
          d.re = fma(2.0, u.re, -s.re);
          d.im = fma(2.0, u.im, -s.im);
        } else { /* h = v + t' u, t' = (-t.re, t.im) */
          d.re = fma(-t.re, u.re, v.re);
          d.re = fma(-t.im, u.im, d.re);
          d.im = fma(-t.re, u.im, v.im);
          d.im = fma(t.im, u.re, d.im);
        }
        blk[j] = s;
        blk[j + half] = d;
      }
    }
  }
}

/* DIT radix-2, bit-reversed order in, natural order out (unscaled: M * identity). */
static void fft_inv_scalar(const oracle_ctx *c, cplx *x) {
  const int M = c->M;
  for (int half = 1, stride = M / 2; half <= M / 2; half <<= 1, stride >>= 1) {
    for (int b = 0; b < M; b += 2 * half) {
      for (int j = 0; j < half; j++) {
        const cplx u = x[b + j], v = x[b + j + half];
        cplx s, d;
        if (half == 1) { /* first stage: the twiddle is exactly 1 */
          s.re = u.re + v.re; s.im = u.im + v.im;
          d.re = u.re - v.re; d.im = u.im - v.im;
        } else { /* fused butterfly: s = u + v*conj(w) in 4 FMAs, d = 2u - s in 2 */
          const cplx w = c->tw[j * stride];
          s.re = fma(v.re, w.re, u.re);
          s.re = fma(v.im, w.im, s.re);
          s.im = fma(v.im, w.re, u.im);
          s.im = fma(-v.re, w.im, s.im);
          d.re = fma(2.0, u.re, -s.re);
          d.im = fma(2.0, u.im, -s.im);
        }
        x[b + j] = s;
        x[b + j + half] = d;
      }
    }
  }
}

#if ORACLE_HAVE_AVX2
/* ---- AVX2 forms: a __m256d holds two complex points (re, im, re, im) --------------------------------------------- */
static inline __m256d v_swap(__m256d a) { return _mm256_permute_pd(a, 0x5); }             /* (im, re, im, re)      */
static inline __m256d v_re(__m256d a) { return _mm256_movedup_pd(a); }                   /* (re, re, re', re')    */
static inline __m256d v_im(__m256d a) { return _mm256_permute_pd(a, 0xF); }              /* (im, im, im', im')    */
#define V_NEG_EVEN _mm256_castsi256_pd(_mm256_set_epi64x(0, (long long)0x8000000000000000ull, 0, (long long)0x8000000000000000ull))
#define V_NEG_ODD _mm256_castsi256_pd(_mm256_set_epi64x((long long)0x8000000000000000ull, 0, (long long)0x8000000000000000ull, 0))
#define V_NEG_HI _mm256_castsi256_pd(_mm256_set_epi64x((long long)0x8000000000000000ull, (long long)0x8000000000000000ull, 0, 0))

/* the butterflies of fft_fwd_scalar, two per instruction (half >= 2) or one per register (half == 1) */
static void fft_fwd_avx2(const oracle_ctx *c, cplx *x) {
  const int M = c->M;
  const __m256d two = _mm256_set1_pd(2.0);
  for (int half = M / 2, nb = 1; half >= 2; half >>= 1, nb <<= 1) {
    for (int b = 0; b < nb; b++) {
      const __m256d t = _mm256_broadcast_pd((const __m128d *)&c->psi[nb + b]); /* (t.re, t.im, t.re, t.im) */
      const __m256d tre = v_re(t);                                             /* t.re everywhere          */
      const __m256d tim = _mm256_xor_pd(v_im(t), V_NEG_EVEN);                  /* (-t.im, t.im, ...)       */
      double *blk = (double *)(x + (size_t)b * 2 * half);
      for (int j = 0; j < half; j += 2) {
        const __m256d u = _mm256_loadu_pd(blk + 2 * j), v = _mm256_loadu_pd(blk + 2 * (j + half));
        __m256d sv = _mm256_fmadd_pd(tre, v, u);          /* (fma(t.re, v.re, u.re), fma(t.re, v.im, u.im))        */
        sv = _mm256_fmadd_pd(tim, v_swap(v), sv);         /* (fma(-t.im, v.im, s.re), fma(t.im, v.re, s.im))       */
        const __m256d d = _mm256_fmsub_pd(two, u, sv);    /* fma(2, u, -s)                                         */
        _mm256_storeu_pd(blk + 2 * j, sv);
        _mm256_storeu_pd(blk + 2 * (j + half), d);
      }
    }
  }
  /* last stage: register = (u, v); own + c * partner with c = (t.re, t.re, -t.re, -t.re) then (-t.im, t.im, -t.im, t.im) */
  const cplx *ps = c->psi + M / 2;
  for (int b = 0; b < M / 2; b++) {
    double *q = (double *)(x + 2 * (size_t)b);
    const __m256d own = _mm256_loadu_pd(q);
    const __m256d par = _mm256_permute2f128_pd(own, own, 0x01);               /* (v, u)                              */
    const __m256d t = _mm256_broadcast_pd((const __m128d *)&ps[b]);
    const __m256d c1 = _mm256_xor_pd(v_re(t), V_NEG_HI);                      /* (t.re, t.re, -t.re, -t.re)          */
    const __m256d c2 = _mm256_xor_pd(v_im(t), V_NEG_EVEN);                    /* (-t.im, t.im, -t.im, t.im)          */
    __m256d r = _mm256_fmadd_pd(c1, par, own);
    r = _mm256_fmadd_pd(c2, v_swap(par), r);
    _mm256_storeu_pd(q, r);
  }
}

static void fft_inv_avx2(const oracle_ctx *c, cplx *x) {
  const int M = c->M;
  const __m256d two = _mm256_set1_pd(2.0);
  for (int b = 0; b < M; b += 2) { /* first stage: (u + v, u - v) */
    double *q = (double *)(x + b);
    const __m256d own = _mm256_loadu_pd(q);
    const __m256d par = _mm256_permute2f128_pd(own, own, 0x01);
    const __m256d sum = _mm256_add_pd(own, par);  /* low half: u + v */
    const __m256d dif = _mm256_sub_pd(par, own);  /* high half: u - v */
    _mm256_storeu_pd(q, _mm256_blend_pd(sum, dif, 0xC));
  }
  for (int half = 2, stride = M / 4; half <= M / 2; half <<= 1, stride >>= 1) {
    const cplx *w = c->itw + half; /* itw[half + j] = tw[j * stride] */
    for (int b = 0; b < M; b += 2 * half) {
      double *blk = (double *)(x + b);
      for (int j = 0; j < half; j += 2) {
        const __m256d u = _mm256_loadu_pd(blk + 2 * j), v = _mm256_loadu_pd(blk + 2 * (j + half));
        const __m256d ww = _mm256_loadu_pd((const double *)(w + j));
        __m256d sv = _mm256_fmadd_pd(v, v_re(ww), u);                                    /* (fma(v.re, w.re, u.re), fma(v.im, w.re, u.im)) */
        sv = _mm256_fmadd_pd(v_swap(v), _mm256_xor_pd(v_im(ww), V_NEG_ODD), sv);         /* (fma(v.im, w.im, .),   fma(v.re, -w.im, .))    */
        const __m256d d = _mm256_fmsub_pd(two, u, sv);
        _mm256_storeu_pd(blk + 2 * j, sv);
        _mm256_storeu_pd(blk + 2 * (j + half), d);
      }
    }
  }
}

/* exact int64 -> f64 for |v| < 2^51 (the digits are below 2^31): the 2^52 + 2^51 trick */
static inline __m256d v_i64_to_f64(__m256i v) {
  const __m256i magic_i = _mm256_set1_epi64x(0x4338000000000000ll);
  const __m256d magic_d = _mm256_set1_pd(0x1.8p52);
  return _mm256_sub_pd(_mm256_castsi256_pd(_mm256_add_epi64(v, magic_i)), magic_d);
}

/* torus_from_f64 on four values */
static inline __m256i v_torus_from_f64(__m256d y) {
  const __m256d t = _mm256_mul_pd(y, _mm256_set1_pd(0x1p-64));
  const __m256d kq = _mm256_floor_pd(_mm256_add_pd(t, _mm256_set1_pd(0.5)));
  const __m256d z = _mm256_fmadd_pd(kq, _mm256_set1_pd(-0x1p64), y);
  const __m256d r = _mm256_round_pd(z, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
  const __m256d h = _mm256_floor_pd(_mm256_mul_pd(r, _mm256_set1_pd(0x1p-32)));
  const __m256d l = _mm256_fmadd_pd(h, _mm256_set1_pd(-0x1p32), r);                 /* [0, 2^32) */
  const __m128i hi = _mm256_cvttpd_epi32(h);                                          /* exact: integers in int32 range */
  const __m128i lo = _mm_xor_si128(_mm256_cvttpd_epi32(_mm256_sub_pd(l, _mm256_set1_pd(0x1p31))), _mm_set1_epi32((int)0x80000000u));
  return _mm256_or_si256(_mm256_slli_epi64(_mm256_cvtepu32_epi64(hi), 32), _mm256_cvtepu32_epi64(lo));
}
/* ---- AVX-512 forms: a __m512d holds four complex points; the same per-element operations once more ------------------ */
AVX512_FN static inline __m512d w_swap(__m512d a) { return _mm512_permute_pd(a, 0x55); }
AVX512_FN static inline __m512d w_re(__m512d a) { return _mm512_movedup_pd(a); }
AVX512_FN static inline __m512d w_im(__m512d a) { return _mm512_permute_pd(a, 0xFF); }
AVX512_FN static inline __m512d w_xor(__m512d a, __m512i m) { return _mm512_castsi512_pd(_mm512_xor_epi64(_mm512_castpd_si512(a), m)); }
#define W_SIGN ((long long)0x8000000000000000ull)
#define W_NEG_EVEN _mm512_set_epi64(0, W_SIGN, 0, W_SIGN, 0, W_SIGN, 0, W_SIGN)
#define W_NEG_ODD _mm512_set_epi64(W_SIGN, 0, W_SIGN, 0, W_SIGN, 0, W_SIGN, 0)
#define W_NEG_ODD_POINTS _mm512_set_epi64(W_SIGN, W_SIGN, 0, 0, W_SIGN, W_SIGN, 0, 0) /* complex points 1 and 3 */

/* one butterfly s = u + t v, d = 2 u - s on four points (t as its two broadcast forms) */
#define W_BFLY_FWD(u, v, tre, tim, sv, dv)          \
  do {                                              \
    sv = _mm512_fmadd_pd(tre, v, u);                \
    sv = _mm512_fmadd_pd(tim, w_swap(v), sv);       \
    dv = _mm512_fmsub_pd(two, u, sv);               \
  } while (0)

AVX512_FN static void fft_fwd_avx512(const oracle_ctx *c, cplx *x) {
  const int M = c->M;
  const __m512d two = _mm512_set1_pd(2.0);
  int half = M / 2, nb = 1;
  /* two stages per pass over the data while both halves hold at least four points: the butterflies of stage (half) and of
   * stage (half / 2) on the four quarters of a block -- the same operations on the same operands, half the loads and stores */
  for (; half >= 8; half >>= 2, nb <<= 2) {
    const int q = half / 2;
    for (int b = 0; b < nb; b++) {
      const __m512d t1 = _mm512_broadcast_f64x2(_mm_loadu_pd((const double *)&c->psi[nb + b]));
      const __m512d ta = _mm512_broadcast_f64x2(_mm_loadu_pd((const double *)&c->psi[2 * nb + 2 * b]));
      const __m512d tb = _mm512_broadcast_f64x2(_mm_loadu_pd((const double *)&c->psi[2 * nb + 2 * b + 1]));
      const __m512d t1re = w_re(t1), t1im = w_xor(w_im(t1), W_NEG_EVEN);
      const __m512d tare = w_re(ta), taim = w_xor(w_im(ta), W_NEG_EVEN);
      const __m512d tbre = w_re(tb), tbim = w_xor(w_im(tb), W_NEG_EVEN);
      double *blk = (double *)(x + (size_t)b * 2 * half);
      for (int j = 0; j < q; j += 4) {
        const __m512d x0 = _mm512_loadu_pd(blk + 2 * j), x1 = _mm512_loadu_pd(blk + 2 * (j + q));
        const __m512d x2 = _mm512_loadu_pd(blk + 2 * (j + half)), x3 = _mm512_loadu_pd(blk + 2 * (j + half + q));
        __m512d s0, d0, s1, d1, y0, y1, y2, y3;
        W_BFLY_FWD(x0, x2, t1re, t1im, s0, d0); /* stage (half): pairs (j, j + half) */
        W_BFLY_FWD(x1, x3, t1re, t1im, s1, d1);
        W_BFLY_FWD(s0, s1, tare, taim, y0, y1); /* stage (half / 2), first half-block */
        W_BFLY_FWD(d0, d1, tbre, tbim, y2, y3); /* second half-block */
        _mm512_storeu_pd(blk + 2 * j, y0);
        _mm512_storeu_pd(blk + 2 * (j + q), y1);
        _mm512_storeu_pd(blk + 2 * (j + half), y2);
        _mm512_storeu_pd(blk + 2 * (j + half + q), y3);
      }
    }
  }
  for (; half >= 4; half >>= 1, nb <<= 1) {
    for (int b = 0; b < nb; b++) {
      const __m512d t = _mm512_broadcast_f64x2(_mm_loadu_pd((const double *)&c->psi[nb + b]));
      const __m512d tre = w_re(t), tim = w_xor(w_im(t), W_NEG_EVEN);
      double *blk = (double *)(x + (size_t)b * 2 * half);
      for (int j = 0; j < half; j += 4) {
        const __m512d u = _mm512_loadu_pd(blk + 2 * j), v = _mm512_loadu_pd(blk + 2 * (j + half));
        __m512d sv, dv;
        W_BFLY_FWD(u, v, tre, tim, sv, dv);
        _mm512_storeu_pd(blk + 2 * j, sv);
        _mm512_storeu_pd(blk + 2 * (j + half), dv);
      }
    }
  }
  /* half == 2: a register is one block (u0, u1, v0, v1); two blocks are regrouped into (u0, u1, u0', u1') / (v0, v1, v0', v1') */
  for (int b = 0; b < nb; b += 2) {
    double *q = (double *)(x + (size_t)b * 4);
    const __m512d x0 = _mm512_loadu_pd(q), x1 = _mm512_loadu_pd(q + 8);
    const __m512d u = _mm512_shuffle_f64x2(x0, x1, 0x44), v = _mm512_shuffle_f64x2(x0, x1, 0xEE);
    const __m512d tt = _mm512_castpd256_pd512(_mm256_loadu_pd((const double *)&c->psi[nb + b]));
    const __m512d t = _mm512_shuffle_f64x2(tt, tt, 0x50); /* (t_b, t_b, t_b+1, t_b+1) */
    __m512d sv = _mm512_fmadd_pd(w_re(t), v, u);
    sv = _mm512_fmadd_pd(w_xor(w_im(t), W_NEG_EVEN), w_swap(v), sv);
    const __m512d d = _mm512_fmsub_pd(two, u, sv);
    _mm512_storeu_pd(q, _mm512_shuffle_f64x2(sv, d, 0x44));
    _mm512_storeu_pd(q + 8, _mm512_shuffle_f64x2(sv, d, 0xEE));
  }
  /* last stage: a register is two pairs (u, v, u', v'); own + c * partner */
  const cplx *ps = c->psi + M / 2;
  for (int b = 0; b < M / 2; b += 2) {
    double *q = (double *)(x + 2 * (size_t)b);
    const __m512d own = _mm512_loadu_pd(q);
    const __m512d par = _mm512_shuffle_f64x2(own, own, 0xB1); /* (v, u, v', u') */
    const __m512d tt = _mm512_castpd256_pd512(_mm256_loadu_pd((const double *)&ps[b]));
    const __m512d t = _mm512_shuffle_f64x2(tt, tt, 0x50);
    const __m512d c1 = w_xor(w_re(t), W_NEG_ODD_POINTS); /* (t.re, t.re, -t.re, -t.re, ...) */
    const __m512d c2 = w_xor(w_im(t), W_NEG_EVEN);       /* (-t.im, t.im, -t.im, t.im, ...) */
    __m512d r = _mm512_fmadd_pd(c1, par, own);
    r = _mm512_fmadd_pd(c2, w_swap(par), r);
    _mm512_storeu_pd(q, r);
  }
}

AVX512_FN static void fft_inv_avx512(const oracle_ctx *c, cplx *x) {
  const int M = c->M;
  const __m512d two = _mm512_set1_pd(2.0);
  for (int b = 0; b < M; b += 4) { /* first stage: (u + v, u - v) twice per register */
    double *q = (double *)(x + b);
    const __m512d own = _mm512_loadu_pd(q);
    const __m512d par = _mm512_shuffle_f64x2(own, own, 0xB1);
    _mm512_storeu_pd(q, _mm512_mask_blend_pd(0xCC, _mm512_add_pd(own, par), _mm512_sub_pd(par, own)));
  }
  { /* half == 2 */
    const __m512d ww = _mm512_broadcast_f64x4(_mm256_loadu_pd((const double *)(c->itw + 2))); /* (w0, w1, w0, w1) */
    const __m512d wre = w_re(ww), wim = w_xor(w_im(ww), W_NEG_ODD);
    for (int b = 0; b < M; b += 8) {
      double *q = (double *)(x + b);
      const __m512d x0 = _mm512_loadu_pd(q), x1 = _mm512_loadu_pd(q + 8);
      const __m512d u = _mm512_shuffle_f64x2(x0, x1, 0x44), v = _mm512_shuffle_f64x2(x0, x1, 0xEE);
      __m512d sv = _mm512_fmadd_pd(v, wre, u);
      sv = _mm512_fmadd_pd(w_swap(v), wim, sv);
      const __m512d d = _mm512_fmsub_pd(two, u, sv);
      _mm512_storeu_pd(q, _mm512_shuffle_f64x2(sv, d, 0x44));
      _mm512_storeu_pd(q + 8, _mm512_shuffle_f64x2(sv, d, 0xEE));
    }
  }
#define W_BFLY_INV(u, v, ww, sv, dv)                                 \
  do {                                                               \
    sv = _mm512_fmadd_pd(v, w_re(ww), u);                            \
    sv = _mm512_fmadd_pd(w_swap(v), w_xor(w_im(ww), W_NEG_ODD), sv); \
    dv = _mm512_fmsub_pd(two, u, sv);                                \
  } while (0)
  int half = 4;
  for (; 2 * half <= M / 2; half <<= 2) { /* stages (half) and (2 half) in one pass over blocks of 4 half points */
    const cplx *wa = c->itw + half, *wb = c->itw + 2 * half;
    for (int b = 0; b < M; b += 4 * half) {
      double *blk = (double *)(x + b);
      for (int j = 0; j < half; j += 4) {
        const __m512d x0 = _mm512_loadu_pd(blk + 2 * j), x1 = _mm512_loadu_pd(blk + 2 * (j + half));
        const __m512d x2 = _mm512_loadu_pd(blk + 2 * (j + 2 * half)), x3 = _mm512_loadu_pd(blk + 2 * (j + 3 * half));
        const __m512d wA = _mm512_loadu_pd((const double *)(wa + j));
        const __m512d wB0 = _mm512_loadu_pd((const double *)(wb + j)), wB1 = _mm512_loadu_pd((const double *)(wb + j + half));
        __m512d s0, d0, s1, d1, y0, y1, y2, y3;
        W_BFLY_INV(x0, x1, wA, s0, d0);  /* stage (half): pairs (j, j + half) of both sub-blocks */
        W_BFLY_INV(x2, x3, wA, s1, d1);
        W_BFLY_INV(s0, s1, wB0, y0, y2); /* stage (2 half): pairs (j, j + 2 half) and (j + half, j + 3 half) */
        W_BFLY_INV(d0, d1, wB1, y1, y3);
        _mm512_storeu_pd(blk + 2 * j, y0);
        _mm512_storeu_pd(blk + 2 * (j + half), y1);
        _mm512_storeu_pd(blk + 2 * (j + 2 * half), y2);
        _mm512_storeu_pd(blk + 2 * (j + 3 * half), y3);
      }
    }
  }
  for (; half <= M / 2; half <<= 1) {
    const cplx *w = c->itw + half;
    for (int b = 0; b < M; b += 2 * half) {
      double *blk = (double *)(x + b);
      for (int j = 0; j < half; j += 4) {
        const __m512d u = _mm512_loadu_pd(blk + 2 * j), v = _mm512_loadu_pd(blk + 2 * (j + half));
        const __m512d ww = _mm512_loadu_pd((const double *)(w + j));
        __m512d sv, dv;
        W_BFLY_INV(u, v, ww, sv, dv);
        _mm512_storeu_pd(blk + 2 * j, sv);
        _mm512_storeu_pd(blk + 2 * (j + half), dv);
      }
    }
  }
}

AVX512_FN static inline __m512i w_torus_from_f64(__m512d y) {
  const __m512d t = _mm512_mul_pd(y, _mm512_set1_pd(0x1p-64));
  const __m512d kq = _mm512_roundscale_pd(_mm512_add_pd(t, _mm512_set1_pd(0.5)), 0x09); /* floor */
  const __m512d z = _mm512_fmadd_pd(kq, _mm512_set1_pd(-0x1p64), y);
  const __m512d r = _mm512_roundscale_pd(z, 0x08);                                      /* rint  */
  const __m512d h = _mm512_roundscale_pd(_mm512_mul_pd(r, _mm512_set1_pd(0x1p-32)), 0x09);
  const __m512d l = _mm512_fmadd_pd(h, _mm512_set1_pd(-0x1p32), r);
  const __m512i hi = _mm512_and_si512(_mm512_cvttpd_epi64(h), _mm512_set1_epi64(0xFFFFFFFFll)); /* (uint32_t)(int32_t)h */
  return _mm512_or_si512(_mm512_slli_epi64(hi, 32), _mm512_cvttpd_epu64(l));
}

/* poly_inv's closing loop: multiply by conj(twist), scale, round to the torus, split into coefficients j and j + M */
AVX512_FN static void untwist_round_avx512(const oracle_ctx *c, const cplx *spec, uint64_t *out) {
  const int M = c->M;
  const __m512d sc = _mm512_set1_pd(1.0 / (double)M);
  const __m512i idx_re = _mm512_set_epi64(14, 12, 10, 8, 6, 4, 2, 0), idx_im = _mm512_set_epi64(15, 13, 11, 9, 7, 5, 3, 1);
  for (int j = 0; j < M; j += 8) {
    __m512d r[2];
    for (int q = 0; q < 2; q++) {
      const __m512d a = _mm512_loadu_pd((const double *)(spec + j + 4 * q));
      const __m512d w = _mm512_loadu_pd((const double *)(c->twist + j + 4 * q));
      const __m512d pp = w_xor(_mm512_mul_pd(w_swap(a), w_im(w)), W_NEG_ODD);
      r[q] = _mm512_mul_pd(_mm512_fmadd_pd(a, w_re(w), pp), sc);
    }
    _mm512_storeu_si512((void *)(out + j), w_torus_from_f64(_mm512_permutex2var_pd(r[0], idx_re, r[1])));
    _mm512_storeu_si512((void *)(out + j + M), w_torus_from_f64(_mm512_permutex2var_pd(r[0], idx_im, r[1])));
  }
}

AVX512_FN static void digits_avx512(const oracle_ctx *c, const uint64_t *x, cplx *Dcc) {
  const int M = c->M, L = c->L, bl = c->p.pbs_base_log, total = bl * L;
  const int64_t B = 1ll << bl;
  const __m128i sh0 = _mm_cvtsi32_si128(64 - total - 1), shb = _mm_cvtsi32_si128(bl);
  const __m512i one = _mm512_set1_epi64(1), mtot = _mm512_set1_epi64((long long)((1ull << total) - 1));
  const __m512i mB = _mm512_set1_epi64(B - 1), vB = _mm512_set1_epi64(B), halfm1 = _mm512_set1_epi64(B / 2 - 1);
  const __m512i idx_lo = _mm512_set_epi64(11, 3, 10, 2, 9, 1, 8, 0), idx_hi = _mm512_set_epi64(15, 7, 14, 6, 13, 5, 12, 4);
  for (int j = 0; j < M; j += 8) {
    __m512i v[2];
    for (int h = 0; h < 2; h++) {
      const __m512i xx = _mm512_loadu_si512((const void *)(x + j + (size_t)h * M));
      v[h] = _mm512_and_si512(_mm512_srli_epi64(_mm512_add_epi64(_mm512_srl_epi64(xx, sh0), one), 1), mtot);
    }
    for (int l = L - 1; l >= 0; l--) {
      __m512d f[2];
      for (int h = 0; h < 2; h++) {
        __m512i d = _mm512_and_si512(v[h], mB);
        v[h] = _mm512_srl_epi64(v[h], shb);
        const __mmask8 m = _mm512_cmpgt_epi64_mask(d, halfm1);
        d = _mm512_mask_sub_epi64(d, m, d, vB);
        v[h] = _mm512_mask_add_epi64(v[h], m, v[h], one);
        f[h] = _mm512_cvtepi64_pd(d);
      }
      double *o = (double *)(Dcc + (size_t)l * M + j);
      _mm512_storeu_pd(o, _mm512_permutex2var_pd(f[0], idx_lo, f[1]));     /* (re0, im0, re1, im1, ...) */
      _mm512_storeu_pd(o + 8, _mm512_permutex2var_pd(f[0], idx_hi, f[1]));
    }
  }
}

/* k = 1, one level (every 2_2 set): O[col] = D0 * G[0][col] + D1 * G[1][col] */
AVX512_FN static void external_product_avx512(const oracle_ctx *c, const cplx *D, const cplx *G, cplx *O) {
  const int M = c->M, kp1 = c->k + 1, L = c->L, nt = kp1 * L;
  for (int m = 0; m < M; m += 4) {
    __m512d dre[12], dim[12];
    for (int t = 0; t < nt; t++) {
      const __m512d d = _mm512_loadu_pd((const double *)(D + (size_t)t * M + m));
      dre[t] = w_re(d);
      dim[t] = w_xor(w_im(d), W_NEG_EVEN);
    }
    for (int col = 0; col < kp1; col++) {
      __m512d acc = _mm512_setzero_pd();
      for (int row = 0; row < kp1; row++)
        for (int l = 0; l < L; l++) {
          const int t = row * L + l;
          const __m512d g = _mm512_loadu_pd((const double *)(G + (((size_t)l * kp1 + row) * kp1 + col) * M + m));
          acc = t == 0 ? _mm512_mul_pd(dre[0], g) : _mm512_fmadd_pd(dre[t], g, acc);
          acc = _mm512_fmadd_pd(dim[t], w_swap(g), acc);
        }
      _mm512_storeu_pd((double *)(O + (size_t)col * M + m), acc);
    }
  }
}

AVX512_FN static void bundle_fma_avx512(int M, const cplx *W, const cplx *Gc, cplx *Bq) {
  for (int m = 0; m < M; m += 4) {
    const __m512d w = _mm512_loadu_pd((const double *)(W + m));
    const __m512d x = _mm512_loadu_pd((const double *)(Gc + m));
    __m512d bb = _mm512_fmadd_pd(w_re(x), w, _mm512_loadu_pd((const double *)(Bq + m)));
    bb = _mm512_fmadd_pd(w_xor(w_im(x), W_NEG_EVEN), w_swap(w), bb);
    _mm512_storeu_pd((double *)(Bq + m), bb);
  }
}
#endif /* ORACLE_HAVE_AVX2 */

static void fft_fwd(const oracle_ctx *c, cplx *x) {
#if ORACLE_HAVE_AVX2
  if (simd_level() == 2 && c->M >= 8) { fft_fwd_avx512(c, x); return; }
  if (simd_level() && c->M >= 4) { fft_fwd_avx2(c, x); return; }
#endif
  fft_fwd_scalar(c, x);
}
static void fft_inv(const oracle_ctx *c, cplx *x) {
#if ORACLE_HAVE_AVX2
  if (simd_level() == 2 && c->M >= 8) { fft_inv_avx512(c, x); return; }
  if (simd_level() && c->M >= 4) { fft_inv_avx2(c, x); return; }
#endif
  fft_inv_scalar(c, x);
}

/* exact: rint(y) mod 2^64 for |y| < 2^100 */
static inline uint64_t torus_from_f64(double y) {
  double t = y * 0x1p-64;
  double kq = floor(t + 0.5);
  double z = fma(kq, -0x1p64, y); /* in [-2^63, 2^63) */
  double r = rint(z);
  double h = floor(r * 0x1p-32); /* in [-2^31, 2^31) */
  double l = fma(h, -0x1p32, r); /* in [0, 2^32)     */
  int32_t hi = (int32_t)h;
  uint32_t lo = (uint32_t)l;
  return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)lo;
}

static void poly_fwd(const oracle_ctx *c, const int64_t *p, cplx *out) {
  const int M = c->M;
  for (int j = 0; j < M; j++) {
    out[j].re = (double)p[j];
    out[j].im = (double)p[j + M];
  }
  fft_fwd(c, out);
}

static void poly_inv(const oracle_ctx *c, cplx *spec, uint64_t *out) {
  const int M = c->M;
  const double scale = 1.0 / (double)M; /* power of two: exact */
  fft_inv(c, spec);
#if ORACLE_HAVE_AVX2
  if (simd_level() == 2 && M >= 8) { untwist_round_avx512(c, spec, out); return; }
  if (simd_level() && M >= 4) {
    const __m256d sc = _mm256_set1_pd(scale);
    for (int j = 0; j < M; j += 4) {
      __m256d r[2];
      for (int q = 0; q < 2; q++) { /* cmul_conj: (fma(a.re, w.re, a.im * w.im), fma(a.im, w.re, -(a.re * w.im))) */
        const __m256d a = _mm256_loadu_pd((const double *)(spec + j + 2 * q));
        const __m256d w = _mm256_loadu_pd((const double *)(c->twist + j + 2 * q));
        const __m256d p = _mm256_xor_pd(_mm256_mul_pd(v_swap(a), v_im(w)), V_NEG_ODD); /* (a.im * w.im, -(a.re * w.im)) */
        r[q] = _mm256_mul_pd(_mm256_fmadd_pd(a, v_re(w), p), sc);
      }
      /* r[0] = (re0, im0, re1, im1), r[1] = (re2, im2, re3, im3) -> four real parts, four imaginary parts */
      const __m256d lo = _mm256_permute2f128_pd(r[0], r[1], 0x20), hi = _mm256_permute2f128_pd(r[0], r[1], 0x31);
      _mm256_storeu_si256((__m256i *)(out + j), v_torus_from_f64(_mm256_unpacklo_pd(lo, hi)));
      _mm256_storeu_si256((__m256i *)(out + j + M), v_torus_from_f64(_mm256_unpackhi_pd(lo, hi)));
    }
    return;
  }
#endif
  for (int j = 0; j < M; j++) {
    cplx r = cmul_conj(spec[j], c->twist[j]);
    out[j] = torus_from_f64(r.re * scale);
    out[j + M] = torus_from_f64(r.im * scale);
  }
}

/* ---- context ------------------------------------------------------------------------------ */
oracle_ctx *oracle_ctx_create(const oracle_params *p) {
  if (!p || p->poly_size < 8 || (p->poly_size & (p->poly_size - 1))) return NULL;
  if (p->glwe_dim < 1 || p->pbs_level < 1 || p->lwe_dim < 1) return NULL;
  oracle_ctx *c = (oracle_ctx *)calloc(1, sizeof(*c));
  (void)simd_level(); /* the path is chosen here, before any worker thread asks */
  c->p = *p;
  c->N = p->poly_size;
  c->M = c->N / 2;
  c->k = p->glwe_dim;
  c->n = p->lwe_dim;
  c->L = p->pbs_level;
  c->big = c->k * c->N;
  c->twist = (cplx *)malloc(sizeof(cplx) * c->M);
  c->tw = (cplx *)malloc(sizeof(cplx) * (c->M / 2));
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int j = 0; j < c->M; j++) {
    long double a = pi * (long double)j / (long double)c->N;
    c->twist[j].re = (double)cosl(a);
    c->twist[j].im = (double)sinl(a);
  }
  /* Tables are EXACTLY symmetric under quarter turns (round 3): tw[t + M/4] = -i * tw[t] = (tw[t].im, -tw[t].re)
   * as doubles, bit for bit -- only the first quarter comes from cosl / sinl --, so that an implementation may derive
   * those entries from their partners instead of loading them.  Exact values at the trivial angles, so that
   * "multiply by 1 / by -i" may be skipped without changing a numeric value. */
  for (int t = 0; t < c->M / 2; t++) {
    if (c->M >= 4 && t >= c->M / 4) {
      c->tw[t].re = c->tw[t - c->M / 4].im;
      c->tw[t].im = -c->tw[t - c->M / 4].re;
      continue;
    }
    long double a = -2.0L * pi * (long double)t / (long double)c->M;
    c->tw[t].re = (double)cosl(a);
    c->tw[t].im = (double)sinl(a);
    if (t == 0) { c->tw[0].re = 1.0; c->tw[0].im = 0.0; }
  }
  c->twist[0].re = 1.0; c->twist[0].im = 0.0;
  c->itw = (cplx *)calloc(c->M, sizeof(cplx));
  for (int half = 1, stride = c->M / 2; half <= c->M / 2; half <<= 1, stride >>= 1)
    for (int j = 0; j < half; j++) c->itw[half + j] = c->tw[(size_t)j * stride];
  c->g = p->grouping_factor > 1 ? p->grouping_factor : 1;
  if (c->g > 3 || c->n % c->g) { oracle_ctx_destroy(c); return NULL; }
  c->psi = (cplx *)calloc(c->M, sizeof(cplx));
  c->odd_fix = (cplx *)calloc(c->M, sizeof(cplx));
  c->n_ggsw = c->g > 1 ? (long)(c->n / c->g) << c->g : c->n;
  /* multi-bit tables.  The forward transform evaluates a polynomial at
   *   x_f = twist * exp(-2 pi i f / M) = exp(i pi (1 - 4 f) / N),  f = 0 .. M-1,
   * and stores the result bit-reversed (slot m holds f = bitrev(m)); X^d therefore has the
   * spectrum root[(slot_exp[m] * d) mod 2N] with root[t] = exp(i pi t / N). */
  c->root = (cplx *)malloc(sizeof(cplx) * 2 * c->N);
  c->slot_exp = (int32_t *)malloc(sizeof(int32_t) * c->M);
  /* root[t + N/2] = i * root[t] = (-root[t].im, root[t].re), bit for bit: only the first quadrant comes from cosl / sinl */
  for (int t = 0; t < 2 * c->N; t++) {
    if (t >= c->N / 2) {
      c->root[t].re = -c->root[t - c->N / 2].im;
      c->root[t].im = c->root[t - c->N / 2].re;
      continue;
    }
    long double a = pi * (long double)t / (long double)c->N;
    c->root[t].re = (double)cosl(a);
    c->root[t].im = (double)sinl(a);
    if (t == 0) { c->root[0].re = 1.0; c->root[0].im = 0.0; }
  }
  int logm = 0;
  while ((1 << logm) < c->M) logm++;
  for (int m = 0; m < c->M; m++) {
    int f = 0;
    for (int b = 0; b < logm; b++) f |= ((m >> b) & 1) << (logm - 1 - b);
    c->slot_exp[m] = (int32_t)(((1 - 4 * f) % (2 * c->N) + 2 * c->N) % (2 * c->N));
  }
  /* forward twiddles: the stage with 2^l blocks has half = M >> (l+1); the first slot of block b is
   * m0 = b * 2 * half, its root y = root[slot_exp[m0]], t = y^half = root[(slot_exp[m0] * half) mod 2N] */
  for (int l = 0; l < logm; l++) {
    const int nb = 1 << l, half = c->M >> (l + 1);
    for (int b = 0; b < nb; b++) {
      const long e = ((long)c->slot_exp[(size_t)b * 2 * half] * half) % (2 * c->N);
      c->psi[nb + b] = c->root[e];
      if (half == 1) {
        c->odd_fix[2 * b].re = 1.0;
        c->odd_fix[2 * b + 1] = c->root[(2 * e) % (2 * c->N)];
      }
    }
  }
  return c;
}

void oracle_ctx_destroy(oracle_ctx *c) {
  if (!c) return;
  free(c->twist);
  free(c->tw);
  free(c->itw);
  free(c->bsk_f);
  free(c->ksk);
  free(c->root);
  free(c->slot_exp);
  free(c->psi);
  free(c->odd_fix);
  free(c);
}

int oracle_set_keys(oracle_ctx *c, const uint64_t *bsk, const uint64_t *ksk) {
  const int kp1 = c->k + 1;
  size_t npoly = (size_t)c->n_ggsw * c->L * kp1 * kp1;
  free(c->bsk_f);
  free(c->ksk);
  c->bsk_f = (cplx *)malloc(sizeof(cplx) * npoly * c->M);
  size_t ksk_words = (size_t)c->big * c->p.ks_level * (c->n + 1);
  c->ksk = (uint64_t *)malloc(sizeof(uint64_t) * ksk_words);
  if (!c->bsk_f || !c->ksk) return -1;
  memcpy(c->ksk, ksk, sizeof(uint64_t) * ksk_words);
#pragma omp parallel for schedule(static)
  for (long q = 0; q < (long)npoly; q++) {
    /* BSK coefficients enter the FFT as signed 64-bit integers. */
    cplx *sp = c->bsk_f + (size_t)q * c->M;
    poly_fwd(c, (const int64_t *)(bsk + (size_t)q * c->N), sp);
    for (int m = 1; m < c->M; m += 2) sp[m] = cmul(sp[m], c->odd_fix[m]); /* see fft_fwd */
  }
  return 0;
}

const double *oracle_bsk_fourier(const oracle_ctx *c) { return (const double *)c->bsk_f; }
const double *oracle_twist_table(const oracle_ctx *c) { return (const double *)c->twist; }
const double *oracle_twiddle_table(const oracle_ctx *c) { return (const double *)c->tw; }
const double *oracle_root_table(const oracle_ctx *c) { return (const double *)c->root; }
const int32_t *oracle_slot_exp_table(const oracle_ctx *c) { return c->slot_exp; }

/* the TRUE spectrum (evaluations at root^slot_exp[m]): odd slots of the internal form times -t */
void oracle_fft_forward_i64(const oracle_ctx *c, const int64_t *poly, double *spec) {
  cplx *sp = (cplx *)spec;
  poly_fwd(c, poly, sp);
  for (int m = 1; m < c->M; m += 2) {
    const cplx t = c->psi[c->M / 2 + m / 2];
    const cplx mt = {-t.re, -t.im};
    sp[m] = cmul(sp[m], mt);
  }
}
/* the internal form: what the blind rotate multiplies with the Fourier key (odd slots: see fft_fwd) */
void oracle_fft_forward_internal(const oracle_ctx *c, const int64_t *poly, double *spec) {
  poly_fwd(c, poly, (cplx *)spec);
}
const double *oracle_psi_table(const oracle_ctx *c) { return (const double *)c->psi; }
const double *oracle_odd_fix_table(const oracle_ctx *c) { return (const double *)c->odd_fix; }
void oracle_fft_inverse_torus(const oracle_ctx *c, const double *spec, uint64_t *poly) {
  cplx *tmp = (cplx *)malloc(sizeof(cplx) * c->M);
  memcpy(tmp, spec, sizeof(cplx) * c->M);
  poly_inv(c, tmp, poly);
  free(tmp);
}

void oracle_negacyclic_mul_exact(const oracle_ctx *c, const int64_t *a, const uint64_t *b,
                                 uint64_t *out) {
  const int N = c->N;
  for (int t = 0; t < N; t++) {
    unsigned __int128 acc = 0;
    for (int i = 0; i < N; i++) {
      int j = t - i;
      unsigned __int128 prod = (unsigned __int128)(__int128)a[i] *
                               (unsigned __int128)(__int128)(int64_t)b[j < 0 ? j + N : j];
      if (j < 0) acc -= prod; else acc += prod;
    }
    out[t] = (uint64_t)acc;
  }
}

/* ---- LUT accumulator (tfhe shortint "generate_lookup_table") ------ */
void oracle_make_lut_poly(const oracle_ctx *c, const uint64_t *f, uint64_t *poly) {
  const int N = c->N;
  const int bits = c->p.message_bits + c->p.carry_bits;
  const int space = 1 << bits;          /* 16 */
  const int box = N / space;            /* 128 */
  const uint64_t delta = 1ull << (63 - bits); /* 2^59 */
  for (int j = 0; j < N; j++) {
    int jj = j + box / 2;
    if (jj < N) poly[j] = f[jj / box] * delta;
    else poly[j] = (uint64_t)0 - f[0] * delta;
  }
}

/* ---- keyswitch ------------------------------------------------- */
void oracle_keyswitch(const oracle_ctx *c, const uint64_t *in, uint64_t *out) {
  const int n = c->n, big = c->big, lv = c->p.ks_level, bl = c->p.ks_base_log;
  const int total = lv * bl;
  for (int j = 0; j < n; j++) out[j] = 0;
  out[n] = in[big];
  const int64_t B = 1ll << bl;
  for (int i = 0; i < big; i++) {
    /* closest representable with `total` bits, then balanced digits LSB-first */
    uint64_t v = ((in[i] >> (64 - total - 1)) + 1) >> 1;
    v &= (1ull << total) - 1;
    for (int l = lv - 1; l >= 0; l--) {
      int64_t d = (int64_t)(v & (uint64_t)(B - 1));
      v >>= bl;
      if (d >= B / 2) { d -= B; v += 1; }
      if (d == 0) continue;
      const uint64_t *row = c->ksk + ((size_t)i * lv + l) * (n + 1);
      const uint64_t du = (uint64_t)d;
      for (int j = 0; j <= n; j++) out[j] -= du * row[j];
    }
  }
}

void oracle_modswitch(const oracle_ctx *c, const uint64_t *small, uint32_t *at) {
  int log2n = 0;
  while ((1 << log2n) < 2 * c->N) log2n++;
  for (int j = 0; j <= c->n; j++)
    at[j] = (uint32_t)((((small[j] >> (64 - log2n - 1)) + 1) >> 1) & (uint64_t)(2 * c->N - 1));
}

/* out = X^t * p  (t in [0, 2N)) in Z[X]/(X^N+1) */
static void poly_rotate(int N, const uint64_t *p, int t, uint64_t *out) {
  if (simd_level()) { /* the same map as two contiguous runs: out[j] = sgn p[j - s] (j >= s), -sgn p[j - s + N] (j < s) */
    const int s = t >= N ? t - N : t;
    if (t >= N) {
      for (int j = s; j < N; j++) out[j] = (uint64_t)0 - p[j - s];
      for (int j = 0; j < s; j++) out[j] = p[j - s + N];
    } else {
      for (int j = s; j < N; j++) out[j] = p[j - s];
      for (int j = 0; j < s; j++) out[j] = (uint64_t)0 - p[j - s + N];
    }
    return;
  }
  for (int j = 0; j < N; j++) {
    int src = j - t;
    int neg = 0;
    while (src < 0) { src += N; neg ^= 1; }
    out[j] = neg ? (uint64_t)0 - p[src] : p[src];
  }
}

/* Signed digits of the N words x (closest representable, balanced, LSB first) and their forward transforms:
 * Dcc[l][M].  `dig` is scratch for the scalar statement, int64 [L][N]. */
static void digits_fwd(const oracle_ctx *c, const uint64_t *x, cplx *Dcc, int64_t *dig) {
  const int N = c->N, M = c->M, L = c->L, bl = c->p.pbs_base_log, total = bl * L;
  const int64_t B = 1ll << bl;
#if ORACLE_HAVE_AVX2
  if (simd_level() == 2 && M >= 8) {
    digits_avx512(c, x, Dcc);
    for (int l = 0; l < L; l++) fft_fwd(c, Dcc + (size_t)l * M);
    return;
  }
  if (simd_level() && M >= 4) {
    const __m128i sh0 = _mm_cvtsi32_si128(64 - total - 1), sh1 = _mm_cvtsi32_si128(1), shb = _mm_cvtsi32_si128(bl);
    const __m256i one = _mm256_set1_epi64x(1), mtot = _mm256_set1_epi64x((long long)((1ull << total) - 1));
    const __m256i mB = _mm256_set1_epi64x(B - 1), vB = _mm256_set1_epi64x(B), halfm1 = _mm256_set1_epi64x(B / 2 - 1);
    for (int j = 0; j < M; j += 4) {
      __m256i v[2]; /* coefficients j .. j+3 (real parts) and j+M .. j+M+3 (imaginary parts) */
      for (int h = 0; h < 2; h++) {
        const __m256i xx = _mm256_loadu_si256((const __m256i *)(x + j + (size_t)h * M));
        v[h] = _mm256_and_si256(_mm256_srl_epi64(_mm256_add_epi64(_mm256_srl_epi64(xx, sh0), one), sh1), mtot);
      }
      for (int l = L - 1; l >= 0; l--) {
        __m256d f[2];
        for (int h = 0; h < 2; h++) {
          __m256i d = _mm256_and_si256(v[h], mB);
          v[h] = _mm256_srl_epi64(v[h], shb);
          const __m256i m = _mm256_cmpgt_epi64(d, halfm1); /* d >= B/2 */
          d = _mm256_sub_epi64(d, _mm256_and_si256(m, vB));
          v[h] = _mm256_sub_epi64(v[h], m);                /* += 1 where the digit wrapped */
          f[h] = v_i64_to_f64(d);
        }
        const __m256d lo = _mm256_unpacklo_pd(f[0], f[1]), hi = _mm256_unpackhi_pd(f[0], f[1]);
        double *o = (double *)(Dcc + (size_t)l * M + j);
        _mm256_storeu_pd(o, _mm256_permute2f128_pd(lo, hi, 0x20));
        _mm256_storeu_pd(o + 4, _mm256_permute2f128_pd(lo, hi, 0x31));
      }
    }
    for (int l = 0; l < L; l++) fft_fwd(c, Dcc + (size_t)l * M);
    return;
  }
#endif
  for (int j = 0; j < N; j++) {
    uint64_t v = ((x[j] >> (64 - total - 1)) + 1) >> 1;
    v &= (1ull << total) - 1;
    for (int l = L - 1; l >= 0; l--) {
      int64_t d = (int64_t)(v & (uint64_t)(B - 1));
      v >>= bl;
      if (d >= B / 2) { d -= B; v += 1; }
      dig[(size_t)l * N + j] = d;
    }
  }
  for (int l = 0; l < L; l++) poly_fwd(c, dig + (size_t)l * N, Dcc + (size_t)l * M);
}

/* O[col][m] = sum over (row, l) of D[row][l][m] * G[l][row][col][m], in that order, as the FMA chain of DESIGN.md 4.7 */
static void external_product(const oracle_ctx *c, const cplx *D, const cplx *G, cplx *O) {
  const int M = c->M, kp1 = c->k + 1, L = c->L;
#if ORACLE_HAVE_AVX2
  if (simd_level() == 2 && M >= 8) { external_product_avx512(c, D, G, O); return; }
  if (simd_level() && M >= 4) {
    for (int m = 0; m < M; m += 2) {
      __m256d dre[12], dim[12]; /* (k + 1) * L <= 4 * 3 */
      const int nt = kp1 * L;
      if (nt > 12) break;
      for (int t = 0; t < nt; t++) {
        const __m256d d = _mm256_loadu_pd((const double *)(D + (size_t)t * M + m));
        dre[t] = v_re(d);
        dim[t] = _mm256_xor_pd(v_im(d), V_NEG_EVEN); /* (-d.im, d.im) */
      }
      for (int col = 0; col < kp1; col++) {
        __m256d acc = _mm256_setzero_pd();
        for (int row = 0; row < kp1; row++)
          for (int l = 0; l < L; l++) {
            const int t = row * L + l;
            const __m256d g = _mm256_loadu_pd((const double *)(G + (((size_t)l * kp1 + row) * kp1 + col) * M + m));
            acc = t == 0 ? _mm256_mul_pd(dre[0], g) : _mm256_fmadd_pd(dre[t], g, acc);
            acc = _mm256_fmadd_pd(dim[t], v_swap(g), acc);
          }
        _mm256_storeu_pd((double *)(O + (size_t)col * M + m), acc);
      }
    }
    if (kp1 * L <= 12) return;
  }
#endif
  for (int col = 0; col < kp1; col++)
    for (int m = 0; m < M; m++) {
      double re = 0, im = 0;
      int first = 1;
      for (int row = 0; row < kp1; row++) {
        for (int l = 0; l < L; l++) {
          const cplx d = D[((size_t)row * L + l) * M + m];
          const cplx g = G[(((size_t)l * kp1 + row) * kp1 + col) * M + m];
          if (first) {
            re = d.re * g.re;
            im = d.re * g.im;
            first = 0;
          } else {
            re = fma(d.re, g.re, re);
            im = fma(d.re, g.im, im);
          }
          re = fma(-d.im, g.im, re);
          im = fma(d.im, g.re, im);
        }
      }
      O[(size_t)col * M + m].re = re;
      O[(size_t)col * M + m].im = im;
    }
}

void oracle_blind_rotate(const oracle_ctx *c, const uint32_t *at, const uint64_t *lut,
                         uint64_t *acc) {
  const int N = c->N, M = c->M, k = c->k, kp1 = k + 1, L = c->L, n = c->n;
  uint64_t *rot = (uint64_t *)malloc(sizeof(uint64_t) * N);
  int64_t *dig = (int64_t *)malloc(sizeof(int64_t) * N * L);
  cplx *D = (cplx *)malloc(sizeof(cplx) * M * kp1 * L);
  cplx *O = (cplx *)malloc(sizeof(cplx) * M * kp1);
  uint64_t *delta = (uint64_t *)malloc(sizeof(uint64_t) * N);

  memset(acc, 0, sizeof(uint64_t) * N * k);
  poly_rotate(N, lut, (2 * N - (int)at[n]) % (2 * N), acc + (size_t)k * N);

  for (int i = 0; i < n; i++) {
    const int a = (int)at[i];
    if (a == 0) continue; /* diff == 0: exact no-op */
    /* (software prefetch of the step's key row -- 64 KB for the default set -- was measured and dropped: 21.8 against 17.1 ms per
     * blind rotate; the hardware streamer does better on its own) */
    for (int cc = 0; cc < kp1; cc++) {
      uint64_t *ac = acc + (size_t)cc * N;
      poly_rotate(N, ac, a, rot);
      for (int j = 0; j < N; j++) rot[j] -= ac[j];
      digits_fwd(c, rot, D + (size_t)cc * L * M, dig);
    }
    external_product(c, D, c->bsk_f + (size_t)i * L * kp1 * kp1 * M /* [L][row][col][M] */, O);
    for (int col = 0; col < kp1; col++) {
      poly_inv(c, O + (size_t)col * M, delta);
      uint64_t *ac = acc + (size_t)col * N;
      /* NOTE: acc[col] is updated only after every diff of this iteration was taken
       * (D holds them), so column order does not matter. */
      for (int j = 0; j < N; j++) ac[j] += delta[j];
    }
  }
  free(rot); free(dig); free(D); free(O); free(delta);
}

/* Multi-bit blind rotate (tfhe-rs multi_bit_blind_rotate_assign, deterministic order).
 * `small` is the keyswitched ciphertext BEFORE the modulus switch: the monomial degree of GGSW c
 * of group j is modswitch(sum_t bit_t(c) * a_{jg+t}) -- sum first (mod 2^64), then switch.
 *   bundle[row][col] = G_0 + sum_{c>=1} G_c * X^{deg_c}        (Fourier domain, c ascending,
 *                      complex fused multiply-adds in the order of cmul)
 *   acc <- bundle (x) acc : D_row = FFT(digits(acc[row])), out[col] = sum_row D_row*bundle[row][col],
 *   acc[col] = round(IFFT(out[col]))  -- replaced, not accumulated.   (pbs_level == 1) */
static inline uint32_t modswitch1(uint64_t v, int log2n) {
  return (uint32_t)((((v >> (64 - log2n - 1)) + 1) >> 1) & (((uint64_t)1 << log2n) - 1));
}

void oracle_blind_rotate_multibit(const oracle_ctx *c, const uint64_t *small, const uint64_t *lut,
                                  uint64_t *acc) {
  const int N = c->N, M = c->M, k = c->k, kp1 = k + 1, L = c->L, n = c->n, g = c->g;
  const int nc = 1 << g;
  int log2n = 0;
  while ((1 << log2n) < 2 * N) log2n++;
  int64_t *dig = (int64_t *)malloc(sizeof(int64_t) * N * L);
  cplx *D = (cplx *)malloc(sizeof(cplx) * M * kp1 * L);
  cplx *Bn = (cplx *)malloc(sizeof(cplx) * M * kp1 * kp1 * L);
  cplx *O = (cplx *)malloc(sizeof(cplx) * M * kp1);
  cplx *W = (cplx *)malloc(sizeof(cplx) * M);

  memset(acc, 0, sizeof(uint64_t) * N * k);
  poly_rotate(N, lut, (2 * N - (int)modswitch1(small[n], log2n)) % (2 * N), acc + (size_t)k * N);

  for (int j = 0; j < n / g; j++) {
    /* bundle */
    const cplx *G0 = c->bsk_f + (size_t)j * nc * L * kp1 * kp1 * M;
    memcpy(Bn, G0, sizeof(cplx) * M * kp1 * kp1 * L);
    for (int cc = 1; cc < nc; cc++) {
      uint64_t sum = 0;
      for (int t = 0; t < g; t++)
        if ((cc >> (g - 1 - t)) & 1) sum += small[(size_t)j * g + t];
      const int deg = (int)modswitch1(sum, log2n);
      const cplx *Gc = G0 + (size_t)cc * L * kp1 * kp1 * M;
      /* Fourier image of X^deg at slot m: the exponent E(m) * deg, E(m) = 1 - 4 b, b = bitrev(m), is split
       * along FIXED bit positions of b -- b_S = the three bits [LOGM-4, LOGM-2], b_L = the rest -- and the factor is the PRODUCT of
       * two table entries, W = root[deg (1 - 4 b_S)] * root[-4 deg b_L] (the multiply of 4.2).  Exact to one rounding more than a
       * single entry; the point is the device: a thread's eight slots share b_L (one gather per monomial instead of eight) and
       * b_S is the same for a whole workgroup (scalar loads). */
      {
        int logm = 0;
        while ((1 << logm) < M) logm++;
        const long two_n = 2L * N;
        const int smask = logm >= 4 ? (7 << (logm - 4)) : 0;
        for (int m = 0; m < M; m++) {
          const long b = (((two_n + 1 - (long)c->slot_exp[m]) % two_n) / 4);   /* slot_exp[m] = (1 - 4 b) mod 2N */
          const long bs = b & smask, bl = b - bs;
          const cplx a = c->root[(int)((((1 - 4 * bs) % two_n + two_n) * (long)deg) % two_n)];
          const cplx w = c->root[(int)((((-4 * bl) % two_n + two_n) * (long)deg) % two_n)];
          W[m].re = fma(a.re, w.re, -(a.im * w.im));
          W[m].im = fma(a.re, w.im, a.im * w.re);
        }
      }
      for (int q = 0; q < L * kp1 * kp1; q++) {
        int m = 0;
#if ORACLE_HAVE_AVX2
        if (simd_level() == 2 && M >= 8) {
          bundle_fma_avx512(M, W, Gc + (size_t)q * M, Bn + (size_t)q * M);
          m = M;
        } else if (simd_level())
          for (; m + 2 <= M; m += 2) {
            const __m256d w = _mm256_loadu_pd((const double *)(W + m));
            const __m256d x = _mm256_loadu_pd((const double *)(Gc + (size_t)q * M + m));
            double *bp = (double *)(Bn + (size_t)q * M + m);
            __m256d bb = _mm256_fmadd_pd(v_re(x), w, _mm256_loadu_pd(bp));       /* (fma(x.re, w.re, b.re), fma(x.re, w.im, b.im)) */
            bb = _mm256_fmadd_pd(_mm256_xor_pd(v_im(x), V_NEG_EVEN), v_swap(w), bb); /* (fma(-x.im, w.im, .), fma(x.im, w.re, .)) */
            _mm256_storeu_pd(bp, bb);
          }
#endif
        for (; m < M; m++) {
          const cplx w = W[m];
          const cplx x = Gc[(size_t)q * M + m];
          cplx *bq = Bn + (size_t)q * M + m;
          bq->re = fma(x.re, w.re, bq->re);
          bq->re = fma(-x.im, w.im, bq->re);
          bq->im = fma(x.re, w.im, bq->im);
          bq->im = fma(x.im, w.re, bq->im);
        }
      }
    }
    /* digits of the accumulator itself */
    for (int cc = 0; cc < kp1; cc++) digits_fwd(c, acc + (size_t)cc * N, D + (size_t)cc * L * M, dig);
    external_product(c, D, Bn, O);
    for (int col = 0; col < kp1; col++)
      poly_inv(c, O + (size_t)col * M, acc + (size_t)col * N); /* every digit of this step is already in D */
  }
  free(dig); free(D); free(Bn); free(O); free(W);
}

void oracle_sample_extract(const oracle_ctx *c, const uint64_t *glwe, uint64_t *out) {
  const int N = c->N, k = c->k;
  for (int cc = 0; cc < k; cc++) {
    const uint64_t *a = glwe + (size_t)cc * N;
    out[(size_t)cc * N] = a[0];
    for (int j = 1; j < N; j++) out[(size_t)cc * N + j] = (uint64_t)0 - a[N - j];
  }
  out[(size_t)k * N] = glwe[(size_t)k * N];
}

/* out -= d * row over n1 words, d a small signed digit: the same wrap-around integers as `out[j] -= (uint64_t)d * row[j]`,
 * written with a 32-bit multiplier so that the compiler's vector form is two 32 x 32 -> 64 multiplies */
static inline void ks_axpy(uint64_t *out, const uint64_t *row, int64_t d, int n1) {
  if (d > 0) {
    const uint64_t m = (uint64_t)d;
    for (int j = 0; j < n1; j++) out[j] -= (row[j] & 0xFFFFFFFFull) * m + (((row[j] >> 32) * m) << 32);
  } else {
    const uint64_t m = (uint64_t)(-d);
    for (int j = 0; j < n1; j++) out[j] += (row[j] & 0xFFFFFFFFull) * m + (((row[j] >> 32) * m) << 32);
  }
}

/* Keyswitch of `count` ciphertexts in one pass over the key (round 4): every key row is read once for the whole group
 * instead of once per ciphertext (the key is 61 MB; a lone keyswitch is bound by streaming it).  Per ciphertext the
 * digits, their order and the wrap-around sums are those of oracle_keyswitch. */
void oracle_keyswitch_batch(const oracle_ctx *c, const uint64_t *const *ins, uint64_t *const *outs, int count) {
  const int n = c->n, big = c->big, lv = c->p.ks_level, bl = c->p.ks_base_log;
  const int total = lv * bl;
  const int64_t B = 1ll << bl;
  if (lv > 16 || B > (1ll << 31)) { /* outside what ks_axpy's 32-bit multiplier covers: the plain statement */
    for (int q = 0; q < count; q++) oracle_keyswitch(c, ins[q], outs[q]);
    return;
  }
  for (int q = 0; q < count; q++) {
    for (int j = 0; j < n; j++) outs[q][j] = 0;
    outs[q][n] = ins[q][big];
  }
  for (int i = 0; i < big; i++) {
    const uint64_t *rows = c->ksk + (size_t)i * lv * (n + 1);
    for (int q = 0; q < count; q++) {
      uint64_t v = ((ins[q][i] >> (64 - total - 1)) + 1) >> 1;
      v &= (1ull << total) - 1;
      for (int l = lv - 1; l >= 0; l--) {
        int64_t d = (int64_t)(v & (uint64_t)(B - 1));
        v >>= bl;
        if (d >= B / 2) { d -= B; v += 1; }
        if (d == 0) continue;
        ks_axpy(outs[q], rows + (size_t)l * (n + 1), d, n + 1);
      }
    }
  }
}

static void pbs_after_keyswitch(const oracle_ctx *c, const uint64_t *small, const uint64_t *lut, uint64_t *out,
                                uint32_t *at, uint64_t *glwe) {
  if (c->g > 1) {
    oracle_blind_rotate_multibit(c, small, lut, glwe);
  } else {
    oracle_modswitch(c, small, at);
    oracle_blind_rotate(c, at, lut, glwe);
  }
  oracle_sample_extract(c, glwe, out);
}

void oracle_pbs(const oracle_ctx *c, const uint64_t *in, const uint64_t *lut, uint64_t *out) {
  uint64_t *small = (uint64_t *)malloc(sizeof(uint64_t) * (c->n + 1));
  uint32_t *at = (uint32_t *)malloc(sizeof(uint32_t) * (c->n + 1));
  uint64_t *glwe = (uint64_t *)malloc(sizeof(uint64_t) * (c->k + 1) * c->N);
  oracle_keyswitch(c, in, small);
  pbs_after_keyswitch(c, small, lut, out, at, glwe);
  free(small); free(at); free(glwe);
}

#define ORACLE_KS_GROUP 8
int oracle_exec_levels(const oracle_ctx *c, uint64_t *arena, size_t stride, int32_t n_levels,
                       const int32_t *level_ptr, const int32_t *term_ptr,
                       const int32_t *term_idx, const int64_t *term_coef,
                       const uint64_t *consts, const int32_t *lut_ids,
                       const uint64_t *lut_polys, const int32_t *out_idx, int32_t n_threads) {
  const int W = c->big + 1;
  if (n_threads < 1) n_threads = 1;
  for (int lv = 0; lv < n_levels; lv++) {
    const int lo = level_ptr[lv], hi = level_ptr[lv + 1];
    /* nodes of a level are independent: a worker takes a GROUP of them through one pass over the keyswitching key
     * (wide levels only: a level narrower than the team keeps one node per worker) */
    int group = (hi - lo) / (2 * n_threads);
    group = group < 1 ? 1 : group > ORACLE_KS_GROUP ? ORACLE_KS_GROUP : group;
    const int n_groups = (hi - lo + group - 1) / group;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int gq = 0; gq < n_groups; gq++) {
      const int j0 = lo + gq * group, j1 = j0 + group < hi ? j0 + group : hi;
      uint64_t *x = (uint64_t *)calloc((size_t)W * (j1 - j0), sizeof(uint64_t));
      uint64_t *small = (uint64_t *)malloc(sizeof(uint64_t) * (c->n + 1) * (j1 - j0));
      uint32_t *at = (uint32_t *)malloc(sizeof(uint32_t) * (c->n + 1));
      uint64_t *glwe = (uint64_t *)malloc(sizeof(uint64_t) * (c->k + 1) * c->N);
      const uint64_t *ins[ORACLE_KS_GROUP];
      uint64_t *outs[ORACLE_KS_GROUP];
      int n_pbs = 0;
      for (int j = j0; j < j1; j++) {
        uint64_t *xj = x + (size_t)W * (j - j0);
        for (int t = term_ptr[j]; t < term_ptr[j + 1]; t++) {
          const uint64_t *src = arena + (size_t)term_idx[t] * stride;
          const uint64_t cf = (uint64_t)term_coef[t];
          for (int q = 0; q < W; q++) xj[q] += cf * src[q];
        }
        xj[W - 1] += consts[j];
        if (lut_ids[j] >= 0) {
          ins[n_pbs] = xj;
          outs[n_pbs] = small + (size_t)(c->n + 1) * n_pbs;
          n_pbs++;
        }
      }
      /* every input of the group is formed before any output is written: a node never reads a node of its own level */
      oracle_keyswitch_batch(c, ins, outs, n_pbs);
      n_pbs = 0;
      for (int j = j0; j < j1; j++) {
        uint64_t *dst = arena + (size_t)out_idx[j] * stride;
        if (lut_ids[j] < 0) memcpy(dst, x + (size_t)W * (j - j0), sizeof(uint64_t) * W);
        else pbs_after_keyswitch(c, outs[n_pbs++], lut_polys + (size_t)lut_ids[j] * c->N, dst, at, glwe);
      }
      free(x); free(small); free(at); free(glwe);
    }
  }
  return 0;
}

uint64_t oracle_lwe_phase(const uint64_t *ct, const uint64_t *sk, int32_t dim) {
  uint64_t acc = ct[dim];
  for (int i = 0; i < dim; i++) acc -= ct[i] * sk[i];
  return acc;
}

/* ---- independent client side (test infrastructure) ------------------------------------------
 * Key generation, encryption and decoding restated here from the published scheme, with a PRNG of
 * their own (xoshiro256**, Box-Muller), so that a parity test can run WITHOUT the product's
 * client.cpp on either side: keys made here go in through fhes_upload_keys, inputs encrypted here,
 * outputs decoded here.  Shortint encoding (tfhe-rs shortint, the set named at
 * /root/reference/src/lib.rs:45): value v in [0, 2^(message_bits+carry_bits)) sits at
 * v * 2^(63 - message_bits - carry_bits), one padding bit on top.
 * GGSW row (i, level l, row r): a GLWE encryption of zero with g_l = 2^(64 - B*(l+1)) added to the
 * constant coefficient of component r when the plaintext bit is 1 (classic: the key bit s_i;
 * multi-bit: prod_t [s_{jg+t} == bit (g-1-t) of c]).  Keyswitch row (i, l): LWE encryption under
 * the small key of S_i * 2^(64 - b*(l+1)). */
typedef struct { uint64_t s[4]; int have_spare; double spare; } orng;
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t orng_next(orng *r) {
  const uint64_t res = rotl64(r->s[1] * 5, 7) * 9, t = r->s[1] << 17;
  r->s[2] ^= r->s[0]; r->s[3] ^= r->s[1]; r->s[1] ^= r->s[2]; r->s[0] ^= r->s[3];
  r->s[2] ^= t; r->s[3] = rotl64(r->s[3], 45);
  return res;
}
static void orng_seed(orng *r, uint64_t seed, uint64_t stream) {
  uint64_t x = seed ^ (stream * 0xD1342543DE82EF95ull);
  for (int i = 0; i < 4; i++) {  /* splitmix64 */
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    r->s[i] = z ^ (z >> 31);
  }
  r->have_spare = 0;
  for (int i = 0; i < 8; i++) (void)orng_next(r);
}
static double orng_gauss(orng *r) {
  if (r->have_spare) { r->have_spare = 0; return r->spare; }
  const double u1 = ((double)(orng_next(r) >> 11) + 1.0) * 0x1p-53, u2 = (double)(orng_next(r) >> 11) * 0x1p-53;
  const double m = sqrt(-2.0 * log(u1)), a = 6.283185307179586476925286766559 * u2;
  r->spare = m * sin(a);
  r->have_spare = 1;
  return m * cos(a);
}
static uint64_t orng_noise(orng *r, double std_rel) { return (uint64_t)(int64_t)llrint(orng_gauss(r) * std_rel * 0x1p64); }

int oracle_keygen(const oracle_params *p, uint64_t seed, uint64_t *big_sk, uint64_t *small_sk, uint64_t *bsk,
                  uint64_t *ksk) {
  if (!p || !big_sk || !small_sk || !bsk || !ksk) return -1;
  const int N = p->poly_size, k = p->glwe_dim, kp1 = k + 1, L = p->pbs_level, n = p->lwe_dim, big = k * N;
  const int g = p->grouping_factor > 1 ? p->grouping_factor : 1;
  orng r;
  orng_seed(&r, seed, 1);
  for (int i = 0; i < big; i++) big_sk[i] = orng_next(&r) >> 63;
  for (int i = 0; i < n; i++) small_sk[i] = orng_next(&r) >> 63;
  const long n_ggsw = g > 1 ? ((long)(n / g) << g) : n;
  const long rows = n_ggsw * L * kp1;
#pragma omp parallel for schedule(dynamic, 8)
  for (long row_id = 0; row_id < rows; row_id++) {
    const long i = row_id / (L * kp1);
    const int l = (int)((row_id / kp1) % L), row = (int)(row_id % kp1);
    uint64_t *glwe = bsk + (size_t)row_id * kp1 * N, *body = glwe + (size_t)k * N;
    orng q;
    orng_seed(&q, seed, 0x100000000ull + (uint64_t)row_id);
    for (int j = 0; j < N; j++) body[j] = orng_noise(&q, p->glwe_noise_std);
    for (int c = 0; c < k; c++) {
      uint64_t *a = glwe + (size_t)c * N;
      for (int j = 0; j < N; j++) a[j] = orng_next(&q);
      for (int s = 0; s < N; s++) {  /* body += a * S_c, negacyclic */
        if (!big_sk[(size_t)c * N + s]) continue;
        for (int j = 0; j < N - s; j++) body[j + s] += a[j];
        for (int j = N - s; j < N; j++) body[j + s - N] -= a[j];
      }
    }
    uint64_t bit;
    if (g == 1) bit = small_sk[i];
    else {
      const long grp = i >> g;
      const int cc = (int)(i & ((1 << g) - 1));
      bit = 1;
      for (int t = 0; t < g; t++) bit &= (small_sk[grp * g + t] == (uint64_t)((cc >> (g - 1 - t)) & 1)) ? 1u : 0u;
    }
    if (bit) glwe[(size_t)row * N] += 1ull << (64 - p->pbs_base_log * (l + 1));
  }
  const long krows = (long)big * p->ks_level;
#pragma omp parallel for schedule(static)
  for (long rr = 0; rr < krows; rr++) {
    const int i = (int)(rr / p->ks_level), l = (int)(rr % p->ks_level);
    uint64_t *ct = ksk + (size_t)rr * (n + 1);
    orng q;
    orng_seed(&q, seed, 0x200000000ull + (uint64_t)rr);
    uint64_t b = orng_noise(&q, p->lwe_noise_std);
    for (int j = 0; j < n; j++) {
      ct[j] = orng_next(&q);
      if (small_sk[j]) b += ct[j];
    }
    if (big_sk[i]) b += 1ull << (64 - p->ks_base_log * (l + 1));
    ct[n] = b;
  }
  return 0;
}

void oracle_encrypt_blocks(const oracle_params *p, const uint64_t *big_sk, uint64_t seed, const uint8_t *values,
                           int64_t count, uint64_t *out) {
  const int big = p->glwe_dim * p->poly_size, bits = p->message_bits + p->carry_bits;
  for (int64_t q = 0; q < count; q++) {
    orng r;
    orng_seed(&r, seed, 0x300000000ull + (uint64_t)q);
    uint64_t *ct = out + (size_t)q * (big + 1);
    uint64_t b = orng_noise(&r, p->glwe_noise_std) + ((uint64_t)values[q] << (63 - bits));
    for (int j = 0; j < big; j++) {
      ct[j] = orng_next(&r);
      if (big_sk[j]) b += ct[j];
    }
    ct[big] = b;
  }
}

/* decoded value including the padding bit: round(phase / delta) mod 2^(bits+1) */
void oracle_decrypt_blocks(const oracle_params *p, const uint64_t *big_sk, const uint64_t *cts, int64_t count,
                           uint8_t *values) {
  const int big = p->glwe_dim * p->poly_size, bits = p->message_bits + p->carry_bits;
  for (int64_t q = 0; q < count; q++) {
    const uint64_t ph = oracle_lwe_phase(cts + (size_t)q * (big + 1), big_sk, big);
    const uint64_t delta = 1ull << (63 - bits);
    values[q] = (uint8_t)(((ph + delta / 2) >> (63 - bits)) & (uint64_t)((2u << bits) - 1));
  }
}
