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

typedef struct {
  double re, im;
} cplx;

struct oracle_ctx {
  oracle_params p;
  int N, M, k, n, L;  /* M = N/2 complex points */
  int big;            /* k*N */
  cplx *twist;        /* [M]   exp(+i*pi*j/N)   (untwist after the inverse transform) */
  cplx *tw;           /* [M/2] exp(-2*pi*i*t/M) (inverse transform)                   */
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
static void fft_fwd(const oracle_ctx *c, cplx *x) {
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
static void fft_inv(const oracle_ctx *c, cplx *x) {
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
  for (int j = 0; j < N; j++) {
    int src = j - t;
    int neg = 0;
    while (src < 0) { src += N; neg ^= 1; }
    out[j] = neg ? (uint64_t)0 - p[src] : p[src];
  }
}

void oracle_blind_rotate(const oracle_ctx *c, const uint32_t *at, const uint64_t *lut,
                         uint64_t *acc) {
  const int N = c->N, M = c->M, k = c->k, kp1 = k + 1, L = c->L, n = c->n;
  const int bl = c->p.pbs_base_log;
  const int total = bl * L;
  uint64_t *rot = (uint64_t *)malloc(sizeof(uint64_t) * N);
  int64_t *dig = (int64_t *)malloc(sizeof(int64_t) * N * L);
  cplx *D = (cplx *)malloc(sizeof(cplx) * M * kp1 * L);
  cplx *O = (cplx *)malloc(sizeof(cplx) * M);
  uint64_t *delta = (uint64_t *)malloc(sizeof(uint64_t) * N);
  const int64_t B = 1ll << bl;

  memset(acc, 0, sizeof(uint64_t) * N * k);
  poly_rotate(N, lut, (2 * N - (int)at[n]) % (2 * N), acc + (size_t)k * N);

  for (int i = 0; i < n; i++) {
    const int a = (int)at[i];
    if (a == 0) continue; /* diff == 0: exact no-op */
    for (int cc = 0; cc < kp1; cc++) {
      uint64_t *ac = acc + (size_t)cc * N;
      poly_rotate(N, ac, a, rot);
      for (int j = 0; j < N; j++) {
        uint64_t x = rot[j] - ac[j];
        uint64_t v = ((x >> (64 - total - 1)) + 1) >> 1;
        v &= (1ull << total) - 1;
        for (int l = L - 1; l >= 0; l--) {
          int64_t d = (int64_t)(v & (uint64_t)(B - 1));
          v >>= bl;
          if (d >= B / 2) { d -= B; v += 1; }
          dig[(size_t)l * N + j] = d;
        }
      }
      for (int l = 0; l < L; l++) poly_fwd(c, dig + (size_t)l * N, D + ((size_t)cc * L + l) * M);
    }
    const cplx *G = c->bsk_f + (size_t)i * L * kp1 * kp1 * M; /* [L][row][col][M] */
    for (int col = 0; col < kp1; col++) {
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
        O[m].re = re;
        O[m].im = im;
      }
      poly_inv(c, O, delta);
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
  const int bl = c->p.pbs_base_log, total = bl * L, nc = 1 << g;
  int log2n = 0;
  while ((1 << log2n) < 2 * N) log2n++;
  int64_t *dig = (int64_t *)malloc(sizeof(int64_t) * N * L);
  cplx *D = (cplx *)malloc(sizeof(cplx) * M * kp1 * L);
  cplx *Bn = (cplx *)malloc(sizeof(cplx) * M * kp1 * kp1 * L);
  cplx *O = (cplx *)malloc(sizeof(cplx) * M);
  const int64_t B = 1ll << bl;

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
      for (int q = 0; q < L * kp1 * kp1; q++)
        for (int m = 0; m < M; m++) {
          const cplx w = c->root[(int)(((long)c->slot_exp[m] * deg) % (2 * N))];
          const cplx x = Gc[(size_t)q * M + m];
          cplx *b = Bn + (size_t)q * M + m;
          b->re = fma(x.re, w.re, b->re);
          b->re = fma(-x.im, w.im, b->re);
          b->im = fma(x.re, w.im, b->im);
          b->im = fma(x.im, w.re, b->im);
        }
    }
    /* digits of the accumulator itself */
    for (int cc = 0; cc < kp1; cc++) {
      const uint64_t *ac = acc + (size_t)cc * N;
      for (int jj = 0; jj < N; jj++) {
        uint64_t v = ((ac[jj] >> (64 - total - 1)) + 1) >> 1;
        v &= (1ull << total) - 1;
        for (int l = L - 1; l >= 0; l--) {
          int64_t d = (int64_t)(v & (uint64_t)(B - 1));
          v >>= bl;
          if (d >= B / 2) { d -= B; v += 1; }
          dig[(size_t)l * N + jj] = d;
        }
      }
      for (int l = 0; l < L; l++) poly_fwd(c, dig + (size_t)l * N, D + ((size_t)cc * L + l) * M);
    }
    for (int col = 0; col < kp1; col++) {
      for (int m = 0; m < M; m++) {
        double re = 0, im = 0;
        int first = 1;
        for (int row = 0; row < kp1; row++)
          for (int l = 0; l < L; l++) {
            const cplx d = D[((size_t)row * L + l) * M + m];
            const cplx gg = Bn[(((size_t)l * kp1 + row) * kp1 + col) * M + m];
            if (first) { re = d.re * gg.re; im = d.re * gg.im; first = 0; }
            else { re = fma(d.re, gg.re, re); im = fma(d.re, gg.im, im); }
            re = fma(-d.im, gg.im, re);
            im = fma(d.im, gg.re, im);
          }
        O[m].re = re;
        O[m].im = im;
      }
      poly_inv(c, O, acc + (size_t)col * N); /* every digit of this step is already in D */
    }
  }
  free(dig); free(D); free(Bn); free(O);
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

void oracle_pbs(const oracle_ctx *c, const uint64_t *in, const uint64_t *lut, uint64_t *out) {
  uint64_t *small = (uint64_t *)malloc(sizeof(uint64_t) * (c->n + 1));
  uint32_t *at = (uint32_t *)malloc(sizeof(uint32_t) * (c->n + 1));
  uint64_t *glwe = (uint64_t *)malloc(sizeof(uint64_t) * (c->k + 1) * c->N);
  oracle_keyswitch(c, in, small);
  if (c->g > 1) {
    oracle_blind_rotate_multibit(c, small, lut, glwe);
  } else {
    oracle_modswitch(c, small, at);
    oracle_blind_rotate(c, at, lut, glwe);
  }
  oracle_sample_extract(c, glwe, out);
  free(small); free(at); free(glwe);
}

int oracle_exec_levels(const oracle_ctx *c, uint64_t *arena, size_t stride, int32_t n_levels,
                       const int32_t *level_ptr, const int32_t *term_ptr,
                       const int32_t *term_idx, const int64_t *term_coef,
                       const uint64_t *consts, const int32_t *lut_ids,
                       const uint64_t *lut_polys, const int32_t *out_idx, int32_t n_threads) {
  const int W = c->big + 1;
  if (n_threads < 1) n_threads = 1;
  for (int lv = 0; lv < n_levels; lv++) {
    const int lo = level_ptr[lv], hi = level_ptr[lv + 1];
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int j = lo; j < hi; j++) {
      uint64_t *x = (uint64_t *)calloc((size_t)W, sizeof(uint64_t));
      for (int t = term_ptr[j]; t < term_ptr[j + 1]; t++) {
        const uint64_t *src = arena + (size_t)term_idx[t] * stride;
        const uint64_t cf = (uint64_t)term_coef[t];
        for (int q = 0; q < W; q++) x[q] += cf * src[q];
      }
      x[W - 1] += consts[j];
      uint64_t *dst = arena + (size_t)out_idx[j] * stride;
      if (lut_ids[j] < 0) memcpy(dst, x, sizeof(uint64_t) * W);
      else oracle_pbs(c, x, lut_polys + (size_t)lut_ids[j] * c->N, dst);
      free(x);
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
