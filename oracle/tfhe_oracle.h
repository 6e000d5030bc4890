/*
 * tfhe_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the server-side hot path that every string operation
 * of the reference bottoms out in:
 *     keyswitch -> modulus switch -> blind rotate (n CMUX) -> sample extract
 * i.e. what `tfhe::shortint::ServerKey::apply_lookup_table` does for the
 * reference's `k.k.*_parallelized` calls (reference call sites:
 * /root/reference/src/ciphertext/compare.rs:130-132, mod.rs:345-348,
 * logic.rs:15,24,49,63; parameter set named at /root/reference/src/lib.rs:45).
 *
 * PARITY STATUS: "parity unpinned" at ciphertext level.  The arithmetic lives in
 * the third-party crate `tfhe = "0.5.0"` (/root/reference/Cargo.toml:7), which is
 * not vendored under /root/reference and cannot be built here (no cargo/rustc, no
 * network).  The reference's own tests never inspect ciphertext bits and use
 * unseeded keys (/root/reference/src/ciphertext/tests/mod.rs:15-18), so no
 * ciphertext-level golden vector exists.  This file restates the published
 * CGGI/TFHE algorithm with the tfhe-rs 0.5 parameter set
 * PARAM_MESSAGE_2_CARRY_2_KS_PBS and is pinned at the level the reference pins
 * itself: decrypt(op(encrypt(x))) == std::str op(x) on the reference's test
 * tables (tests/golden/reference_tables.json).
 *
 * Multi-bit parameter sets (grouping_factor 2 / 3, BASELINE.json config 5) are restated the
 * same way from tfhe-rs's multi-bit PBS (key layout of generate_lwe_multi_bit_bootstrap_key,
 * bundle + external product of multi_bit_blind_rotate_assign; sum-then-modswitch degrees), in
 * the deterministic order documented in DESIGN.md 4.8.  The reference cannot select them
 * (/root/reference/src/lib.rs:50), so there is no reference-side table for them at all:
 * they are pinned by the same decrypted tables run under multi-bit keys.
 *
 * Checked from outside by oracle/exact_pbs.py (round 5): the same bootstrap in exact integer arithmetic, no transform --
 * integer stages bit for bit, every CMUX / multi-bit step within the rounding of the f64 FFT (tests/test_exact_pbs.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product library (fhe_string_amd/csrc) never links it.
 *
 * Canonical arithmetic ("the spec" both this file and the HIP kernels follow,
 * so that their outputs are bit-identical) is documented in DESIGN.md section 4.
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t lwe_dim;       /* n   : small LWE dimension (742)              */
  int32_t glwe_dim;      /* k   : GLWE dimension (1)                     */
  int32_t poly_size;     /* N   : polynomial size (2048)                 */
  int32_t pbs_base_log;  /* 23                                           */
  int32_t pbs_level;     /* 1                                            */
  int32_t ks_base_log;   /* 3                                            */
  int32_t ks_level;      /* 5                                            */
  int32_t message_bits;  /* 2                                            */
  int32_t carry_bits;    /* 2                                            */
  int32_t grouping_factor; /* 0/1: classic CMUX blind rotate; 2/3: multi-bit */
  double lwe_noise_std;  /* std-dev relative to the torus (7.07e-6)      */
  double glwe_noise_std; /* 2.94e-16                                     */
} oracle_params;

typedef struct oracle_ctx oracle_ctx;

/* Round 4: the hot loops have an AVX2 form that performs the scalar statement's IEEE-754 operation sequence on several
 * elements per instruction (same bits; tests/test_oracle.py compares the two).  0 selects the scalar statement. */
void oracle_set_simd(int on);
int oracle_get_simd(void);

oracle_ctx *oracle_ctx_create(const oracle_params *p);
void oracle_ctx_destroy(oracle_ctx *c);

/* Key material, standard domain.
 *   bsk: u64 [n][pbs_level][k+1 (row)][k+1 (component)][N]   (classic)
 *        u64 [n/g][2^g][pbs_level][k+1][k+1][N]               (multi-bit)
 *   ksk: u64 [k*N][ks_level][n+1]
 * The BSK is transformed to the Fourier domain here (canonical forward FFT). */
int oracle_set_keys(oracle_ctx *c, const uint64_t *bsk, const uint64_t *ksk);

/* Fourier BSK, canonical (bit-reversed) order, odd slots carrying the factor of the forward transform's
 * last stage (tfhe_oracle.c: fft_fwd):  double [n][pbs_level][k+1][k+1][N/2][2 (re,im)] */
const double *oracle_bsk_fourier(const oracle_ctx *c);
/* FFT tables: untwist[N/2][2] and inverse twiddles[N/4][2]; forward twiddles psi[N/2][2] (psi[2^l + b]:
 * block b of the stage with 2^l blocks); odd_fix[N/2][2] (odd slots: the factor applied to the Fourier key) */
const double *oracle_twist_table(const oracle_ctx *c);
const double *oracle_twiddle_table(const oracle_ctx *c);
const double *oracle_psi_table(const oracle_ctx *c);
const double *oracle_odd_fix_table(const oracle_ctx *c);

/* Canonical transforms (exposed for unit tests).  _i64: the true spectrum (slot m = evaluation at
 * root^slot_exp[m]); _internal: the form the blind rotate works on (odd slots divided by -t, see fft_fwd). */
void oracle_fft_forward_i64(const oracle_ctx *c, const int64_t *poly, double *spec /*[N/2][2]*/);
void oracle_fft_forward_internal(const oracle_ctx *c, const int64_t *poly, double *spec /*[N/2][2]*/);
void oracle_fft_inverse_torus(const oracle_ctx *c, const double *spec, uint64_t *poly);
/* Exact negacyclic product (schoolbook, __int128): out = a (small signed) * b (torus) */
void oracle_negacyclic_mul_exact(const oracle_ctx *c, const int64_t *a, const uint64_t *b,
                                 uint64_t *out);

/* LUT accumulator body polynomial for f: [0, 2^(message_bits+carry_bits)) -> same. */
void oracle_make_lut_poly(const oracle_ctx *c, const uint64_t *f_table, uint64_t *poly /*[N]*/);

void oracle_keyswitch(const oracle_ctx *c, const uint64_t *in_big /*[kN+1]*/,
                      uint64_t *out_small /*[n+1]*/);
/* `count` keyswitches in one pass over the key (same integers per ciphertext as oracle_keyswitch) */
void oracle_keyswitch_batch(const oracle_ctx *c, const uint64_t *const *ins, uint64_t *const *outs, int count);
void oracle_modswitch(const oracle_ctx *c, const uint64_t *small /*[n+1]*/,
                      uint32_t *a_tilde /*[n+1]*/);
void oracle_blind_rotate(const oracle_ctx *c, const uint32_t *a_tilde, const uint64_t *lut_poly,
                         uint64_t *glwe /*[k+1][N]*/);
/* multi-bit blind rotate (grouping_factor 2 or 3); `small` is the keyswitched ciphertext
 * u64[n+1] before the modulus switch.  BSK layout: [n/g][2^g][level][k+1][k+1][N]. */
void oracle_blind_rotate_multibit(const oracle_ctx *c, const uint64_t *small,
                                  const uint64_t *lut_poly, uint64_t *glwe /*[k+1][N]*/);
/* multi-bit tables: root[2N][2] = exp(i*pi*t/N), slot_exp[N/2] */
const double *oracle_root_table(const oracle_ctx *c);
const int32_t *oracle_slot_exp_table(const oracle_ctx *c);
void oracle_sample_extract(const oracle_ctx *c, const uint64_t *glwe, uint64_t *out_big);
/* keyswitch + modswitch + blind rotate + sample extract */
void oracle_pbs(const oracle_ctx *c, const uint64_t *in_big, const uint64_t *lut_poly,
                uint64_t *out_big);

/* Level-scheduled execution of a node list over an arena of big LWE
 * ciphertexts (arena[idx] has kN+1 words, row stride `stride` words).
 * Node j (level_ptr gives contiguous node ranges per level) computes
 *     x = sum_t term_coef[t] * arena[term_idx[t]]  (t in term_ptr[j]..term_ptr[j+1])
 *         + consts[j] on the body            (all mod 2^64)
 *     arena[out_idx[j]] = lut_ids[j] < 0 ? x : PBS(x, lut_polys[lut_ids[j]])
 * Returns 0 on success. */
int oracle_exec_levels(const oracle_ctx *c, uint64_t *arena, size_t stride, int32_t n_levels,
                       const int32_t *level_ptr, const int32_t *term_ptr,
                       const int32_t *term_idx, const int64_t *term_coef,
                       const uint64_t *consts, const int32_t *lut_ids,
                       const uint64_t *lut_polys, const int32_t *out_idx, int32_t n_threads);

/* phase = body - <mask, sk> (sk entries 0/1), for decrypt checks. */
uint64_t oracle_lwe_phase(const uint64_t *ct, const uint64_t *sk, int32_t dim);

/* Independent client side (own PRNG, no product code): secret keys as 0/1 words, standard-domain
 * BSK / KSK in the layouts of oracle_set_keys, shortint encryption and decoding (value incl. the
 * padding bit).  A parity test built on these never touches fhe_string_amd/csrc/client.cpp. */
int oracle_keygen(const oracle_params *p, uint64_t seed, uint64_t *big_sk /*[kN]*/, uint64_t *small_sk /*[n]*/,
                  uint64_t *bsk, uint64_t *ksk);
void oracle_encrypt_blocks(const oracle_params *p, const uint64_t *big_sk, uint64_t seed, const uint8_t *values,
                           int64_t count, uint64_t *out /*[count][kN+1]*/);
void oracle_decrypt_blocks(const oracle_params *p, const uint64_t *big_sk, const uint64_t *cts, int64_t count,
                           uint8_t *values);

#ifdef __cplusplus
}
#endif
#endif
