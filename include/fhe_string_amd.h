/*
 * fhe_string_amd.h -- C ABI of the MI355X-native encrypted-string engine.
 *
 * This is the drop-in boundary for the reference's hot path.  In the reference
 * (matthiasgeihs/fhe_string) every string operation is a circuit over the methods
 * of `ServerKey.k : tfhe::integer::ServerKey` (/root/reference/src/server_key.rs:13);
 * each of those methods bottoms out in batches of shortint programmable
 * bootstraps.  The ABI has three levels:
 *
 *   Level 0  kernel boundary: device context, key upload, LUT registry, ciphertext
 *            arena, `fhes_pbs_batch` (linear combination -> keyswitch -> blind
 *            rotate -> sample extract) and `fhes_linear_batch`.
 *            Replaces tfhe's `apply_lookup_table` as invoked under every
 *            `k.k.*_parallelized` call site (e.g. compare.rs:130-132,
 *            mod.rs:345-348, logic.rs:15,24,49,63).
 *   Level 1  circuit submit: a compiled, level-scheduled program (static shapes)
 *            executed with one batched launch pair per level.
 *   Level 2  the reference's own API surface, 1:1:  generate_keys / ClientKey /
 *            ServerKey / FheString / FheUsize / FheOption / FheStringSliceVector
 *            (/root/reference/src/lib.rs:34-89 and Appendix C of SURVEY.md).
 *
 * All entry points are `extern "C"`, take plain pointers/sizes, return `int`
 * status (0 = ok, negative = error; `fhes_last_error()` gives the message) or an
 * opaque handle (NULL on error).  No exceptions cross the boundary.
 * Homomorphic entry points FAIL (never fall back to the CPU) when no HIP device
 * is present.
 */
#ifndef FHE_STRING_AMD_H
#define FHE_STRING_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FHES_OK 0
#define FHES_ERR_INVALID -1      /* bad argument                                   */
#define FHES_ERR_NO_DEVICE -2    /* no HIP device / HIP runtime failure            */
#define FHES_ERR_UNSUPPORTED -3  /* parameter set not supported by the HIP kernels */
#define FHES_ERR_OOM -4
#define FHES_ERR_STATE -5        /* e.g. keys not uploaded                         */
/* FheString::new validation errors (/root/reference/src/ciphertext/mod.rs:59-70) */
#define FHES_ERR_NOT_ASCII -10
#define FHES_ERR_HAS_TERMINATOR -11
#define FHES_ERR_TOO_LONG -12
#define FHES_ERR_PAD_TOO_LONG -13
#define FHES_ERR_PAD_TOO_SHORT -14

/* ---- parameters (tfhe ClassicPBSParameters; /root/reference/src/lib.rs:45,50; with
 * grouping_factor >= 2: tfhe MultiBitPBSParameters, BASELINE.json config 5) ----------------- */
typedef struct fhes_params {
  int32_t lwe_dim;       /* n   */
  int32_t glwe_dim;      /* k   */
  int32_t poly_size;     /* N   */
  int32_t pbs_base_log;
  int32_t pbs_level;
  int32_t ks_base_log;
  int32_t ks_level;
  int32_t message_bits;  /* log2(message_modulus) */
  int32_t carry_bits;    /* log2(carry_modulus)   */
  int32_t grouping_factor; /* 0 or 1: classic PBS; 2 or 3: multi-bit PBS (n % g == 0)        */
  double lwe_noise_std;
  double glwe_noise_std;
} fhes_params;

/* PARAM_MESSAGE_2_CARRY_2_KS_PBS (the set the reference fixes at lib.rs:45). */
int fhes_params_default(fhes_params *out);
/* Small set for CPU-only tests (N=256, n=32, tiny noise).  Not secure. */
int fhes_params_toy(fhes_params *out);
/* PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_{2,3}_KS_PBS of tfhe-rs 0.5 (g = 2 or 3).  The
 * reference cannot select these (lib.rs:50 takes ClassicPBSParameters); BASELINE.json config 5
 * asks for them.  Key layout: BSK u64[n/g][2^g][level][k+1][k+1][N], GGSW c of a group encrypts
 * prod_t [s_t == bit (g-1-t) of c]. */
int fhes_params_multibit(fhes_params *out, int32_t grouping_factor);
/* toy-sized multi-bit set for CPU-only tests (g = 2: n = 32, g = 3: n = 33).  Not secure. */
int fhes_params_toy_multibit(fhes_params *out, int32_t grouping_factor);
/* Words of one big-key LWE ciphertext (k*N + 1). */
int64_t fhes_ct_words(const fhes_params *p);
int64_t fhes_bsk_words(const fhes_params *p);
int64_t fhes_ksk_words(const fhes_params *p);

/* thread-local message of the last failing call on this thread */
const char *fhes_last_error(void);

/* ============================ client side (host CPU) ====================================
 * Mirrors `generate_keys_with_params` (lib.rs:50-77) and `ClientKey`
 * (client_key.rs:15-62).  Key generation and (de)encryption are client-side in the
 * reference too and are not part of the accelerated path. */
typedef struct fhes_client_key fhes_client_key;

/* `l` = maximum string length (lib.rs:50).
 * seed == 0: production -- secret keys, server-key masks / noise and every encryption call use
 * independent 256-bit keys from the OS (getrandom); encryption randomness is never serialised.
 * seed != 0: TEST-ONLY deterministic mode (64 bits of entropy): keys and ciphertexts are reproducible
 * from the seed and a serialised counter, which is what the bit-exact parity tests need. */
fhes_client_key *fhes_client_key_generate(const fhes_params *p, uint64_t l, uint64_t seed);
void fhes_client_key_destroy(fhes_client_key *ck);
int fhes_client_key_params(const fhes_client_key *ck, fhes_params *out);
int32_t fhes_client_key_num_blocks_char(const fhes_client_key *ck);   /* lib.rs:60-61 */
int32_t fhes_client_key_num_blocks_usize(const fhes_client_key *ck);  /* lib.rs:62    */
/* secret keys as 0/1 words (tests / oracle checks only) */
const uint64_t *fhes_client_key_big_sk(const fhes_client_key *ck);    /* [k*N] */
const uint64_t *fhes_client_key_small_sk(const fhes_client_key *ck);  /* [n]   */
/* Server key material in the standard domain:
 *   bsk u64 [n][pbs_level][k+1][k+1][N],  ksk u64 [k*N][ks_level][n+1]           */
int fhes_client_key_make_server_keys(const fhes_client_key *ck, uint64_t *bsk, uint64_t *ksk);
/* Encrypt `count` shortint blocks (values < 2^(message_bits+carry_bits)) under the
 * big key; out is [count][k*N+1]. */
int fhes_client_encrypt_blocks(fhes_client_key *ck, const uint8_t *values, int64_t count,
                               uint64_t *out);
/* Decode `count` blocks to their value in [0, 2^(message_bits+carry_bits+1)). */
int fhes_client_decrypt_blocks(const fhes_client_key *ck, const uint64_t *cts, int64_t count,
                               uint8_t *values);

/* ============================ Level 0: device engine ====================================
 * Threading (the reference calls its key from nested rayon workers and requires `ServerKey: Sync`,
 * search.rs:34-40, compare.rs:27-42, mod.rs:292): every entry point may be called from any thread.
 *   - fhes_program_run / fhes_program_run_batch / fhes_string_op from SEVERAL threads on one context
 *     overlap on the device: a call that finds the context's own stream carrying another caller's work
 *     gets an execution lane of its own (stream, keyswitch scratch, slot map) and returns when its results
 *     are complete; single-instance runs of the same program that arrive while others are in flight are combined
 *     into one batched run (FHES_COMBINE=0 disables, FHES_COMBINE_WINDOW_US sets the 300 us window).  A single caller
 *     keeps the asynchronous behaviour: its runs queue on the context's stream and return at once; fhes_sync /
 *     fhes_arena_download wait.
 *   - the Level-0 batch calls (fhes_pbs_batch, fhes_linear_batch), key upload and statistics take the
 *     context's own stream exclusively and serialise with each other.
 *   - HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4: at most about four lanes make
 *     progress at once).  The library sets the variable to 16 when it is loaded unless the application has
 *     set it; an application that initialises HIP before loading the library sets it itself. */
typedef struct fhes_ctx fhes_ctx;

typedef struct fhes_stats {
  uint64_t pbs_count;        /* bootstraps executed on the device                   */
  uint64_t pbs_launches;     /* blind-rotate kernel launches                        */
  uint64_t linear_count;     /* materialised linear combinations                    */
  uint64_t arena_capacity;   /* ciphertext slots                                    */
  uint64_t arena_in_use;
  double last_ks_ms;         /* HIP-event duration of the last keyswitch launch     */
  double last_br_ms;         /* HIP-event duration of the last blind-rotate launch  */
  double total_ks_ms;        /* accumulated while timing is enabled                 */
  double total_br_ms;
  uint64_t timed_launches;   /* launch pairs included in total_*_ms                 */
  uint64_t host_bytes_up;    /* ciphertext bytes fhes_arena_upload moved host -> device   (round 5) */
  uint64_t host_bytes_down;  /* ciphertext bytes fhes_arena_download moved device -> host           */
} fhes_stats;

fhes_ctx *fhes_ctx_create(const fhes_params *p, int32_t device);
void fhes_ctx_destroy(fhes_ctx *ctx);
/* BSK is transformed to the Fourier domain on the device (kernel bsk_to_fourier). */
int fhes_upload_keys(fhes_ctx *ctx, const uint64_t *bsk, const uint64_t *ksk);
/* The inverse: the standard-domain words exactly as uploaded (or received by fhes_comm_broadcast_keys); bsk[fhes_bsk_words],
 * ksk[fhes_ksk_words].  What a server-side `ServerKey: Serialize` (/root/reference/src/server_key.rs:11) writes. */
int fhes_download_keys(fhes_ctx *ctx, uint64_t *bsk, uint64_t *ksk);
/* Register f: [0,2^(msg+carry)) -> [0,2^(msg+carry)); returns lut id >= 0. */
int32_t fhes_lut_register(fhes_ctx *ctx, const uint64_t *table, int32_t n_entries);
/* Ciphertext arena: contiguous handle ranges of big-key LWE ciphertexts. */
int32_t fhes_arena_alloc(fhes_ctx *ctx, int32_t count);
int fhes_arena_free(fhes_ctx *ctx, int32_t first, int32_t count);
int fhes_arena_upload(fhes_ctx *ctx, int32_t first, int32_t count, const uint64_t *host);
int fhes_arena_download(fhes_ctx *ctx, int32_t first, int32_t count, uint64_t *host);
/* out[j] = PBS_lut[j]( sum_t coef[t]*ct[idx[t]] + consts[j] )   (one batched launch pair)
 * consts are raw torus words added to the body. */
int fhes_pbs_batch(fhes_ctx *ctx, int32_t count, const int32_t *term_ptr,
                   const int32_t *term_idx, const int64_t *term_coef, const uint64_t *consts,
                   const int32_t *lut_ids, const int32_t *out_handles);
int fhes_linear_batch(fhes_ctx *ctx, int32_t count, const int32_t *term_ptr,
                      const int32_t *term_idx, const int64_t *term_coef, const uint64_t *consts,
                      const int32_t *out_handles);
int fhes_sync(fhes_ctx *ctx);
int fhes_get_stats(fhes_ctx *ctx, fhes_stats *out);
int fhes_reset_stats(fhes_ctx *ctx);
/* enable per-launch HIP-event timing (used by bench.py for the roofline object) */
int fhes_set_timing(fhes_ctx *ctx, int32_t enabled);
/* blind-rotate kernel selection for A/B benchmarking: 0 = auto (variant 7), 1 = simple kernel, 5 / 6 / 7 = throughput
 * kernel with three wave-priority policies (all produce the same bits; multi-bit sets: 2 = latency
 * build, 3 = throughput build; 12 = the generic kernel on a 2_2 set.  9 / 10 are timing-only ablations with WRONG
 * results and are refused unless FHES_DEV_ABLATIONS=1 is in the environment) */
int fhes_set_kernel_variant(fhes_ctx *ctx, int32_t br_variant);
/* test hooks: Fourier BSK in the oracle's canonical order [n][L][k+1][k+1][N/2][2];
 * FFT tables twist[N/2][2], twiddle[N/4][2]. */
int fhes_debug_bsk_fourier(fhes_ctx *ctx, double *out);
/* forward-transform tables: psi[N/2][2] (psi[2^l + b]: twiddle of block b of the stage with 2^l blocks),
 * odd_fix[N/2][2] (odd slots: the factor the Fourier key carries) */
int fhes_debug_fwd_tables(const fhes_params *p, double *psi, double *odd_fix);
int fhes_debug_fft_tables(const fhes_params *p, double *twist, double *twiddle);
/* test hook: keyswitch+modswitch only; a_tilde out is [count][n+1] u32 */
int fhes_debug_keyswitch(fhes_ctx *ctx, int32_t count, const int32_t *in_handles,
                         uint32_t *a_tilde);
/* test hook: the keyswitched ciphertexts themselves, words [count][n+1] */
int fhes_debug_keyswitch_words(fhes_ctx *ctx, int32_t count, const int32_t *in_handles, uint64_t *words);
/* launches of at least `min_count` ciphertexts run the keyswitch as an i8 matrix product on the matrix
 * cores (default 1 = always: measured faster at every launch size; 0 = never).  Both forms compute the
 * same integers. */
int fhes_set_keyswitch_mfma_min(fhes_ctx *ctx, int32_t min_count);

/* ============================ Level 1: compiled circuits ================================ */
typedef struct fhes_program fhes_program;

/* Operand signature of an operation: one letter per operand in the order of the reference's
 * method arguments (self first): S = FheString, U = FheUsize, I = clear integer, C = FheAsciiChar
 * (num_blocks_char blocks: "is_uppercase" / "is_lowercase" / "is_whitespace" / "char_to_lowercase" /
 * "char_to_uppercase" -> "C"; convert.rs:14-49, trim.rs:15), B = BooleanBlock ("usize_from_bool" ->
 * "B"; mod.rs:431), "substr_eq" -> "SIS" (self, clear index, pattern; compare.rs:119), V =
 * FheStringSliceVector (split.rs:21-28), which is THREE ciphertext operands is_some[n],
 * start[n][Bu], end[n][Bu] as returned by the split family.  E.g. "find" -> "SS", "splitn" ->
 * "SUS" (self, n, pattern), "replacen" -> "SSSUI", "repeat" -> "SUI" (insert.rs:20),
 * "slices_len" -> "V" (split.rs:32), "slices_get" -> "VSU" (v, v.s, i; split.rs:46),
 * "slices_truncate" -> "VU" (split.rs:86), "slices_reverse" -> "V" (split.rs:229), L = one
 * BooleanBlock per character in one object ("find_pred" / "rfind_pred" -> "L": the index scan of
 * find_pred_unchecked / rfind_pred_unchecked, search.rs:196-213, over the closure's results).
 * NULL if unknown. */
const char *fhes_op_signature(const char *op);
/* Compile a string operation for static shapes (host only, no device needed).
 *   op: the reference's method name (see fhe_string_amd/csrc/circuit/strings.cpp: kOps)
 *   shape: in operand order, the slot count (chars incl. terminator/padding) of each S
 *          operand and the value of each I operand; U operands contribute no entry
 *   num_blocks_usize: lib.rs:62 */
fhes_program *fhes_program_compile(const fhes_params *p, const char *op, const int32_t *shape,
                                   int32_t n_shape, int32_t num_blocks_usize);
void fhes_program_destroy(fhes_program *prog);
/* sizes: [0]=n_inputs [1]=n_outputs [2]=n_nodes [3]=n_levels [4]=n_terms [5]=n_luts
 *        [6]=n_slots (inputs + node outputs) [7]=n_pbs */
int fhes_program_info(const fhes_program *prog, int64_t info[8]);
/* Block counts of the result objects, in order (e.g. find -> {1, Bu}; split -> {n, n*Bu, n*Bu};
 * strip_prefix -> {1, slots*Bc}).  Returns the number of parts. */
int32_t fhes_program_out_parts(const fhes_program *prog, int32_t *parts, int32_t max_parts);
/* Degree bound and noise (in units of one bootstrap output) of every output block as the builder tracks
 * them: results cross the program boundary clean (degree < message modulus, noise <= 1), so that a
 * following operation may treat them like fresh operands.  Arrays of n_outputs entries. */
int fhes_program_out_meta(const fhes_program *prog, int32_t *degree, double *noise);
/* Dependency depth of the circuit.  info[3] of fhes_program_info counts LAUNCHES: a level wider than 2^16 bootstraps is
 * tiled into several launches by the memory-bounded schedule, so it can exceed the depth (faithful trim at 512
 * characters: 881 launches, depth 530). */
int32_t fhes_program_depth(const fhes_program *prog);
/* Arena cells one instance of the program occupies while it runs (results excluded): node
 * results share cells once every reader of the previous occupant ran in an earlier level. */
int32_t fhes_program_scratch_cells(const fhes_program *prog);
/* Export for execution elsewhere (tests run programs on the oracle / a plaintext model).
 * Slot numbering: inputs are slots [0,n_inputs); node j writes slot out_idx[j]. */
int fhes_program_export(const fhes_program *prog, int32_t *level_ptr, int32_t *term_ptr,
                        int32_t *term_idx, int64_t *term_coef, uint64_t *consts,
                        int32_t *lut_ids, int32_t *out_idx, uint64_t *lut_tables /*[n_luts][2^(message_bits+carry_bits)]: 16 entries per table for the 2_2 sets*/,
                        int32_t *out_slots /*[n_outputs], -1 = trivial*/,
                        uint64_t *out_trivial /*[n_outputs] message value if trivial*/);
/* Run on the device: inputs are arena handles (n_inputs), outputs are freshly
 * allocated arena handles (n_outputs). */
int fhes_program_run(fhes_ctx *ctx, const fhes_program *prog, const int32_t *in_handles,
                     int32_t *out_handles);
/* The same program on n_inst independent inputs in one pass: every level is one launch pair
 * over n_inst * (nodes of the level) bootstraps.  in_handles [n_inst][n_inputs],
 * out_handles [n_inst][n_outputs] (one contiguous handle range, instance-major). */
int fhes_program_run_batch(fhes_ctx *ctx, const fhes_program *prog, int32_t n_inst,
                           const int32_t *in_handles, int32_t *out_handles);

/* ============================ Level 2: reference API mirror ============================= */
typedef struct fhes_server_key fhes_server_key; /* ServerKey (server_key.rs:11-17) */
typedef struct fhes_ct fhes_ct;                 /* radix ciphertext: FheString / FheUsize /
                                                   BooleanBlock / RadixCiphertext payload */

/* generate_keys_with_params's server half: builds BSK/KSK and uploads them. */
fhes_server_key *fhes_server_key_create(const fhes_client_key *ck, int32_t device);
void fhes_server_key_destroy(fhes_server_key *sk);
fhes_ctx *fhes_server_key_ctx(fhes_server_key *sk);
/* The public facts of a key: its parameter set and the block counts of lib.rs:60-62 (`num_blocks_char`, `num_blocks_usize`;
 * crate-private fields of server_key.rs:12-17 that a binding needs after fhes_server_key_deserialize).  Any pointer may be NULL. */
int fhes_server_key_info(const fhes_server_key *sk, fhes_params *params, int32_t *num_blocks_char, int32_t *num_blocks_usize);
/* Opt-in: every later fhes_string_op on this key runs the "<op>_logdepth" form of its circuit -- the
 * serial scans of the reference (mod.rs:341-349, search.rs:91-109, compare.rs:89-96, split.rs:274-319)
 * replaced by parallel-prefix forms with identical results (tests/test_circuits_plain.py).  Default
 * off: the faithful circuits.  An operation name may also carry the suffix itself. */
int fhes_server_key_set_logdepth(fhes_server_key *sk, int32_t enabled);

/* FheString::new (mod.rs:58-92).  pad < 0 means None.  Returns NULL and sets
 * *status to one of FHES_ERR_NOT_ASCII.. on validation failure. */
fhes_ct *fhes_string_encrypt(fhes_client_key *ck, const char *s, int64_t pad, int32_t *status);
/* FheString::new_trivial (mod.rs:96-119) */
fhes_ct *fhes_string_trivial(const fhes_server_key *sk, const char *s, int32_t *status);
/* FheUsize::new (mod.rs:406) */
fhes_ct *fhes_usize_encrypt(fhes_client_key *ck, uint64_t v);
/* FheUsize::new_trivial (mod.rs:426-429) */
fhes_ct *fhes_usize_trivial(const fhes_server_key *sk, uint64_t v);
/* ClientKey::encrypt (client_key.rs:23-28) / FheAsciiChar::new (mod.rs:35-37): num_blocks_char blocks */
fhes_ct *fhes_radix_encrypt(fhes_client_key *ck, uint64_t v);
void fhes_ct_destroy(fhes_ct *ct);
int32_t fhes_ct_num_blocks(const fhes_ct *ct);
/* decrypt: FheString::decrypt (mod.rs:121-129) -> NUL-terminated into buf */
int fhes_string_decrypt(const fhes_client_key *ck, fhes_ct *s, char *buf, int64_t buf_len);
int fhes_usize_decrypt(const fhes_client_key *ck, fhes_ct *v, uint64_t *out);
int fhes_bool_decrypt(const fhes_client_key *ck, fhes_ct *b, int32_t *out);
/* ClientKey::decrypt::<T> (client_key.rs:31-33): radix value of any ciphertext object */
int fhes_radix_decrypt(const fhes_client_key *ck, fhes_ct *v, uint64_t *out);
/* raw decoded value (message + carry + padding bit) of every block; values[num_blocks] */
int fhes_ct_decrypt_blocks(const fhes_client_key *ck, fhes_ct *c, uint8_t *values);

/* ---- serialisation (the reference derives serde on ClientKey / ServerKey / FheAsciiChar / FheString /
 * FheUsize: client_key.rs:15, server_key.rs:11, ciphertext/mod.rs:30,42,401).  Versioned little-endian:
 *   client key   "FHES" | u32 version | u32 kind = 1 | fhes_params (sizeof = 56) | l | block counts | mode |
 *                randomness state of the mode | secret key bits
 *   ciphertext   "FHES" | u32 version | u32 kind = 2 | u32 n_blocks | u64 ct_words | per block {u8 degree,
 *                u8 noise level in bootstrap units} | u64 words [n_blocks][ct_words]
 *   server key   "FHES" | u32 version | u32 kind = 3 | fhes_params | u64 l | block counts |
 *                u64 bsk_words | u64 ksk_words | standard-domain BSK | KSK   (170 MB for the default set)
 * An FheOption<T> is {BooleanBlock, T} and an FheStringSliceVector {is_some[n], start[n][Bu], end[n][Bu]}
 * plus its string: each part is a ciphertext blob of its own.
 * fhes_*_serialize returns the byte size; call with buf == NULL to query it. */
int64_t fhes_client_key_serialize(const fhes_client_key *ck, uint8_t *buf, int64_t buf_len);
fhes_client_key *fhes_client_key_deserialize(const uint8_t *buf, int64_t len);
int64_t fhes_ct_serialize(fhes_ct *ct, uint8_t *buf, int64_t buf_len);
fhes_ct *fhes_ct_deserialize(const uint8_t *buf, int64_t len);
/* per-block metadata of a ciphertext object: degree (upper bound of the block's value) and noise level in
 * units of one bootstrap output (0 = fresh or trivial encryption); results of fhes_string_op are clean:
 * degree < message modulus (booleans: 1), noise <= 1.  Arrays of fhes_ct_num_blocks entries. */
int fhes_ct_block_meta(const fhes_ct *ct, uint8_t *degree, uint8_t *noise);
/* Block-level access for callers that evaluate their own per-character circuits (the closure of
 * find_pred_unchecked / rfind_pred_unchecked, search.rs:196-213): a new object holding copies of
 * `count` blocks from `first` (character i of a string = blocks [i*Bc, (i+1)*Bc)), and the
 * concatenation of objects (e.g. one BooleanBlock per character -> the "L" operand of "find_pred" /
 * "rfind_pred").  Device-side copies; the caller destroys the result. */
fhes_ct *fhes_ct_slice(fhes_server_key *sk, fhes_ct *ct, int32_t first, int32_t count);
/* `Clone` of a ciphertext object (the reference clones freely: compare.rs:155-160, mod.rs:250-263, split.rs:322-325).  A
 * device-resident object is copied ON the device (new arena rows, no PCIe traffic: fhes_stats.host_bytes_* do not move); an
 * object that only exists in host memory is copied there.  The caller destroys the result. */
fhes_ct *fhes_ct_clone(const fhes_ct *ct);
fhes_ct *fhes_ct_concat(fhes_server_key *sk, fhes_ct *const *parts, int32_t n_parts);
/* ServerKey blob (/root/reference/src/server_key.rs:11: `#[derive(Serialize, Deserialize)] struct ServerKey`): the
 * standard-domain key material plus the block counts.  fhes_server_key_serialize works on the SERVER's object -- created from a
 * client key, restored from a blob or filled by a key broadcast: the device keeps the standard-domain words next to the
 * Fourier-domain ones -- and needs no client key; serialise -> deserialise -> serialise gives identical bytes.
 * fhes_client_key_serialize_server_key makes the same blob on a machine WITHOUT a device (the client that generated the key;
 * round 1-4's fhes_server_key_serialize).  Deserialising uploads to `device` (Fourier transform on the device). */
int64_t fhes_server_key_serialize(const fhes_server_key *sk, uint8_t *buf, int64_t buf_len);
int64_t fhes_client_key_serialize_server_key(const fhes_client_key *ck, uint8_t *buf, int64_t buf_len);
fhes_server_key *fhes_server_key_deserialize(const uint8_t *buf, int64_t len, int32_t device);

/* Generic operation dispatch: op names and operand order as in fhes_op_signature.  `args` are
 * the ciphertext operands (S, U and the three parts of each V, in signature order), `iargs` the
 * clear integers (I).
 * Results are returned in `outs` (caller destroys), one object per part of
 * fhes_program_out_parts: FheOption<T> -> {is_some, val}; split family -> {is_some[n],
 * start[n][Bu], end[n][Bu]} over `self`; split_once -> {is_some, prefix, suffix}. */
int fhes_string_op(fhes_server_key *sk, const char *op, fhes_ct *const *args, int32_t n_args,
                   const int64_t *iargs, int32_t n_iargs, fhes_ct **outs, int32_t max_outs,
                   int32_t *n_outs);

/* ============================ multi-GPU: one process per GPU ===============================
 * SURVEY.md 8(e) / 2.4 C1-C2.  The reference has no distributed code (rayon only); what shards is the
 * position axis of `find_all` (/root/reference/src/ciphertext/search.rs:33-41): rank g evaluates
 * substr_eq for its contiguous block of start positions, ONE all-gather of the per-position
 * BooleanBlocks follows, and the reduction (`any` / first-match scan) runs replicated -- it is
 * deterministic, so all ranks end with identical ciphertexts.  Keys and inputs are replicated with a
 * broadcast.  Transports:
 *   RCCL  ncclAllGather / ncclBroadcast directly on arena rows and key buffers (device pointers, the
 *         context's stream; librccl.so.1 is resolved at run time, so hosts without it still load this
 *         library); xGMI between the GPUs of a node;
 *   host  the same calls staged through host buffers and carried by two caller-supplied callbacks
 *         (MPI, gloo, a socket...): the CPU-only tests run it over gloo. */
typedef struct fhes_comm fhes_comm;
#define FHES_COMM_ID_BYTES 128
/* rank 0 creates the id (ncclGetUniqueId); the host runtime hands the 128 bytes to every rank */
int fhes_comm_unique_id(uint8_t id[FHES_COMM_ID_BYTES]);
/* collective over all ranks: ncclCommInitRank on the device of `ctx` */
fhes_comm *fhes_comm_create_rccl(fhes_ctx *ctx, int32_t rank, int32_t world, const uint8_t id[FHES_COMM_ID_BYTES]);
/* all-gather: every rank contributes `bytes_per_rank` bytes, recv gets world * bytes_per_rank in rank
 * order; broadcast: `bytes` bytes from `root`.  Return 0 on success. */
typedef int (*fhes_allgather_fn)(void *user, const void *send, void *recv, int64_t bytes_per_rank);
typedef int (*fhes_bcast_fn)(void *user, void *buf, int64_t bytes, int32_t root);
/* `ctx` may be NULL for a communicator that only moves host buffers (fhes_comm_all_gather_host) */
fhes_comm *fhes_comm_create_host(fhes_ctx *ctx, int32_t rank, int32_t world, fhes_allgather_fn all_gather,
                                 fhes_bcast_fn broadcast, void *user);
void fhes_comm_destroy(fhes_comm *comm);
/* Diagnostics of the transport.  fhes_comm_self_test runs one all-gather, one all-reduce and one broadcast of a few words
 * on device memory of THIS library's HIP runtime (collective: every rank calls it) -- with RCCL also at world == 1, where the
 * data path needs no collective -- and checks the words.  The RCCL binding is resolved against the HIP runtime this library
 * is bound to (a process may hold a second one, e.g. the runtime a PyTorch wheel bundles): fhes_comm_create_rccl fails with
 * FHES_ERR_STATE and a message naming both files instead of RCCL's "unhandled cuda error" when no matching librccl exists. */
int fhes_comm_self_test(fhes_comm *comm);
const char *fhes_rccl_library_path(void);
const char *fhes_rccl_hip_runtime_path(void);
int32_t fhes_comm_rank(const fhes_comm *comm);
int32_t fhes_comm_world(const fhes_comm *comm);
/* 1 = RCCL, 0 = host callbacks */
int32_t fhes_comm_is_rccl(const fhes_comm *comm);
/* Block partition of n positions: rank r owns [lo, hi) = [min(r*c, n), min((r+1)*c, n)), c = ceil(n / world)
 * (equal-sized segments, so the gathered rows are contiguous in position order). */
int fhes_shard_range(int64_t n, int32_t world, int32_t rank, int64_t *lo, int64_t *hi);
/* Host-buffer all-gather through the communicator's transport (RCCL: staged through the device). */
int fhes_comm_all_gather_host(fhes_comm *comm, const void *send, void *recv, int64_t bytes_per_rank);
/* In-place all-gather of arena rows: rank r contributes the `count_per_rank` ciphertexts at handles
 * [first + r*count_per_rank, ...); afterwards every rank holds all world*count_per_rank rows. */
int fhes_comm_all_gather_cts(fhes_comm *comm, int32_t first, int32_t count_per_rank);
/* Broadcast of arena rows [first, first+count) from `root`. */
int fhes_comm_broadcast_cts(fhes_comm *comm, int32_t first, int32_t count, int32_t root);
/* Broadcast of the device-resident server keys (keyswitch key + bias, Fourier BSK in both layouts)
 * from `root`; the other ranks need not have called fhes_upload_keys. */
int fhes_comm_broadcast_keys(fhes_comm *comm, int32_t root);
/* Position-sharded operations on ONE string.  a / p are handles of the string (a_slots chars incl. terminator /
 * padding) and the pattern, identical on every rank (fhes_comm_broadcast_cts).  Collective: every rank makes the
 * same call; the results (freshly allocated, one contiguous range) are identical ciphertexts on every rank.
 *   "find" (search.rs:23-28) -> out[0..1+Bu) = is_some, index; "rfind_last" (the reverse scan of search.rs:155-159):
 *       rank g evaluates substr_eq for its block of start positions (search.rs:33-41), ONE all-gather of the
 *       per-position BooleanBlocks, the first-match scan replicated (bit-identical to the unsharded program);
 *   "contains" (search.rs:17-20) -> 1 block: `any` of the local block, ONE all-reduce (ncclUint64 / ncclSum: LWE
 *       addition is wrapping u64 addition) of that single boolean, final lookup [sum > 0]  (world <= 15; else as find);
 *   "split" / "rsplit" (split.rs:238-333) -> out[0..3) = first handles of is_some[n], start[n][Bu], end[n][Bu],
 *       n = a_slots + 1: found[] as above, the non-overlapping matches replicated, then the O(n^2) grid
 *       `element_at_bool(matches, i - len(p))` (mod.rs:271-287 per index at split.rs:283-285) partitioned by OUTPUT
 *       index i with one all-gather of the booleans, the carried next_match chain replicated.
 * An "<op>_logdepth" name selects the parallel-prefix forms of the replicated scans; for split / rsplit everything behind
 * find_all then runs as ONE replicated program (the barrel shifter that replaces the per-index gathers is cheaper than the
 * grid divided by the ranks).  A rank that fails locally
 * still enters every collective, and a closing status exchange makes EVERY rank return an error (nobody hangs,
 * nobody keeps a void result). */
int fhes_sharded_op(fhes_comm *comm, const char *op, int32_t a_first, int32_t a_slots, int32_t p_first,
                    int32_t p_slots, int32_t num_blocks_usize, int32_t *out_handles, int32_t max_out,
                    int32_t *n_out);
/* The same for EVERY operation whose circuit starts with find_all(self, pattern): rfind, split / rsplit / split_inclusive /
 * split_terminator / rsplit_terminator / splitn / rsplitn, split_once / rsplit_once, strip_suffix, replace / replacen (and find /
 * contains).  found[] is computed per rank and all-gathered, the rest of the operation ("<op>.rest": the same circuit with its
 * find_all handed in) runs replicated.  Operands: the operation's ciphertext operands in the order of fhes_op_signature (S / U / C /
 * B letters), each as first handle + block count; `iargs` its I operands.  Result: one contiguous handle range starting at
 * *out_first, split into `parts` as fhes_program_out_parts lays the operation's results out. */
int fhes_sharded_op_ex(fhes_comm *comm, const char *op, const int32_t *operand_first, const int32_t *operand_blocks, int32_t n_operands,
                       const int64_t *iargs, int32_t n_iargs, int32_t num_blocks_usize, int32_t *out_first, int32_t *parts,
                       int32_t max_parts, int32_t *n_parts);
/* the search subset of fhes_sharded_op under its round-1 name */
int fhes_sharded_search(fhes_comm *comm, const char *op, int32_t a_first, int32_t a_slots, int32_t p_first,
                        int32_t p_slots, int32_t num_blocks_usize, int32_t *out_handles, int32_t max_out,
                        int32_t *n_out);
/* Executor hook: the SAME planner (fhes_sharded_op) over ciphertext rows that live in host memory and programs run
 * by the caller -- a CPU implementation, another device, a test double.  tests/test_parallel_gloo.py runs the
 * shipped planner this way with the CPU oracle as the executor, two ranks over gloo.  Handles are the executor's
 * own row numbers; rows are `row_words` u64 apart and contiguous in handle order. */
typedef struct fhes_executor {
  void *user;
  int64_t row_words;                                             /* >= fhes_ct_words(params) */
  int32_t (*alloc)(void *user, int32_t count);                   /* first handle of `count` fresh rows, < 0 on failure */
  int (*release)(void *user, int32_t first, int32_t count);
  uint64_t *(*rows)(void *user, int32_t first);                  /* host address of row `first` */
  int (*run)(void *user, const fhes_program *prog, const int32_t *in_handles, int32_t *out_handles); /* 0 = ok; allocates the outputs contiguously */
} fhes_executor;
fhes_comm *fhes_comm_create_host_exec(const fhes_params *p, int32_t rank, int32_t world, fhes_allgather_fn all_gather,
                                      fhes_bcast_fn broadcast, void *user, const fhes_executor *exec);
/* Level 2: with a communicator attached, `contains`, `find`, `split` and `rsplit` of fhes_string_op run
 * position-sharded over its ranks (every rank makes the same call with the same operands).  NULL detaches; detach
 * before the communicator is destroyed. */
int fhes_server_key_set_comm(fhes_server_key *sk, fhes_comm *comm);

#ifdef __cplusplus
}
#endif
#endif
