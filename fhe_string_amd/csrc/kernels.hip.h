// kernels.hip.h -- gfx950 device code for the PBS hot path.
//
// Replaces, for the reference's `k.k.*_parallelized` call sites
// (/root/reference/src/ciphertext/compare.rs:130-132, mod.rs:345-348, logic.rs:15,24,49,63),
// what tfhe-rs executes per shortint lookup: keyswitch -> modulus switch -> blind
// rotate (n CMUX with the external product in the f64 Fourier domain) -> sample
// extract.  The arithmetic follows the canonical op sequence of DESIGN.md section 4
// (same as oracle/tfhe_oracle.c), so outputs are bit-identical to the oracle:
//   * forward transform: radix-2 Cooley-Tukey stages with the negacyclic twist merged into the
//     twiddles (natural order in, bit-reversed out; fused butterfly s = u + t v, d = 2u - s);
//     inverse: radix-2 DIT (bit-reversed in, natural out) + untwist; twiddles from tables;
//   * -ffp-contract=off: every FMA is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fhes {
namespace dev {

// ------------------------------------------------------------------------------------------
// canonical scalar helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 w) {
  double2 r;
  r.x = __builtin_fma(a.x, w.x, -(a.y * w.y));
  r.y = __builtin_fma(a.x, w.y, a.y * w.x);
  return r;
}
__device__ __forceinline__ double2 cmul_conj(double2 a, double2 w) {
  double2 r;
  r.x = __builtin_fma(a.x, w.x, a.y * w.y);
  r.y = __builtin_fma(a.y, w.x, -(a.x * w.y));
  return r;
}
// exact rint(y) mod 2^64 for |y| < 2^100
__device__ __forceinline__ uint64_t torus_from_f64(double y) {
  const double t = y * 0x1p-64;
  const double kq = __builtin_floor(t + 0.5);
  const double z = __builtin_fma(kq, -0x1p64, y);
  const double r = __builtin_rint(z);
  const double h = __builtin_floor(r * 0x1p-32);
  const double l = __builtin_fma(h, -0x1p32, r);
  const int hi = (int)h;
  const unsigned lo = (unsigned)l;
  return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)lo;
}
// Same value (rint(y) mod 2^64, exact for |y| < 2^115) with 7 f64 adds and 2 integer ops:
// magic-constant splitting.  y + 1.5*2^116 rounds y to a multiple of 2^64; z = y - that is in
// [-2^63, 2^63]; z + 1.5*2^84 leaves rint(z/2^32) mod 2^32 in the low mantissa dword;
// rem + 1.5*2^52 leaves rint(rem) as a 52-bit two's complement value offset by 2^51.
__device__ __forceinline__ uint64_t torus_from_f64_fast(double y) {
  const double C0 = 0x1.8p116, C1 = 0x1.8p84, C2 = 0x1.8p52;
  const double h0 = (y + C0) - C0;
  const double z = y - h0;
  const double t1 = z + C1;
  const double rem = z - (t1 - C1);
  const double t2 = rem + C2;
  const uint32_t hi32 = (uint32_t)__double2loint(t1);
  const uint32_t lo = (uint32_t)__double2loint(t2);
  const uint32_t mhi = ((uint32_t)__double2hiint(t2) & 0xFFFFFu) - 0x80000u;  // 0 or 0xFFFFFFFF
  return ((uint64_t)(hi32 + mhi) << 32) | (uint64_t)lo;
}
// correctly rounded int64 -> f64 (single rounding)
__device__ __forceinline__ double i64_to_f64(uint64_t v) {
  const int hi = (int)(v >> 32);
  const unsigned lo = (unsigned)v;
  return __builtin_fma((double)hi, 0x1p32, (double)lo);
}
// one-level closest-representable signed digit (base 2^BL) as f64:
//   v = ((x >> (63-BL)) + 1) >> 1  (mod 2^BL, read as a signed BL-bit number)
// Only the high dword matters (63-BL >= 32): adding the rounding bit 2^(31-BL) to it and shifting
// arithmetically by 32-BL gives exactly that signed value (the 32-bit wrap is the mod 2^BL).
template <int BL>
__device__ __forceinline__ double digit1_f64(uint64_t x) {
  static_assert(BL >= 1 && BL <= 31, "single-level digit must fit the high dword");
  const uint32_t hi = (uint32_t)(x >> 32);
  const int d = (int)(hi + (1u << (31 - BL))) >> (32 - BL);
  return (double)d;
}
// (X^a * p)[j] for a in [1, 2N), polynomial in LDS/global
template <int N>
__device__ __forceinline__ uint64_t rot_read(const uint64_t *p, int a, int j) {
  const int s = (j - a) & (2 * N - 1);
  const uint64_t v = p[s & (N - 1)];
  return (s >= N) ? (uint64_t)0 - v : v;
}

// ------------------------------------------------------------------------------------------
// FFT passes.  M = 2^LOGM complex points handled by TP = M/8 threads, 8 points per
// thread.  A pass owns index bits [LO, LO+NB): it performs the radix-2 stages of those
// bits in registers; passes exchange data through LDS.
// ------------------------------------------------------------------------------------------
template <int LOGM, int LO, int NB>
__device__ __forceinline__ int fft_point(int q, int e) {
  constexpr int LOGTP = LOGM - 3;
  const int f = e & ((1 << NB) - 1), g = e >> NB;
  const int rest = (g << LOGTP) | q;
  return ((rest >> LO) << (LO + NB)) | (f << LO) | (rest & ((1 << LO) - 1));
}

// inverse (DIT) butterfly, fused form: s = u + v*conj(w) in 4 FMAs, d = 2u - s in 2
__device__ __forceinline__ void inv_butterfly(double2 &a, double2 &b, double2 w) {
  const double2 u = a, v = b;
  double sre = __builtin_fma(v.x, w.x, u.x);
  sre = __builtin_fma(v.y, w.y, sre);
  double sim = __builtin_fma(v.y, w.x, u.y);
  sim = __builtin_fma(-v.x, w.y, sim);
  a = make_double2(sre, sim);
  b = make_double2(__builtin_fma(2.0, u.x, -sre), __builtin_fma(2.0, u.y, -sim));
}

// forward (Cooley-Tukey, twist merged into the twiddles): s = u + t v in 4 FMAs, d = 2u - s in 2
__device__ __forceinline__ void fwd_butterfly(double2 &a, double2 &b, double2 t) {
  const double2 u = a, v = b;
  double sre = __builtin_fma(t.x, v.x, u.x);
  sre = __builtin_fma(-t.y, v.y, sre);
  double sim = __builtin_fma(t.x, v.y, u.y);
  sim = __builtin_fma(t.y, v.x, sim);
  a = make_double2(sre, sim);
  b = make_double2(__builtin_fma(2.0, u.x, -sre), __builtin_fma(2.0, u.y, -sim));
}
// last forward stage (adjacent slots): the even slot gets s = u + t v, the odd slot HOLDS h = v + t' u with
// t' = (-t.x, t.y) = -conj(t), the true value u - t v divided by -t.  A lane pair computes both as
// own + c * partner (fft_fwd_stage0_dpp); the Fourier key carries t^2 on its odd slots (bsk_to_fourier_kernel).
__device__ __forceinline__ void fwd_butterfly_last(double2 &a, double2 &b, double2 t) {
  const double2 u = a, v = b;
  double sre = __builtin_fma(t.x, v.x, u.x);
  sre = __builtin_fma(-t.y, v.y, sre);
  double sim = __builtin_fma(t.x, v.y, u.y);
  sim = __builtin_fma(t.y, v.x, sim);
  double hre = __builtin_fma(-t.x, u.x, v.x);
  hre = __builtin_fma(-t.y, u.y, hre);
  double him = __builtin_fma(-t.x, u.y, v.y);
  him = __builtin_fma(t.y, u.x, him);
  a = make_double2(sre, sim);
  b = make_double2(hre, him);
}

// FWD: tw = psi (psi[2^l + b]: block b of the stage with 2^l blocks, l = LOGM-1-beta);
// inverse: tw = exp(-2 pi i t / M) table
template <int LOGM, bool FWD, int LO, int NB>
__device__ __forceinline__ void fft_butterflies(double2 (&x)[8], int q,
                                                const double2 *__restrict__ tw) {
#pragma unroll
  for (int st = 0; st < NB; ++st) {
    const int fb = FWD ? (NB - 1 - st) : st;
    const int beta = LO + fb;           // absolute index bit of this stage
    const int sh = LOGM - 1 - beta;     // inverse: twiddle stride = 2^sh; forward: the stage has 2^sh blocks
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e & (1 << fb)) continue;
      const int e1 = e | (1 << fb);
      const int p = fft_point<LOGM, LO, NB>(q, e);
      if (FWD) {
        const double2 t = tw[(1 << sh) + (p >> (beta + 1))];
        if (beta == 0) fwd_butterfly_last(x[e], x[e1], t);
        else fwd_butterfly(x[e], x[e1], t);
      } else if (beta == 0) {  // twiddle index 0: exactly 1
        const double2 u = x[e], v = x[e1];
        x[e] = make_double2(u.x + v.x, u.y + v.y);
        x[e1] = make_double2(u.x - v.x, u.y - v.y);
      } else {
        inv_butterfly(x[e], x[e1], tw[(p & ((1 << beta) - 1)) << sh]);
      }
    }
  }
}

template <int LOGM>
struct FftPlan {
  static constexpr int NB_LAST = (LOGM % 3 == 0) ? 3 : (LOGM % 3);  // pass that owns bit 0
  static constexpr int LO_FIRST = LOGM - 3;                          // first forward pass
};

// forward: x holds the points of pass (LO = HI-NB); ends with the spectrum in `buf`
// (canonical bit-reversed order, position == in-place index) and a barrier.
// The twiddles of a pass depend on the thread only: left alone, the compiler hoists the loads of EVERY pass of both
// transforms out of the blind-rotate loop and keeps them live across it (28 registers per pass: the source of the generic
// kernel's spills at N >= 2048).  `tw_here` makes the table pointer opaque per pass, so a pass's twiddles are fetched where
// the pass runs.  (Shapes below N = 2048 have the registers to keep them and run 5-10 % faster hoisted: fft_forward / fft_inverse
// apply it from LOGM = 10 on.)
// (An opaque OFFSET, not an opaque pointer: a pointer that went through the asm statement loses its address space and every
// load from it becomes a flat_load, which also counts against the LDS wait counter.)
__device__ __forceinline__ const double2 *tw_here(const double2 *tw) {
  int z = 0;
  asm volatile("" : "+s"(z));
  return tw + z;
}

template <int LOGM, int HI>
__device__ __forceinline__ void fft_forward(double2 (&x)[8], int q, double2 *buf,
                                            const double2 *__restrict__ tw) {
  constexpr int NB = HI >= 3 ? 3 : HI;
  constexpr int LO = HI - NB;
  fft_butterflies<LOGM, true, LO, NB>(x, q, LOGM >= 10 ? tw_here(tw) : tw);
#pragma unroll
  for (int e = 0; e < 8; ++e) buf[fft_point<LOGM, LO, NB>(q, e)] = x[e];
  __syncthreads();
  if constexpr (LO > 0) {
    constexpr int NB2 = LO >= 3 ? 3 : LO;
    constexpr int LO2 = LO - NB2;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = buf[fft_point<LOGM, LO2, NB2>(q, e)];
    fft_forward<LOGM, LO>(x, q, buf, tw);
  }
}

// inverse: x holds the spectrum points of pass (LO, NB); ends with natural-order data of
// the pass (LOGM-3, 3) in registers (no trailing barrier).
template <int LOGM, int LO, int NB>
__device__ __forceinline__ void fft_inverse(double2 (&x)[8], int q, double2 *buf,
                                            const double2 *__restrict__ tw) {
  fft_butterflies<LOGM, false, LO, NB>(x, q, LOGM >= 10 ? tw_here(tw) : tw);
  if constexpr (LO + NB < LOGM) {
#pragma unroll
    for (int e = 0; e < 8; ++e) buf[fft_point<LOGM, LO, NB>(q, e)] = x[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = buf[fft_point<LOGM, LO + NB, 3>(q, e)];
    fft_inverse<LOGM, LO + NB, 3>(x, q, buf, tw);
  }
}

// ------------------------------------------------------------------------------------------
// K7: standard-domain BSK polynomial -> canonical Fourier spectrum
// grid = number of polynomials, block = M/8 threads, dyn LDS = M * 16 bytes
// ------------------------------------------------------------------------------------------
template <int LOGN>
__global__ void bsk_to_fourier_kernel(const uint64_t *__restrict__ bsk, double2 *__restrict__ out,
                                      const double2 *__restrict__ psi,
                                      const double2 *__restrict__ odd_fix) {
  constexpr int N = 1 << LOGN, M = N / 2, LOGM = LOGN - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double2 *buf = reinterpret_cast<double2 *>(smem_raw);
  const int q = threadIdx.x;
  const uint64_t *poly = bsk + (size_t)blockIdx.x * N;
  double2 x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int p = fft_point<LOGM, LOGM - 3, 3>(q, e);
    x[e] = make_double2(i64_to_f64(poly[p]), i64_to_f64(poly[p + M]));
  }
  fft_forward<LOGM, LOGM>(x, q, buf, psi);
  // odd slots: times the square of the last-stage twiddle (see fwd_butterfly_last), so that the product
  // of this spectrum with a digit spectrum in the same internal form is the true product
  double2 *dst = out + (size_t)blockIdx.x * M;
  for (int p = q; p < M; p += blockDim.x) dst[p] = (p & 1) ? cmul(buf[p], odd_fix[p]) : buf[p];
}

// ------------------------------------------------------------------------------------------
// node descriptors shared by the batched kernels
// ------------------------------------------------------------------------------------------
struct NodeDesc {
  const int32_t *term_ptr;   // [nodes+1]
  const int32_t *term_idx;   // slot (or arena handle if slot_map == nullptr)
  const int64_t *term_coef;
  const uint64_t *consts;    // raw torus word added to the body
  const int32_t *lut_ids;
  const int32_t *out_idx;    // slot / handle
  const int32_t *slot_map;   // optional [n_inst][n_slots]: slot -> arena handle
  int32_t first;             // first node (of the program) of this level
  int32_t count;             // nodes of this level, per instance
  int32_t n_slots;           // slots per instance (stride of slot_map)
  int32_t v0;                // first virtual index of this launch
  int32_t v_count;           // virtual nodes in this launch
};
// A launch covers virtual indices v in [v0, v0 + v_count) of the space
// (instance, node-in-level):  instance = v / count, node = first + v % count.
// Batched execution of one program on many independent inputs is the same launch with
// n_inst * count virtual nodes.

__device__ __forceinline__ int32_t resolve(const NodeDesc &d, int32_t inst, int32_t s) {
  return d.slot_map ? d.slot_map[(size_t)inst * d.n_slots + s] : s;
}

// ------------------------------------------------------------------------------------------
// K1: batched keyswitch (big LWE -> small LWE) of linear combinations.
// grid = (tiles of TB nodes, splits of the mask range); block = 256 threads.
// Each thread accumulates JPT output words for TB ciphertexts; partial sums over the
// mask split are combined with u64 atomics (integer adds commute: deterministic).
// ks_out [count][ks_stride] must be zero on entry.
// ------------------------------------------------------------------------------------------
constexpr int KS_TB = 8;     // ciphertexts per block (narrow launches)
constexpr int KS_TB_WIDE = 16;  // developer knob FHES_KS_TB=16 (measured: no faster, see the loop below)
constexpr int KS_IC = 64;    // mask coefficients staged per chunk
constexpr int KS_MAXLV = 8;  // ks_level limit

template <int JPT, int TB>
__global__ __launch_bounds__(256) void keyswitch_kernel(NodeDesc d, const uint64_t *__restrict__ arena,
                                                        int ct_stride, int big,
                                                        const uint64_t *__restrict__ ksk,
                                                        const uint64_t *__restrict__ ksk_bias,
                                                        int ks_level, int ks_base_log, int ks_stride,
                                                        int n_small, int coeffs_per_split,
                                                        uint64_t *__restrict__ ks_out) {
  // digits are staged biased: u = d + B/2 in [0, B), so that the product u * K needs no sign
  // handling; sum_{i,l} (B/2) * K[i][l][j] is the precomputed `ksk_bias[j]` (added once).
  __shared__ __attribute__((aligned(8))) uint8_t dig[KS_IC][KS_MAXLV][TB];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int i_begin = blockIdx.y * coeffs_per_split;
  const int i_end = i_begin + coeffs_per_split;
  const int vbase = tile * TB;  // virtual index (relative to v0) of this block's first ciphertext
  const int nodes_here = min(TB, d.v_count - vbase);
  const int total = ks_level * ks_base_log;
  const int B = 1 << ks_base_log;

  // sum u*K split into a 64-bit sum of u*K_lo and a 32-bit sum of u*K_hi (mod 2^32)
  uint64_t acc_lo[TB][JPT];
  uint32_t acc_hi[TB][JPT];
#pragma unroll
  for (int b = 0; b < TB; ++b)
#pragma unroll
    for (int q = 0; q < JPT; ++q) { acc_lo[b][q] = 0; acc_hi[b][q] = 0; }

  for (int chunk = i_begin; chunk < i_end; chunk += KS_IC) {
    const int lim = min(KS_IC, i_end - chunk);  // small launches split finer than one chunk
    // stage the digits of TB x KS_IC mask coefficients
    for (int idx = tid; idx < TB * KS_IC; idx += 256) {
      const int b = idx / KS_IC, ii = idx % KS_IC;
      if (ii >= lim) continue;
      uint64_t x = 0;
      if (b < nodes_here) {
        const int v = d.v0 + vbase + b;
        const int inst = v / d.count, node = d.first + v % d.count;
        for (int t = d.term_ptr[node]; t < d.term_ptr[node + 1]; ++t) {
          const uint64_t *src = arena + (size_t)resolve(d, inst, d.term_idx[t]) * ct_stride;
          x += (uint64_t)d.term_coef[t] * src[chunk + ii];
        }
      }
      uint32_t v = (uint32_t)((((x >> (64 - total - 1)) + 1) >> 1) & ((1ull << total) - 1));
      for (int l = ks_level - 1; l >= 0; --l) {
        int dd = (int)(v & (uint32_t)(B - 1));
        v >>= ks_base_log;
        if (dd >= B / 2) { dd -= B; v += 1; }
        dig[ii][l][b] = (uint8_t)(dd + B / 2);
      }
    }
    __syncthreads();
    // (Round 2, measured per 16 384 ciphertexts: this loop 11.7 ms; a second 64-bit accumulator instead of the
    // 32-bit one 11.6 ms -- the compiler already emits v_mad_u64_u32 for both; 16 ciphertexts per block 11.9 ms;
    // key words of the next row requested one row ahead 13.5 ms.  The loop runs at 67 % of the multiplier
    // rate: 48 multiply-accumulates per row and wave against the latency of the row's three key loads, four
    // waves per SIMD.)
    for (int ii = 0; ii < lim; ++ii) {
      const uint64_t *krow = ksk + ((size_t)(chunk + ii) * ks_level) * ks_stride;
      for (int l = 0; l < ks_level; ++l) {
        uint64_t kv[JPT];
#pragma unroll
        for (int q = 0; q < JPT; ++q) kv[q] = krow[(size_t)l * ks_stride + tid + 256 * q];
        uint64_t packed[TB / 8];
#pragma unroll
        for (int w8 = 0; w8 < TB / 8; ++w8) packed[w8] = *reinterpret_cast<const uint64_t *>(&dig[ii][l][8 * w8]);
#pragma unroll
        for (int b = 0; b < TB; ++b) {
          const uint32_t u = (uint32_t)(packed[b / 8] >> (8 * (b % 8))) & 0xFFu;
#pragma unroll
          for (int q = 0; q < JPT; ++q) {
            acc_lo[b][q] += (uint64_t)u * (uint32_t)kv[q];          // v_mad_u64_u32
            acc_hi[b][q] += u * (uint32_t)(kv[q] >> 32);            // a second v_mad_u64_u32 (low dword kept)
          }
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int b = 0; b < TB; ++b) {
    if (b >= nodes_here) continue;  // (no `break`: the loop must unroll fully, acc[][] lives in registers)
    const int v = d.v0 + vbase + b;
    const int inst = v / d.count, node = d.first + v % d.count;
    uint64_t *dst = ks_out + (size_t)(vbase + b) * ks_stride;
#pragma unroll
    for (int q = 0; q < JPT; ++q) {
      const int j = tid + 256 * q;
      uint64_t val = (uint64_t)0 - (acc_lo[b][q] + ((uint64_t)acc_hi[b][q] << 32));  // - sum u*K
      if (blockIdx.y == 0 && j <= n_small) {
        val += ksk_bias[j];  // + (B/2) * sum K
        if (j == n_small) {  // body of the input combination
          uint64_t body = d.consts[node];
          for (int t = d.term_ptr[node]; t < d.term_ptr[node + 1]; ++t) {
            const uint64_t *src = arena + (size_t)resolve(d, inst, d.term_idx[t]) * ct_stride;
            body += (uint64_t)d.term_coef[t] * src[big];
          }
          val += body;
        }
      }
      if (j <= n_small && val != 0) atomicAdd(reinterpret_cast<unsigned long long *>(dst + j), (unsigned long long)val);
    }
  }
}

// ------------------------------------------------------------------------------------------
// K1b: the keyswitch of a WIDE launch as an integer matrix product on the matrix cores.
//     out[c][j] = body[c] * [j == n]  -  sum_k dig[c][k] * K[k][j]          (mod 2^64)
// with k = l * big + i over the kN * ks_level signed digits of ciphertext c (|dig| <= 2^(base_log-1) <= 64)
// is a (ciphertexts x digits) by (digits x key words) product.  A 64-bit key word is split into 8 signed
// bytes  K = sum_b s_b * 256^b  (s_b in [-128, 127], borrow carried upwards; the carry out of byte 7 vanishes
// mod 2^64), so the product becomes eight i8 x i8 -> i32 products, exact (|sum| <= 16384 * 64 * 128 < 2^31),
// recombined with shifts in the epilogue.  Same integers as keyswitch_kernel, hence the same bits.
// Against the VALU kernel (11.7 ms per 16 384 ciphertexts at the reference's parameters, 67 % of the
// 64-bit multiplier rate) the matrix cores retire the 1.0e15 byte products of that launch in about a tenth.
//   keyswitch_digits_kernel: one workgroup per ciphertext; linear combination of the inputs, rounding,
//     balanced digits, written as i8 in MFMA A-fragment order  dig[row / 16][k / 64][lane][16]  with
//     lane = row % 16 + 16 * ((k % 64) / 16), byte = k % 16; the body of the combination goes to body[row].
//   keyswitch_mfma_kernel: v_mfma_i32_16x16x64_i8; a wave owns 64 ciphertexts x 16 key words (= 8 byte
//     planes = 8 column tiles, so that a lane holds all 8 byte planes of ITS word: recombination needs no
//     lane traffic), a workgroup 2 x 2 waves.  Both operands stream from L2 in fragment order (the key is
//     packed once at upload: kpk[k / 64][word / 16][byte plane][lane][16]); no LDS.
// A and B use the same (lane group, byte) -> k assignment, so the product sums every k exactly once
// whatever order the instruction walks them in.
// ------------------------------------------------------------------------------------------
typedef int ksm_v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(128) void keyswitch_digits_kernel(NodeDesc d, const uint64_t *__restrict__ arena,
                                                               int ct_stride, int big, int ks_level,
                                                               int ks_base_log, int ksteps,
                                                               int8_t *__restrict__ dig, uint64_t *__restrict__ body) {
  const int row = blockIdx.x;
  const int v = d.v0 + row;
  const int inst = v / d.count, node = d.first + v % d.count;
  const int t0 = d.term_ptr[node], t1 = d.term_ptr[node + 1];
  const int total = ks_level * ks_base_log;
  const int B = 1 << ks_base_log;
  for (int i0 = 16 * (int)threadIdx.x; i0 < big; i0 += 16 * (int)blockDim.x) {
    uint64_t x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = 0;
    for (int t = t0; t < t1; ++t) {
      const uint64_t *src = arena + (size_t)resolve(d, inst, d.term_idx[t]) * ct_stride + i0;
      const uint64_t c = (uint64_t)d.term_coef[t];
#pragma unroll
      for (int u = 0; u < 16; ++u) x[u] += c * src[u];
    }
    uint32_t pk[KS_MAXLV][4];
#pragma unroll
    for (int l = 0; l < KS_MAXLV; ++l) pk[l][0] = pk[l][1] = pk[l][2] = pk[l][3] = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      uint32_t w = (uint32_t)((((x[u] >> (64 - total - 1)) + 1) >> 1) & ((1ull << total) - 1));
#pragma unroll
      for (int l = KS_MAXLV - 1; l >= 0; --l) {
        if (l >= ks_level) continue;
        int dd = (int)(w & (uint32_t)(B - 1));
        w >>= ks_base_log;
        if (dd >= B / 2) { dd -= B; w += 1; }
        pk[l][u >> 2] |= ((uint32_t)dd & 0xFFu) << (8 * (u & 3));
      }
    }
#pragma unroll
    for (int l = 0; l < KS_MAXLV; ++l) {
      if (l >= ks_level) continue;
      const int k = l * big + i0;
      const int lane = (row & 15) + 16 * ((k & 63) >> 4);
      uint4 *dst = reinterpret_cast<uint4 *>(dig + ((((size_t)(row >> 4) * ksteps + (k >> 6)) * 64 + lane) << 4));
      *dst = make_uint4(pk[l][0], pk[l][1], pk[l][2], pk[l][3]);
    }
  }
  if (threadIdx.x == 0) {
    uint64_t b = d.consts[node];
    for (int t = t0; t < t1; ++t)
      b += (uint64_t)d.term_coef[t] * arena[(size_t)resolve(d, inst, d.term_idx[t]) * ct_stride + big];
    body[row] = b;
  }
}

__global__ __launch_bounds__(256, 2) void keyswitch_mfma_kernel(const int8_t *__restrict__ dig,
                                                                const int8_t *__restrict__ kpk,
                                                                const uint64_t *__restrict__ body,
                                                                uint64_t *__restrict__ ks_out, int ks_stride,
                                                                int ksteps, int n_wb, int n_small, int count,
                                                                int steps_per_split, int use_atomic) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowtile0 = blockIdx.x * 8 + (wave & 1) * 4;  // 16-row tiles of this wave
  const int wb = blockIdx.y * 2 + (wave >> 1);           // block of 16 key words
  if (wb >= n_wb) return;                                // (no barriers in this kernel)
  const int s0 = blockIdx.z * steps_per_split;
  const int s1 = min(ksteps, s0 + steps_per_split);
  const ksm_v4i *A = reinterpret_cast<const ksm_v4i *>(dig);
  const ksm_v4i *Bm = reinterpret_cast<const ksm_v4i *>(kpk);
  ksm_v4i acc[4][8];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[rt][t] = ksm_v4i{0, 0, 0, 0};
  ksm_v4i a[4], b[8];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) a[rt] = A[((size_t)(rowtile0 + rt) * ksteps + s0) * 64 + lane];
#pragma unroll
  for (int t = 0; t < 8; ++t) b[t] = Bm[(((size_t)s0 * n_wb + wb) * 8 + t) * 64 + lane];
  for (int s = s0; s < s1; ++s) {
    ksm_v4i an[4], bn[8];
    const int sn = s + 1 < s1 ? s + 1 : s;  // the last step re-reads itself (no branch around the loads)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) an[rt] = A[((size_t)(rowtile0 + rt) * ksteps + sn) * 64 + lane];
#pragma unroll
    for (int t = 0; t < 8; ++t) bn[t] = Bm[(((size_t)sn * n_wb + wb) * 8 + t) * 64 + lane];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[rt][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rt], b[t], acc[rt][t], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) a[rt] = an[rt];
#pragma unroll
    for (int t = 0; t < 8; ++t) b[t] = bn[t];
  }
  // C layout: column = lane & 15 (the key word), row = (lane >> 4) * 4 + r (the ciphertext)
  const int word = wb * 16 + (lane & 15);
  if (word > n_small) return;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (rowtile0 + rt) * 16 + (lane >> 4) * 4 + r;
      if (row >= count) continue;
      uint64_t sum = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) sum += (uint64_t)(int64_t)acc[rt][t][r] << (8 * t);
      uint64_t val = (uint64_t)0 - sum;
      if (word == n_small && blockIdx.z == 0) val += body[row];
      uint64_t *dst = ks_out + (size_t)row * ks_stride + word;
      if (use_atomic) atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)val);
      else *dst = val;
    }
  }
}

// ------------------------------------------------------------------------------------------
// K5: materialise linear combinations.  grid = nodes, block = 256.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_kernel(NodeDesc d, uint64_t *__restrict__ arena,
                                                     int ct_stride, int big) {
  const int v = d.v0 + blockIdx.x;
  const int inst = v / d.count, node = d.first + v % d.count;
  uint64_t *dst = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
  const int t0 = d.term_ptr[node], t1 = d.term_ptr[node + 1];
  for (int j = threadIdx.x; j <= big; j += blockDim.x) {
    uint64_t x = (j == big) ? d.consts[node] : 0;
    for (int t = t0; t < t1; ++t)
      x += (uint64_t)d.term_coef[t] * arena[(size_t)resolve(d, inst, d.term_idx[t]) * ct_stride + j];
    dst[j] = x;
  }
}

// ------------------------------------------------------------------------------------------
// K2+K3 (variant 1): blind rotate + sample extract, one workgroup (2 * M/8 threads) per
// ciphertext, k = 1, pbs_level = 1.  LDS: acc u64[2][N] | fbuf double2[2][M] | a_tilde.
// ------------------------------------------------------------------------------------------
#ifdef FHES_LAB  // the simple reference build (variant 1): lab only since round 5
template <int LOGN, int BL>
__global__ __launch_bounds__(((1 << LOGN) / 8) < 64 ? 64 : ((1 << LOGN) / 8)) void blind_rotate_v1_kernel(NodeDesc d, uint64_t *__restrict__ arena, int ct_stride,
                                       const uint64_t *__restrict__ ks_out, int ks_stride,
                                       int n_small, const double2 *__restrict__ bsk_f,
                                       const uint64_t *__restrict__ lut_polys,
                                       const double2 *__restrict__ twist,
                                       const double2 *__restrict__ tw,
                                       const double2 *__restrict__ psi) {
  constexpr int N = 1 << LOGN, M = N / 2, LOGM = LOGN - 1, TP = M / 8;
  constexpr int NB0 = FftPlan<LOGM>::NB_LAST;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t *acc = reinterpret_cast<uint64_t *>(smem_raw);                     // [2][N]
  double2 *fbuf = reinterpret_cast<double2 *>(smem_raw + (size_t)2 * N * 8);  // [2][M]
  uint32_t *at = reinterpret_cast<uint32_t *>(smem_raw + (size_t)2 * N * 8 + (size_t)2 * M * 16);

  const int tid = threadIdx.x;
  const int r = tid / TP;  // polynomial handled by this half of the workgroup
  const int q = tid % TP;
  const int vidx = d.v0 + blockIdx.x;
  const int inst = vidx / d.count, node = d.first + vidx % d.count;

  // ---- modulus switch to Z_{2N} ----
  const uint64_t *ks = ks_out + (size_t)blockIdx.x * ks_stride;
  for (int j = tid; j <= n_small; j += blockDim.x)
    at[j] = (uint32_t)((((ks[j] >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
  __syncthreads();

  // ---- acc = (0, X^{-b~} * LUT) ----
  {
    const uint64_t *lut = lut_polys + (size_t)d.lut_ids[node] * N;
    const int t = (2 * N - (int)at[n_small]) & (2 * N - 1);
    for (int j = tid; j < N; j += blockDim.x) {
      acc[j] = 0;
      acc[N + j] = (t == 0) ? lut[j] : rot_read<N>(lut, t, j);
    }
  }
  __syncthreads();

  uint64_t *acc_r = acc + (size_t)r * N;
  double2 *buf_r = fbuf + (size_t)r * M;
  const double scale = 1.0 / (double)M;

  for (int i = 0; i < n_small; ++i) {
    const int a = (int)at[i];
    if (a == 0) continue;  // X^0*acc - acc == 0: exact no-op (uniform branch)

    // ---- diff -> digit, in the layout of the first forward pass (the twist lives in the twiddles) ----
    double2 x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = fft_point<LOGM, LOGM - 3, 3>(q, e);
      const uint64_t d0 = rot_read<N>(acc_r, a, p) - acc_r[p];
      const uint64_t d1 = rot_read<N>(acc_r, a, p + M) - acc_r[p + M];
      x[e] = make_double2(digit1_f64<BL>(d0), digit1_f64<BL>(d1));
    }
    fft_forward<LOGM, LOGM>(x, q, buf_r, psi);  // ends with a barrier: both spectra in fbuf

    // ---- external product: O_r = D_0 * G[0][r] + D_1 * G[1][r] ----
    const double2 *G = bsk_f + (size_t)i * 4 * M;  // [row][col][M]
    double2 d0v[8], d1v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = fft_point<LOGM, 0, NB0>(q, e);
      d0v[e] = fbuf[p];
      d1v[e] = fbuf[M + p];
    }
    __syncthreads();  // every thread has its operands: fbuf may be overwritten
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = fft_point<LOGM, 0, NB0>(q, e);
      const double2 g0 = G[(size_t)(0 * 2 + r) * M + p];
      const double2 g1 = G[(size_t)(1 * 2 + r) * M + p];
      double re = d0v[e].x * g0.x;
      double im = d0v[e].x * g0.y;
      re = __builtin_fma(-d0v[e].y, g0.y, re);
      im = __builtin_fma(d0v[e].y, g0.x, im);
      re = __builtin_fma(d1v[e].x, g1.x, re);
      im = __builtin_fma(d1v[e].x, g1.y, im);
      re = __builtin_fma(-d1v[e].y, g1.y, re);
      im = __builtin_fma(d1v[e].y, g1.x, im);
      x[e] = make_double2(re, im);
    }
    fft_inverse<LOGM, 0, NB0>(x, q, buf_r, tw);

    // ---- untwist, round to the torus, accumulate ----
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = fft_point<LOGM, LOGM - 3, 3>(q, e);
      const double2 v = cmul_conj(x[e], twist[p]);
      acc_r[p] += torus_from_f64(v.x * scale);
      acc_r[p + M] += torus_from_f64(v.y * scale);
    }
    __syncthreads();
  }

  // ---- sample extract (constant coefficient) ----
  uint64_t *out = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
  for (int j = tid; j < N; j += blockDim.x) out[j] = (j == 0) ? acc[0] : (uint64_t)0 - acc[N - j];
  if (tid == 0) out[N] = acc[N];
}

#endif  // FHES_LAB
// ------------------------------------------------------------------------------------------
// K2+K3 (generic): blind rotate + sample extract for EVERY classic parameter shape the public constructor
// accepts (`generate_keys_with_params(p: ClassicPBSParameters, l)`, /root/reference/src/lib.rs:50-62):
// glwe_dim k in 1..3, pbs_level L in 1..3, any pbs_base_log, N = 2^LOGN (256 .. 8192).  Correctness first: the
// arithmetic is exactly DESIGN.md section 4 (same transforms, FMA chains and rounding as the oracle), the data
// path is the simple one of variant 1 -- block = (k + 1) groups of M/8 threads, group r owns accumulator
// polynomial r -- with the accumulator and the digit spectra of the (k + 1) * L polynomials in a per-workgroup
// GLOBAL scratch (L2-resident; N = 8192 needs 384 KB per ciphertext, more than the LDS) and only the transforms'
// exchange buffers in LDS.  The 2_2 sets (k = 1, L = 1, N = 2048) never come here: they have variant 5.
//   LDS: fbuf double2[k + 1][M] | at u32[n + 1]
//   scratch per workgroup: acc u64[k + 1][N] | D double2[(k + 1) * L][M]
// ------------------------------------------------------------------------------------------
// LDSRES (round 3): accumulator and digit spectra in LDS instead of the global scratch, for the shapes whose working set
// fits next to the exchange buffers (everything but N = 8192 / N = 4096 with several levels): same code, same bits.
//   LDS with LDSRES: fbuf | at | acc u64[k + 1][N] | D double2[(k + 1) * L][M]
template <int LOGN, bool LDSRES = false>
__global__ __launch_bounds__((((LOGN >= 12 ? 2 : 4) << LOGN) / 16) > 1024 ? 1024 : (((LOGN >= 12 ? 2 : 4) << LOGN) / 16)) void blind_rotate_gen_kernel(  // (k + 1) <= 4 groups of N / 16 threads; N >= 4096: glwe_dim 1 only (two groups: 256 instead of 128 registers at N = 4096)
    NodeDesc d, uint64_t *__restrict__ arena, int ct_stride, const uint64_t *__restrict__ ks_out, int ks_stride,
    int n_small, int k, int L, int BL, const double2 *__restrict__ bsk_f, const uint64_t *__restrict__ lut_polys,
    const double2 *__restrict__ twist, const double2 *__restrict__ tw, const double2 *__restrict__ psi,
    unsigned char *__restrict__ scratch, size_t scratch_per_wg) {
  constexpr int N = 1 << LOGN, M = N / 2, LOGM = LOGN - 1, TP = M / 8;
  constexpr int NB0 = FftPlan<LOGM>::NB_LAST;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int kp1 = k + 1;
  double2 *fbuf = reinterpret_cast<double2 *>(smem_raw);                                  // [kp1][M]
  uint32_t *at = reinterpret_cast<uint32_t *>(smem_raw + (size_t)kp1 * M * 16);           // [n + 1]
  const size_t at_bytes = (size_t)((n_small + 1 + 3) & ~3) * 4;                           // keeps what follows 16-byte aligned
  uint64_t *acc = LDSRES ? reinterpret_cast<uint64_t *>(smem_raw + (size_t)kp1 * M * 16 + at_bytes)
                         : reinterpret_cast<uint64_t *>(scratch + (size_t)blockIdx.x * scratch_per_wg);  // [kp1][N]
  double2 *D = reinterpret_cast<double2 *>(acc + (size_t)kp1 * N);                        // [kp1 * L][M]

  const int tid = threadIdx.x;
  const int r = tid / TP;  // accumulator polynomial (and output column) of this thread
  const int q = tid % TP;
  const int vidx = d.v0 + blockIdx.x;
  const int inst = vidx / d.count, node = d.first + vidx % d.count;

  const uint64_t *ks = ks_out + (size_t)blockIdx.x * ks_stride;
  for (int j = tid; j <= n_small; j += blockDim.x)
    at[j] = (uint32_t)((((ks[j] >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
  __syncthreads();
  {  // acc = (0, ..., 0, X^{-b~} * LUT)
    const uint64_t *lut = lut_polys + (size_t)d.lut_ids[node] * N;
    const int t = (2 * N - (int)at[n_small]) & (2 * N - 1);
    for (int j = tid; j < kp1 * N; j += blockDim.x) {
      const int pr = j / N, jj = j % N;
      acc[j] = pr < k ? 0 : ((t == 0) ? lut[jj] : rot_read<N>(lut, t, jj));
    }
  }
  __syncthreads();

  uint64_t *acc_r = acc + (size_t)r * N;
  double2 *buf_r = fbuf + (size_t)r * M;
  const double scale = 1.0 / (double)M;
  const int total = BL * L;
  const uint64_t Bm1 = ((uint64_t)1 << BL) - 1, half = (uint64_t)1 << (BL - 1);

  for (int i = 0; i < n_small; ++i) {
    const int a = (int)at[i];
    if (a == 0) continue;  // X^0 * acc - acc == 0: exact no-op (uniform branch)
    // ---- digits of the rotated difference (closest representable, balanced digits taken LSB
    //      first, level L-1 = least significant), one forward transform per level ----
    uint64_t v0[8], v1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = fft_point<LOGM, LOGM - 3, 3>(q, e);
      const uint64_t d0 = rot_read<N>(acc_r, a, p) - acc_r[p];
      const uint64_t d1 = rot_read<N>(acc_r, a, p + M) - acc_r[p + M];
      v0[e] = (((d0 >> (64 - total - 1)) + 1) >> 1) & (((uint64_t)1 << total) - 1);
      v1[e] = (((d1 >> (64 - total - 1)) + 1) >> 1) & (((uint64_t)1 << total) - 1);
    }
    for (int l = L - 1; l >= 0; --l) {
      double2 x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int64_t g0 = (int64_t)(v0[e] & Bm1), g1 = (int64_t)(v1[e] & Bm1);
        v0[e] >>= BL;
        v1[e] >>= BL;
        if ((uint64_t)g0 >= half) { g0 -= (int64_t)1 << BL; v0[e] += 1; }
        if ((uint64_t)g1 >= half) { g1 -= (int64_t)1 << BL; v1[e] += 1; }
        x[e] = make_double2((double)g0, (double)g1);
      }
      fft_forward<LOGM, LOGM>(x, q, buf_r, psi);  // ends with a barrier: this group's spectrum in buf_r
      double2 *Dl = D + ((size_t)r * L + l) * M;
      for (int m = q; m < M; m += TP) Dl[m] = buf_r[m];
      __syncthreads();  // buf_r is re-used by the next level; D must be complete before the products
    }
    // ---- external product, output column r: sum over (row, level) in the oracle's order ----
    const double2 *G = bsk_f + (size_t)i * L * kp1 * kp1 * M;  // [L][row][col][M]
    double2 x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = fft_point<LOGM, 0, NB0>(q, e);
      double re = 0, im = 0;
      bool first = true;
      for (int row = 0; row < kp1; ++row)
        for (int l = 0; l < L; ++l) {
          const double2 dd = D[((size_t)row * L + l) * M + p];
          const double2 g = G[(((size_t)l * kp1 + row) * kp1 + r) * M + p];
          if (first) {
            re = dd.x * g.x;
            im = dd.x * g.y;
            first = false;
          } else {
            re = __builtin_fma(dd.x, g.x, re);
            im = __builtin_fma(dd.x, g.y, im);
          }
          re = __builtin_fma(-dd.y, g.y, re);
          im = __builtin_fma(dd.y, g.x, im);
        }
      x[e] = make_double2(re, im);
    }
    fft_inverse<LOGM, 0, NB0>(x, q, buf_r, tw);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = fft_point<LOGM, LOGM - 3, 3>(q, e);
      const double2 v = cmul_conj(x[e], twist[p]);
      acc_r[p] += torus_from_f64(v.x * scale);
      acc_r[p + M] += torus_from_f64(v.y * scale);
    }
    __syncthreads();  // every group's accumulator and buf are settled before the next rotation reads them
  }

  // ---- sample extract: a'_{r,0} = a_{r,0}, a'_{r,j} = -a_{r,N-j}, b' = b_0 ----
  uint64_t *out = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
  for (int j = tid; j < k * N; j += blockDim.x) {
    const int pr = j / N, jj = j % N;
    out[j] = (jj == 0) ? acc[(size_t)pr * N] : (uint64_t)0 - acc[(size_t)pr * N + N - jj];
  }
  if (tid == 0) out[(size_t)k * N] = acc[(size_t)k * N];
}

// ------------------------------------------------------------------------------------------
// Shared pieces of the throughput kernels (variant 5, multi-bit): 128 threads x 8 points per
// polynomial, three register passes of 3 index bits + one lane-pair stage on index bit 0 (DPP),
// XOR-swizzled LDS exchange buffers (conflict-free ds_read/write_b128 in every pass layout).
// Requires (log2(N) - 1) % 3 == 1  (N = 2048, N = 256).
// (Round 1 / early round 2 carried three more builds of the classic kernel -- a register-resident
// latency build and two earlier throughput builds, variants 2-4, with a phase-stamped diagnostic
// build: all slower than variant 5 at every launch size; they went when the forward transform changed.)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int fswz(int p) { return p ^ (((p >> 4) & 7) << 1); }

__device__ __forceinline__ double dpp_swap1(double v) {  // value of lane ^ 1
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// last FORWARD stage (index bit 0) across lane pairs: own + c * partner with the per-lane constant
// c = t on even lanes (s = u + t v) and (-t.x, t.y) on odd lanes (h = v + t' u): the FMA sequence of
// fwd_butterfly_last on both.
__device__ __forceinline__ void fft_fwd_stage0_dpp(double2 (&x)[8], const double2 (&c)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const double px = dpp_swap1(x[e].x), py = dpp_swap1(x[e].y);
    double re = __builtin_fma(c[e].x, px, x[e].x);
    re = __builtin_fma(-c[e].y, py, re);
    double im = __builtin_fma(c[e].x, py, x[e].y);
    im = __builtin_fma(c[e].y, px, im);
    x[e] = make_double2(re, im);
  }
}

// first INVERSE stage on index bit 0 across lane pairs (twiddle index 0: exactly 1)
// sgn = +1.0 on even lanes, -1.0 on odd lanes: fma(own, sgn, partner) is own + partner (u + v)
// resp. partner - own (u - v), correctly rounded like the add / sub it replaces.
__device__ __forceinline__ void fft_stage_bit0_dpp(double2 (&x)[8], double sgn) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const double px = dpp_swap1(x[e].x), py = dpp_swap1(x[e].y);
    x[e].x = __builtin_fma(x[e].x, sgn, px);
    x[e].y = __builtin_fma(x[e].y, sgn, py);
  }
}

// one point of the external product: D0*g0 + D1*g1 in the canonical FMA order
__device__ __forceinline__ double2 ext_mac(double2 D0, double2 D1, double2 g0, double2 g1) {
  double re = D0.x * g0.x;
  double im = D0.x * g0.y;
  re = __builtin_fma(-D0.y, g0.y, re);
  im = __builtin_fma(D0.y, g0.x, im);
  re = __builtin_fma(D1.x, g1.x, re);
  im = __builtin_fma(D1.x, g1.y, im);
  re = __builtin_fma(-D1.y, g1.y, re);
  im = __builtin_fma(D1.y, g1.x, im);
  return make_double2(re, im);
}

// butterflies of a 3-bit pass with the 7 twiddles held in registers.
// inverse: w[0..3] for the top bit (by e & 3), w[4..5] for the middle bit (by e & 1), w[6] for the low bit
//          (a twiddle depends on the index bits BELOW the stage's bit);
// forward: w[0] for the top bit, w[1..2] for the middle bit (by e >> 2), w[3..6] for the low bit (by e >> 1)
//          (a twiddle depends on the block, i.e. the index bits ABOVE the stage's bit)
template <bool FWD>
__device__ __forceinline__ void fft_butterflies_reg(double2 (&x)[8], const double2 (&w)[7]) {
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    const int fb = FWD ? (2 - st) : st;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e & (1 << fb)) continue;
      const int e1 = e | (1 << fb);
      if (FWD) {
        fwd_butterfly(x[e], x[e1], fb == 2 ? w[0] : (fb == 1 ? w[1 + (e >> 2)] : w[3 + (e >> 1)]));
      } else {
        inv_butterfly(x[e], x[e1], fb == 2 ? w[e & 3] : (fb == 1 ? w[4 + (e & 1)] : w[6]));
      }
    }
  }
}

// the 7 forward twiddles of the 3-bit pass that owns index bits [LO, LO+3) from a table laid out like psi
// (T[2^l + b]); h = q >> LO are the index bits above the pass.  Pass A (LO = LOGM-3) has h = 0 for every
// thread: uniform addresses.
// -i * w as a value: (w.y, -w.x).  The tables are exactly symmetric under quarter turns, so a twiddle whose
// angle is a quarter turn behind a loaded one is this -- the negation rides on the FMA's source modifier.
__device__ __forceinline__ double2 mul_neg_i(double2 w) { return make_double2(w.y, -w.x); }

// Four loads instead of seven (round 3): the twiddle of block 2b+1 is -i times the twiddle of block 2b (their exponents
// differ by N/2: E(m) = 1 - 4 bitrev(m), the block's lowest index bit becomes the highest bit of the reversed index).
// DERIVE = false loads all seven (the multi-bit kernel: measured 1.9 % slower with the derived form, its limit is the key
// stream, not the LDS pipe; same values either way).
template <int LOGM, int LO, bool DERIVE = true>
__device__ __forceinline__ void load_fwd_twiddles(double2 (&w)[7], int q, const double2 *T) {
  constexpr int L0 = LOGM - 3 - LO;
  const int h = q >> LO;
  w[0] = T[(1 << L0) + h];
  w[1] = T[(2 << L0) + 2 * h];
  w[2] = DERIVE ? mul_neg_i(w[1]) : T[(2 << L0) + 2 * h + 1];
  w[3] = T[(4 << L0) + 4 * h];
  w[4] = DERIVE ? mul_neg_i(w[3]) : T[(4 << L0) + 4 * h + 1];
  w[5] = T[(4 << L0) + 4 * h + 2];
  w[6] = DERIVE ? mul_neg_i(w[5]) : T[(4 << L0) + 4 * h + 3];
}

template <int LOGM, int LO>
__device__ __forceinline__ void load_pass_twiddles(double2 (&w)[7], int q, const double2 *__restrict__ tw) {
  // same index formula as fft_butterflies<LOGM, *, LO, 3>
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int fb = k < 4 ? 2 : (k < 6 ? 1 : 0);
    const int e = k < 4 ? k : (k < 6 ? (k - 4) : 0);  // representative point with bit fb clear
    const int beta = LO + fb, sh = LOGM - 1 - beta;
    const int p = fft_point<LOGM, LO, 3>(q, e);
    w[k] = tw[(p & ((1 << beta) - 1)) << sh];
  }
}


// padded exchange-buffer layout: slot(p) = (p>>7)*142 + ((p>>4)&7)*18 + (p&15).  Every 3-bit pass
// layout is then base(q) + e * const (one address register per layout, immediate offsets) and b128
// accesses stay conflict-free (16 consecutive lanes always cover 16 consecutive 16-byte groups).
__device__ __forceinline__ constexpr int fpad(int p) { return (p >> 7) * 142 + ((p >> 4) & 7) * 18 + (p & 15); }

typedef unsigned int fhes_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double2 buf_load_d2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  const fhes_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}

// the 7 twiddles of a 3-bit INVERSE pass from the LDS stage tables T[2^beta + j] = tw[j << (LOGM-1-beta)]
// (same values and order as load_pass_twiddles): issued BEFORE the barrier that precedes the pass, so that
// only the data reads remain on the critical path after it
// Four loads instead of seven (round 3): setting the pass's second register bit moves a twiddle index by M/4, a quarter turn,
// for the top stage (w[2] = -i w[0], w[3] = -i w[1]), setting its first register bit does the same for the middle stage
// (w[5] = -i w[4]); the tables are exactly symmetric (mul_neg_i).  `wtop`: entries 0 and 1 are used.
template <int LOGM, int LO>
__device__ __forceinline__ void load_pass_twiddles_tab(double2 (&w)[7], int q, const double2 *T, const double2 (&wtop)[4]) {
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int fb = k < 4 ? 2 : (k < 6 ? 1 : 0);
    const int e = k < 4 ? k : (k < 6 ? (k - 4) : 0);
    const int beta = LO + fb;
    if (k == 2 || k == 3) w[k] = mul_neg_i(w[k - 2]);
    else if (k == 5) w[k] = mul_neg_i(w[4]);
    else if (beta == LOGM - 1) w[k] = wtop[e & 3];
    else w[k] = T[(1 << beta) + (fft_point<LOGM, LO, 3>(q, e) & ((1 << beta) - 1))];
  }
}

// ------------------------------------------------------------------------------------------
// K2+K3 (N = 8192, round 4): blind rotate + sample extract for the classic shapes with N = 8192
// (glwe_dim 1, L <= 3; e.g. the 3_3-shaped set: two levels of base 2^15).  The generic kernel above gives
// every polynomial its own thread group -- 1024 threads at N = 8192, 128 registers each, 250 of them spilled -- and keeps
// the (k + 1) * L digit spectra in a global scratch (384 KB per ciphertext).  Here ONE group of N / 16 threads (512 at
// N = 8192: 256 registers, nothing spilled) walks the polynomials one after the other, and the spectra never leave the
// chip: the products are accumulated in registers as each digit transform completes,
//     for (row, level) in the oracle's order:  D = FFT(digit_level(X^a acc_row - acc_row));  O_col += D * G[level][row][col]
// (the FMA chain of DESIGN.md 4.7 term by term: same operands, same order, same bits), then one inverse transform per
// column.  Digits of level l are recomputed from the rotated difference for every level (a carry chain of at most L - 1
// steps) instead of being carried across the transforms.
//   LDS: buf double2[M] | obuf double2[M] (2 x 64 KB) | TF, TI double2[512] each | at u32[n + 1];   scratch per workgroup: acc u64[2][N]
// ------------------------------------------------------------------------------------------
// one 3-bit register pass with its 7 twiddles in registers (index bits [LO, LO + 3); the stage of index bit 0 has its own
// forms: fwd_butterfly_last / the exact u +- v of the inverse transform) -- fft_butterflies with the loads taken out
template <bool FWD, int LO>
__device__ __forceinline__ void fft_pass_w(double2 (&x)[8], const double2 (&w)[7]) {
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    const int fb = FWD ? (2 - st) : st;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e & (1 << fb)) continue;
      const int e1 = e | (1 << fb);
      if (FWD) {
        const double2 t = fb == 2 ? w[0] : (fb == 1 ? w[1 + (e >> 2)] : w[3 + (e >> 1)]);
        if (LO + fb == 0) fwd_butterfly_last(x[e], x[e1], t);
        else fwd_butterfly(x[e], x[e1], t);
      } else if (LO + fb == 0) {  // twiddle exactly 1
        const double2 u = x[e], v = x[e1];
        x[e] = make_double2(u.x + v.x, u.y + v.y);
        x[e1] = make_double2(u.x - v.x, u.y - v.y);
      } else {
        inv_butterfly(x[e], x[e1], fb == 2 ? w[e & 3] : (fb == 1 ? w[4 + (e & 1)] : w[6]));
      }
    }
  }
}

template <int LOGN, int L>
__global__ __launch_bounds__((1 << LOGN) / 16) void blind_rotate_big_kernel(
    NodeDesc d, uint64_t *__restrict__ arena, int ct_stride, const uint64_t *__restrict__ ks_out, int ks_stride,
    int n_small, int BL, const double2 *__restrict__ bsk_f, const uint64_t *__restrict__ lut_polys,
    const double2 *__restrict__ twist, const double2 *__restrict__ tw, const double2 *__restrict__ psi,
    unsigned char *__restrict__ scratch, size_t scratch_per_wg) {
  constexpr int N = 1 << LOGN, M = N / 2, LOGM = LOGN - 1, TP = M / 8;
  static_assert(LOGM == 12, "written for four passes of three index bits: N = 8192 (N = 4096 has LOGM = 11 and stays on the generic kernel)");
  constexpr int KP1 = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double2 *buf = reinterpret_cast<double2 *>(smem_raw);                          // [M]   exchange buffer of the transforms
  double2 *obuf = buf + M;                                                       // [M]   products of column 1
  double2 *TF = obuf + M;                                                        // [512] forward twiddles psi[0, 512): passes A, B, C
  double2 *TI = TF + 512;                                                        // [512] inverse stage tables: TI[2^beta + j] = tw[j << (LOGM - 1 - beta)], beta < 9
  uint32_t *at = reinterpret_cast<uint32_t *>(TI + 512);                         // [n + 1]
  uint64_t *acc = reinterpret_cast<uint64_t *>(scratch + (size_t)blockIdx.x * scratch_per_wg);  // [2][N]

  const int q = threadIdx.x;
  const int vidx = d.v0 + blockIdx.x;
  const int inst = vidx / d.count, node = d.first + vidx % d.count;

  const uint64_t *ks = ks_out + (size_t)blockIdx.x * ks_stride;
  for (int j = q; j <= n_small; j += TP)
    at[j] = (uint32_t)((((ks[j] >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
  // twiddles of the first three forward passes and of the first three inverse passes live in LDS (16 KB): a pass's twiddles
  // cost an LDS round trip instead of one through L2; the last pass of each direction has a different set per thread and keeps
  // the global tables, requested at the START of the transform
  for (int idx = q; idx < 512; idx += TP) {
    TF[idx] = psi[idx];
    const int beta = idx < 2 ? 0 : 31 - __builtin_clz(idx);  // idx = 2^beta + j
    TI[idx] = idx < 2 ? make_double2(1.0, 0.0) : tw[(idx - (1 << beta)) << (LOGM - 1 - beta)];
  }
  __syncthreads();
  auto load_inv_tab = [&](double2 (&w)[7], auto lo_tag) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_tag)::value;
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) {
      const int fb = kk < 4 ? 2 : (kk < 6 ? 1 : 0);
      const int e = kk < 4 ? kk : (kk < 6 ? (kk - 4) : 0);
      const int beta = LO + fb;
      w[kk] = beta == 0 ? make_double2(1.0, 0.0) : TI[(1 << beta) + (fft_point<LOGM, LO, 3>(q, e) & ((1 << beta) - 1))];
    }
  };
  {  // acc = (0, X^{-b~} * LUT)
    const uint64_t *lut = lut_polys + (size_t)d.lut_ids[node] * N;
    const int t = (2 * N - (int)at[n_small]) & (2 * N - 1);
    for (int j = q; j < KP1 * N; j += TP) {
      const int pr = j / N, jj = j % N;
      acc[j] = pr < KP1 - 1 ? 0 : ((t == 0) ? lut[jj] : rot_read<N>(lut, t, jj));
    }
  }
  __syncthreads();

  const double scale = 1.0 / (double)M;
  const int total = BL * L;
  const uint64_t Bm1 = ((uint64_t)1 << BL) - 1, half = (uint64_t)1 << (BL - 1), tmask = ((uint64_t)1 << total) - 1;
  // balanced digit of level `lt` (level L - 1 = least significant) of the rounded prefix v
  auto digit = [&](uint64_t v, int lt) __attribute__((always_inline)) {
    int64_t g = 0;
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
      if (l < lt) break;
      g = (int64_t)(v & Bm1);
      v >>= BL;
      if ((uint64_t)g >= half) { g -= (int64_t)1 << BL; v += 1; }
    }
    return (double)g;
  };
  // exchange between two passes through buf: layout (LO_FROM, 3) -> layout (LO_TO, 3)
#define FHES_XCHG(LO_FROM, LO_TO)                                                              \
  do {                                                                                         \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) buf[fft_point<LOGM, LO_FROM, 3>(q, e)] = x[e]; \
    __syncthreads();                                                                           \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) x[e] = buf[fft_point<LOGM, LO_TO, 3>(q, e)];   \
  } while (0)

  for (int i = 0; i < n_small; ++i) {
    const int a = (int)at[i];
    if (a == 0) continue;  // X^0 * acc - acc == 0: exact no-op (uniform branch)
    const double2 *G = tw_here(bsk_f + (size_t)i * L * KP1 * KP1 * M);  // [L][row][col][M]
    // the products of column 0 accumulate in registers, those of column 1 in LDS (thread-private slots q + e * TP: no conflicts,
    // no barrier) -- both columns in registers next to the prefetched key rows spilled 128 registers
    double2 O0[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      O0[e] = make_double2(-0.0, -0.0);
      obuf[q + e * TP] = make_double2(-0.0, -0.0);
    }
#pragma unroll 1
    for (int row = 0; row < KP1; ++row) {
      const uint64_t *acc_r = acc + (size_t)row * N;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        double2 x[8], w[7], wd[7];
        load_fwd_twiddles<LOGM, 0, false>(wd, q, tw_here(psi));  // the last pass's twiddles (global): in flight across the first three passes
        // the rotated difference is re-read for every level (L2-resident scratch) instead of being carried in registers
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int p = fft_point<LOGM, LOGM - 3, 3>(q, e);
          const uint64_t d0 = rot_read<N>(acc_r, a, p) - acc_r[p];
          const uint64_t d1 = rot_read<N>(acc_r, a, p + M) - acc_r[p + M];
          x[e] = make_double2(digit((((d0 >> (64 - total - 1)) + 1) >> 1) & tmask, l), digit((((d1 >> (64 - total - 1)) + 1) >> 1) & tmask, l));
        }
        // forward transform: every pass's twiddles are requested one pass ahead (behind the butterflies, in front of the
        // exchange and its barrier), the key rows of this term before the last pass
        load_fwd_twiddles<LOGM, LOGM - 3, false>(w, q, TF);
        fft_pass_w<true, LOGM - 3>(x, w);
        load_fwd_twiddles<LOGM, LOGM - 6, false>(w, q, TF);
        FHES_XCHG(LOGM - 3, LOGM - 6);
        fft_pass_w<true, LOGM - 6>(x, w);
        load_fwd_twiddles<LOGM, LOGM - 9, false>(w, q, TF);
        FHES_XCHG(LOGM - 6, LOGM - 9);
        fft_pass_w<true, LOGM - 9>(x, w);
        const double2 *G0 = G + (((size_t)l * KP1 + row) * KP1 + 0) * M, *G1 = G0 + M;
        double2 g0[8], g1[8];
        FHES_XCHG(LOGM - 9, 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) { g0[e] = G0[fft_point<LOGM, 0, 3>(q, e)]; g1[e] = G1[fft_point<LOGM, 0, 3>(q, e)]; }
        fft_pass_w<true, 0>(x, wd);
        // the spectrum is in the registers in the layout the products want (pass (0, 3)): no trip through buf.
        // The chain's first term is a plain product in the oracle; fma(a, b, -0.0) IS that product, bit for bit (a * b + (-0) keeps
        // the product's value and the sign of a zero), so the accumulators start at -0.0 and the loop body has one form
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const double2 dd = x[e];
          double2 o1 = obuf[q + e * TP];
          O0[e] = make_double2(__builtin_fma(dd.x, g0[e].x, O0[e].x), __builtin_fma(dd.x, g0[e].y, O0[e].y));
          o1 = make_double2(__builtin_fma(dd.x, g1[e].x, o1.x), __builtin_fma(dd.x, g1[e].y, o1.y));
          O0[e] = make_double2(__builtin_fma(-dd.y, g0[e].y, O0[e].x), __builtin_fma(dd.y, g0[e].x, O0[e].y));
          o1 = make_double2(__builtin_fma(-dd.y, g1[e].y, o1.x), __builtin_fma(dd.y, g1[e].x, o1.y));
          obuf[q + e * TP] = o1;
        }
        __syncthreads();  // every thread has taken its points out of buf before the next transform writes there
      }
    }
    // every rotated difference of this iteration has been taken: the accumulator may change now
#pragma unroll 1
    for (int col = 0; col < KP1; ++col) {
      double2 x[8], w[7], wl[7];
      load_pass_twiddles<LOGM, 9>(wl, q, tw_here(tw));  // the last pass's twiddles (global): in flight across the first three passes
      load_inv_tab(w, std::integral_constant<int, 0>{});
      if (col == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = O0[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = obuf[q + e * TP];
      }
      fft_pass_w<false, 0>(x, w);
      load_inv_tab(w, std::integral_constant<int, 3>{});
      FHES_XCHG(0, 3);
      fft_pass_w<false, 3>(x, w);
      load_inv_tab(w, std::integral_constant<int, 6>{});
      FHES_XCHG(3, 6);
      fft_pass_w<false, 6>(x, w);
      int zc = 0;
      asm volatile("" : "+s"(zc));  // the accumulator words are fetched behind this point (see tw_here)
      uint64_t *acc_c = acc + (size_t)col * N + zc;
      const double2 *twist_l = tw_here(twist);
      double2 tws[8];
      uint64_t alo[8], ahi[8];
      // untwist factors and the accumulator words of the update: requested in front of the last exchange
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int p = fft_point<LOGM, LOGM - 3, 3>(q, e);
        tws[e] = twist_l[p];
        alo[e] = acc_c[p];
        ahi[e] = acc_c[p + M];
      }
      FHES_XCHG(6, 9);
      fft_pass_w<false, 9>(x, wl);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int p = fft_point<LOGM, LOGM - 3, 3>(q, e);
        const double2 v = cmul_conj(x[e], tws[e]);
        acc_c[p] = alo[e] + torus_from_f64(v.x * scale);
        acc_c[p + M] = ahi[e] + torus_from_f64(v.y * scale);
      }
      __syncthreads();  // buf is re-used by the next column; the accumulator is settled before the next rotation reads it
    }
  }
#undef FHES_XCHG

  // ---- sample extract: a'_0 = a_0, a'_j = -a_{N-j}, b' = b_0 ----
  uint64_t *out = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
  for (int j = q; j < N; j += TP) out[j] = (j == 0) ? acc[0] : (uint64_t)0 - acc[N - j];
  if (q == 0) out[N] = acc[N];
}

// ------------------------------------------------------------------------------------------
// K2+K3 (variant 5): the throughput blind rotate (every launch size).  256 threads = two halves (output
// polynomial r) x 128 threads x 8 points, two workgroups per CU (<= 256 registers, exactly 80 KB LDS each).
// Structure of an iteration (4 workgroup barriers):
//   rotated reads of the image (Y), digits | pass A -> X | B1 | pass B (X) -> Y, wave-local | pass C (Y),
//   lane-pair stage, spectra -> X | B3 | other half's spectrum (X), multiply-accumulate with the GGSW rows,
//   inverse lane-pair stage, inverse pass C -> Y, wave-local | inverse pass B (Y) -> Y in place | B5 |
//   inverse pass A (Y), untwist, round, accumulate, image -> Y in place | B6
//   * the thread's own accumulator coefficients stay in registers, so the image in Y is dead between the
//     rotated reads and the write-back and Y doubles as an exchange buffer; the image lives in the slots of
//     the pass-A layout (image_word below), so the write-back and the exchange before it are written by the
//     thread that read the slots: no barrier between them, and X keeps the spectra until the next iteration
//     (no barrier "the other half has read my spectrum");
//   * the exchanges between passes B and C stay inside a wave (q >> 6 is index bit LOGM-1 in both layouts and
//     the LDS operations of one wave execute in order);
//   * the twiddles of the next pass are requested from LDS tables BEFORE each barrier; pass A of the forward
//     transform takes the same 7 twiddles on every thread (SGPRs); GGSW rows are prefetched one transform
//     ahead, the lane-pair constants after B1, the untwist factors after the multiply-accumulate;
//   * X and Y are XOR-swizzled (fswz): conflict-free ds_read/write_b128 in every pass layout.
// The INTEGER side avoids VCC (measured, scripts/micro/f64_rate.hip: every VALU slot costs ~4.4 cycles next
// to f64 work, a VALU write of VCC followed by a VALU read of it costs two extra wait states, v_cndmask with
// VCC 23 cycles):
//   * registers and the LDS image hold the NEGATED accumulator  nacc = -acc  (mod 2^64), so the
//     decomposed difference  (+-)acc[s] - acc[p]  is  nacc[p] + ((-+)nacc[s])  -- additions only;
//   * the conditional negation of the rotated operand is  (v ^ M) + c  with the wrap mask M taken
//     from one bit of the rotated index (v_bfe_i32), no compare / select;
//   * 64-bit additions are single v_lshl_add_u64 instructions (no carry chain through VCC); the
//     rounding bit 2^40 of the closest-representable digit rides in the high word of the constant
//     pair that carries c;
//   * the accumulator update adds rint(-y) = -rint(y) (round-half-even is symmetric; the untwist
//     factors enter negated through the free f64 source modifiers), the two dwords of the rounded
//     value are assembled in place and added with one v_lshl_add_u64.
// LDS: Y = double2[2][M] (exchange + image of -acc) | X = double2[2][M] | inverse stage tables double2[M/2] |
// forward twiddles double2[M/2]; the switched mask a~_i comes through the scalar cache, one iteration ahead.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t add64(uint64_t a, uint64_t b) {  // one instruction, no VCC
  uint64_t r;
  asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// rint(y) mod 2^64 as in torus_from_f64_fast, returned so that ONE 64-bit add consumes it.  The argument is
// ys = y * 2^-64 (the scale rides in the untwist factors: a power of two, every product and FMA scales exactly), which
// makes the reduction mod 2^64 "ys - rint(ys)" (v_rndne_f64: the same nearest multiple, ties to even, as the
// (y + 1.5*2^116) - 1.5*2^116 of the unscaled form) -- 6 f64 operations instead of 7 -- and the three integer
// operations of the borrow one: with t2 = rem + 1.5*2^-12 the high dword of t2 is 0x3F380000 (rem >= 0) or 0x3F37FFFF
// (rem < 0: borrow one from the high word), so adding it to the low mantissa dword of t1 = z + (1.5*2^20 + K * 2^-32),
// K = -0x3F380000 mod 2^32, gives the high word directly (K is even: ties of t1 round as before).
__device__ __forceinline__ uint64_t torus_from_f64_pair(double ys) {
  const double C1 = 0x1.8p20 + 3234332672.0 /* 0xC0C80000 */ * 0x1p-32, C2 = 0x1.8p-12;
  const double z = ys - __builtin_rint(ys);  // exact, |z| <= 1/2
  const double t1 = z + C1;
  const double rem = z - (t1 - C1);
  const double t2 = rem + C2;
  uint32_t hi = (uint32_t)__double2loint(t1) + (uint32_t)__double2hiint(t2);
  asm("" : "+v"(hi));  // opaque: keeps the compiler from spreading (a + b) << 32 over several 64-bit adds
  return ((uint64_t)hi << 32) | (uint64_t)(uint32_t)__double2loint(t2);
}

// image of -acc inside the exchange buffer Y (variant 5): coefficient j of a polynomial lives in the complex
// slot fswz(j mod M) -- the slot of the pass-A layout -- in the 8-byte word j / M, so that a thread writes its
// pair (coefficient p, coefficient p + M) with ONE ds_write_b128 into the slot it has just read.  Price: the
// rotated 8-byte reads of 32 consecutive lanes stride 16 bytes and hit every bank twice (76 conflict cycles per
// wave and iteration, SQ_LDS_BANK_CONFLICT).  XORing index bit 4 into the word number makes those reads
// conflict-free but turns the write-back into 16 ds_write_b64 at a 16-byte stride (2-way on the 32-bank store
// path): measured 2.7 % slower in the same run, so the plain order stays.
template <int LOGM>
__device__ __forceinline__ int image_word(int j) {
  const int low = j & ((1 << LOGM) - 1);
  return 2 * fswz(low) + ((j >> LOGM) & 1);
}

// PRIO: 0 = head of an iteration at high wave priority (as variant 4), 1 = static priority by the
// wave slot the workgroup occupies (one of the two workgroups of a CU always wins arbitration and runs
// at its solo speed, the other fills its gaps), 2 = no priority hints
// PRIO 4 / 5: timing-only ablations (WRONG results; scripts/bench_pbs.py variants 9 / 10, round 3):
// 4 = without the LDS exchanges between the passes (rotated reads, table reads and the image write-back stay),
// 5 = without the butterfly passes and lane-pair stages (every LDS operation stays).
#define FHES_PASS(FWD, x, w) do { if (PRIO != 5) fft_butterflies_reg<FWD>(x, w); } while (0)
typedef const uint64_t __attribute__((address_space(3))) *lds_u64_ptr;
template <int LOGN, int BL, int PRIO>
__global__ __launch_bounds__(((1 << LOGN) / 8) < 64 ? 64 : ((1 << LOGN) / 8), 2) void blind_rotate_v5_kernel(
    NodeDesc d, uint64_t *__restrict__ arena, int ct_stride, const uint64_t *__restrict__ ks_out,
    int ks_stride, int n_small, const double2 *__restrict__ bsk_e, const uint64_t *__restrict__ lut_polys,
    const double2 *__restrict__ twist_inv /* twist / (M * 2^64): torus_from_f64_pair */, const double2 *__restrict__ tw,
    const double2 *__restrict__ psi /* forward twiddles [M] */, const double2 *__restrict__ cdpp /* [8][M/8] */) {
  constexpr int N = 1 << LOGN, M = N / 2, LOGM = LOGN - 1, TP = M / 8;
  constexpr int NP = (LOGM - 1) / 3;
  static_assert(LOGM % 3 == 1 && (NP == 2 || NP == 3), "supported: N = 256, N = 2048");
  static_assert(BL >= 1 && BL <= 31, "single-level digit must fit the high dword");
  constexpr int SLOTS = M;  // X is XOR-swizzled like Y (fswz): the padded layout of variant 4 costs one
                            // extra LDS cycle per lane group on every ds_read_b128 (the groups are
                            // {0-3,12-15,20-27}.., not 16 consecutive lanes) -- measured 96 conflict cycles per
                            // wave and iteration (SQ_LDS_BANK_CONFLICT), predicted exactly by the bank model
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // Y: exchange buffer AND image of -acc.  Coefficient j of polynomial r lives in the complex slot of half r's
  // pass-A layout: slot fswz(j mod M), component j / M -- a thread writes its accumulator to the slots it has
  // just read in the last inverse pass (in place: no barrier before the write-back)
  double2 *ybuf = reinterpret_cast<double2 *>(smem_raw);                      // [2][M]
  double2 *xbuf = reinterpret_cast<double2 *>(smem_raw + (size_t)2 * N * 8);  // [2][SLOTS]    X
  double2 *T = xbuf + (size_t)2 * SLOTS;                                      // [M/2] inverse stage tables
  double2 *TF = T + M / 2;                                                    // [M/2] forward twiddles psi[0, M/2)
  // (no room for anything else: 2 x 80 KB fill the CU's LDS; the switched mask a~_i is read from the
  // keyswitch output through the scalar cache, one iteration ahead)

  // byte address of the dynamic LDS block (0: the kernel has no static LDS; the rotated reads OR it into their offsets)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
  if (lds0 & (unsigned)(2 * N * 8 - 1)) __builtin_trap();  // the OR needs the block on a multiple of the image size
  const int tid = threadIdx.x;
  // the half (output polynomial) of this thread; with TP >= 64 a wave lies inside one half: r in an SGPR, the two
  // forms of the multiply-accumulate behind a scalar branch (left as a vector value the compiler turned them into 64 selects)
  const int r = TP >= 64 ? __builtin_amdgcn_readfirstlane(tid / TP) : tid / TP;
  const int q = tid % TP;
  const double sgn = (q & 1) ? -1.0 : 1.0;
  const int vidx = d.v0 + blockIdx.x;
  const int inst = vidx / d.count, node = d.first + vidx % d.count;

  const uint64_t *ks = ks_out + (size_t)blockIdx.x * ks_stride;
  auto modswitch = [&](int j) __attribute__((always_inline)) {  // uniform: scalar load + scalar ALU
    return (int)((((ks[j] >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
  };
  for (int idx = 2 + tid; idx < M / 2; idx += blockDim.x) {
    const int beta = 31 - __builtin_clz(idx);  // idx = 2^beta + j
    T[idx] = tw[(idx - (1 << beta)) << (LOGM - 1 - beta)];
  }
  for (int idx = tid; idx < M / 2; idx += blockDim.x) TF[idx] = psi[idx];
  __syncthreads();

  constexpr int LO_A = LOGM - 3, LO_B = LOGM - 6, LO_C = NP == 3 ? LOGM - 9 : LOGM - 6;
  double2 wfa[7];   // forward pass A: the same 7 twiddles for every thread (blocks 0 of the three top stages)
#pragma unroll
  for (int kk = 0; kk < 7; ++kk) wfa[kk] = psi[1 + kk];
  const double2 wnone[4] = {};  // passes B / C take every twiddle from the LDS tables
  // inverse, stage LOGM-1 of pass A: tw[q + TP * e], e = 0..3, fetched from global memory when the pass is near
  const __amdgpu_buffer_rsrc_t rs_twg = __builtin_amdgcn_make_buffer_rsrc((void *)tw, 0, (M / 2) * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g =
      __builtin_amdgcn_make_buffer_rsrc((void *)bsk_e, 0, (unsigned)((size_t)n_small * 4 * M * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cd = __builtin_amdgcn_make_buffer_rsrc((void *)cdpp, 0, M * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ti = __builtin_amdgcn_make_buffer_rsrc((void *)twist_inv, 0, M * 16, 0x00020000);
  const unsigned voff_q = (unsigned)q * 16u;                 // twist[q + e*TP]
  const unsigned voff_g = (unsigned)(r * 8 * TP + q) * 16u;  // G[((row*2 + r)*8 + e)*TP + q]

  // nacc: NEGATED coefficients p = q + e*TP (lo) and p + M (hi) of polynomial r
  uint64_t nlo[8], nhi[8];
  {
    const uint64_t *lut = lut_polys + (size_t)d.lut_ids[node] * N;
    const int t = (2 * N - modswitch(n_small)) & (2 * N - 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = fft_point<LOGM, LO_A, 3>(q, e);
      uint64_t v0 = 0, v1 = 0;
      if (r == 1) {
        v0 = (t == 0) ? lut[p] : rot_read<N>(lut, t, p);
        v1 = (t == 0) ? lut[p + M] : rot_read<N>(lut, t, p + M);
      }
      nlo[e] = (uint64_t)0 - v0;
      nhi[e] = (uint64_t)0 - v1;
      reinterpret_cast<uint64_t *>(ybuf + (size_t)r * M)[image_word<LOGM>(p)] = nlo[e];
      reinterpret_cast<uint64_t *>(ybuf + (size_t)r * M)[image_word<LOGM>(p + M)] = nhi[e];
    }
  }
  __syncthreads();

  double2 *x_r = xbuf + (size_t)r * SLOTS;
  const double2 *x_o = xbuf + (size_t)(1 - r) * SLOTS;
  double2 *y_r = ybuf + (size_t)r * M;

  // closest-representable digit of x with the rounding bit already added to x:
  // (int)hi >> (32 - BL)
  constexpr uint64_t ROUND = (uint64_t)1 << (63 - BL);

  if (PRIO == 1) {
    const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));  // HW_REG_HW_ID.WAVE_ID: slot inside the SIMD
    if (slot & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
  }
  auto modswitch_of = [&](uint64_t w) __attribute__((always_inline)) {
    return (int)((((w >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
  };
  int a_next = modswitch(0);
  for (int i = 0; i < n_small; ++i) {
    const int a = a_next;
    // the next mask word is requested one iteration ahead (the row has n+1 words) and turned into a_next just before
    // barrier B1, where the wave waits for its LDS operations anyway: left to itself the compiler kept the raw word and
    // re-used its registers for the result at the head of the next iteration -- a scalar-load round trip (a
    // write-after-write wait) in front of every iteration's first instruction.
    // a == 0 is not skipped (one iteration in 2N): the digits of X^0 acc - acc are all zero, every transform of zeros
    // is zero and the accumulator receives rint(0) -- the same bits as the oracle's skip, and the loop stays one block.
    double2 x[8];
    {
      // rotated coefficient s = (p - a) mod 2N of the own point p = q + e*TP (h = 0) and of p + M (h = 1): both
      // live in ONE slot (index s mod M), the component is bit LOGM of s.  The wrap bit (bit LOGN) is taken
      // INVERTED: the image holds -acc, the operand wanted is +acc[s] when the rotation wrapped (X^N = -1) and
      // -acc[s] = image[s] when it did not.
      // everything in byte offsets: bits [4, LOGM+4) = slot index, bit LOGM+4 = component, bit LOGM+5 = wrap
      const unsigned sb = ((unsigned)(q - a) << 4) ^ (1u << (LOGM + 5));
      // fswz XORs index bits 4..6 into bits 1..3: byte-offset bits 8..10 into 5..7.  With TP >= 128 the points of
      // a thread differ in index bits >= 7 only, so the term is the same for all of them
      // ... and the byte offset of this half's image rides along (bit LOGM+4 and up: outside the slot mask)
      const unsigned img_base = (unsigned)r * (unsigned)(M * 16) + lds0;
      const unsigned swz_all = ((sb >> 3) & 0xE0u) | img_base;
      // all sixteen rotated reads go out as ONE burst before the first digit is formed (round 3: the compiler's own
      // order was read, wait, use, pair by pair -- eight exposed LDS round trips at the head of every iteration)
      uint64_t va[8], vb[8];
      int hmv[8], m0v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned se = sb + (unsigned)(e * TP * 16);
        const unsigned swz = TP >= 128 ? swz_all : (((se >> 3) & 0xE0u) | img_base);
        hmv[e] = __builtin_amdgcn_sbfe(se, LOGM + 4, 1);             // -1 when h = 0 reads the upper component
        m0v[e] = __builtin_amdgcn_sbfe(se, LOGM + 5, 1);             // h = 0: 0 take image, -1 take -image
        const unsigned off0 = ((se & (unsigned)((M - 1) << 4)) ^ swz) | ((unsigned)hmv[e] & 8u);
        // LDS addresses as integers: through `smem_raw + off` every address pays an add of the (zero) link-time base
        va[e] = *(lds_u64_ptr)(uintptr_t)(off0);
        vb[e] = *(lds_u64_ptr)(uintptr_t)(off0 ^ 8u);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int m0 = m0v[e];
        const int m1 = m0 ^ hmv[e];                                  // h = 1: adding M carries into the wrap bit iff hm
        uint64_t dd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m32 = h == 0 ? m0 : m1;
          const uint64_t v = h == 0 ? va[e] : vb[e];
          const uint64_t mm = ((uint64_t)(uint32_t)m32 << 32) | (uint64_t)(uint32_t)m32;  // both dwords: m32 is 0 or -1
          const uint64_t kk = ROUND | (uint64_t)(uint32_t)(m32 & 1);   // carry of the negation + rounding bit of the digit
          dd[h] = add64(add64(v ^ mm, h == 0 ? nlo[e] : nhi[e]), kk);
        }
        const double d0 = (double)((int)(uint32_t)(dd[0] >> 32) >> (32 - BL));
        const double d1 = (double)((int)(uint32_t)(dd[1] >> 32) >> (32 - BL));
        x[e] = make_double2(d0, d1);  // no twist: it lives in the forward twiddles
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // requested behind the rotated reads: a scalar load in flight turns every counted wait on the LDS reads into "wait for all"
    const uint64_t ks_next = ks[i + 1];
    double2 g0[8], g1[8];
    {
      const unsigned so = (unsigned)i * (unsigned)(4 * M * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        g0[e] = buf_load_d2(rs_g, voff_g, so + (unsigned)((0 * 16 + e) * TP * 16));
        g1[e] = buf_load_d2(rs_g, voff_g, so + (unsigned)((1 * 16 + e) * TP * 16));
      }
    }
    double2 w[7];
    FHES_PASS(true, x, wfa);
    if (PRIO != 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x_r[fswz(fft_point<LOGM, LO_A, 3>(q, e))] = x[e];
    }
    load_fwd_twiddles<LOGM, LO_B>(w, q, TF);
    a_next = modswitch_of(ks_next);
    asm volatile("" : "+s"(a_next));  // materialised here, not at the head of the next iteration
    __syncthreads();  // B1 (also: every rotated read of Y is done)
    if (PRIO == 0) __builtin_amdgcn_s_setprio(0);
    // constants of the lane-pair stage: requested here, consumed after the remaining register passes
    // (measured: requested at the top of the iteration instead, 140.5 against 134.8 ms per 16 384 PBS)
    double2 cd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cd[e] = buf_load_d2(rs_cd, voff_q, (unsigned)(e * TP * 16));
    if (PRIO != 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = x_r[fswz(fft_point<LOGM, LO_B, 3>(q, e))];
      __builtin_amdgcn_sched_barrier(0);  // all eight reads of the exchange go out before the first butterfly
    }
    FHES_PASS(true, x, w);
    if (NP == 3) {
      if (PRIO != 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y_r[fswz(fft_point<LOGM, LO_B, 3>(q, e))] = x[e];
      }
      load_fwd_twiddles<LOGM, LO_C>(w, q, TF);
      __builtin_amdgcn_wave_barrier();
      if (PRIO != 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = y_r[fswz(fft_point<LOGM, LO_C, 3>(q, e))];
        __builtin_amdgcn_sched_barrier(0);  // all eight reads of the exchange go out before the first butterfly
      }
      FHES_PASS(true, x, w);
    }
    if (PRIO != 5) fft_fwd_stage0_dpp(x, cd);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e].x += cd[e].x + cd[e].y;  // keeps the loads of the constants alive
    }
    load_pass_twiddles_tab<LOGM, LO_C>(w, q, T, wnone);  // first inverse pass (LO_C == LO_B when there are two)

    if (PRIO != 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x_r[fswz(fft_point<LOGM, LO_C, 3>(q, e))] = x[e];
    }
    __syncthreads();  // B3
    if (PRIO == 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = ext_mac(x[e], x[e], g0[e], g1[e]);
    } else {
      // the other half's spectrum as ONE burst of reads (left to the compiler: read two, wait, multiply, four times over),
      // inside each form: hoisted above the branch, the two forms are merged behind register shuffles or selects
      if (r == 0) {
        double2 oth[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) oth[e] = x_o[fswz(fft_point<LOGM, LO_C, 3>(q, e))];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = ext_mac(x[e], oth[e], g0[e], g1[e]);
        asm volatile("; end of form r = 0");  // two DIFFERENT closing statements: nothing above them can be sunk into a common tail
      } else {
        double2 oth[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) oth[e] = x_o[fswz(fft_point<LOGM, LO_C, 3>(q, e))];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = ext_mac(oth[e], x[e], g0[e], g1[e]);
        asm volatile("; end of form r = 1");
      }
    }
    // vmcnt counts in order: what is needed first (the top-stage twiddles of inverse pass A) is requested first
    double2 wtop[4] = {};  // entries 2 and 3 are quarter turns of 0 and 1 (load_pass_twiddles_tab derives them)
#pragma unroll
    for (int e = 0; e < 2; ++e) wtop[e] = buf_load_d2(rs_twg, voff_q, (unsigned)(e * TP * 16));
    double2 twi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) twi[e] = buf_load_d2(rs_ti, voff_q, (unsigned)(e * TP * 16));

    if (PRIO != 5) fft_stage_bit0_dpp(x, sgn);
    if (NP == 3) {
      FHES_PASS(false, x, w);  // the pass-C twiddles requested before B3
      if (PRIO != 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y_r[fswz(fft_point<LOGM, LO_C, 3>(q, e))] = x[e];
      }
      load_pass_twiddles_tab<LOGM, LO_B>(w, q, T, wnone);
      __builtin_amdgcn_wave_barrier();  // C -> B stays inside a wave
      if (PRIO != 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = y_r[fswz(fft_point<LOGM, LO_B, 3>(q, e))];
        __builtin_amdgcn_sched_barrier(0);  // all eight reads of the exchange go out before the first butterfly
      }
      FHES_PASS(false, x, w);
    } else {
      FHES_PASS(false, x, w);  // LO_C == LO_B: requested before B3
    }
    // B -> A through Y as well (three passes: in place, the thread wrote/read these slots itself; two passes:
    // Y is free since B1).  X keeps the spectra until the next iteration: no barrier "the other half has read
    // my spectrum" (variant 4's B4) -- every thread's reads of X lie before B5.
    if (PRIO != 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) y_r[fswz(fft_point<LOGM, LO_B, 3>(q, e))] = x[e];
    }
    load_pass_twiddles_tab<LOGM, LO_A>(w, q, T, wtop);
    __syncthreads();  // B5
    if (PRIO != 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = y_r[fswz(fft_point<LOGM, LO_A, 3>(q, e))];
      __builtin_amdgcn_sched_barrier(0);  // all eight reads of the exchange go out before the first butterfly
    }
    FHES_PASS(false, x, w);

    // ---- untwist (negated factors: -y), round, nacc += rint(-y); the image goes back into Y ----
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double2 v = cmul_conj(x[e], make_double2(-twi[e].x, -twi[e].y));
      nlo[e] = add64(torus_from_f64_pair(v.x), nlo[e]);
      nhi[e] = add64(torus_from_f64_pair(v.y), nhi[e]);
      // in place: both words lie in the slot this thread has just read
      reinterpret_cast<ulonglong2 *>(y_r)[fswz(fft_point<LOGM, LO_A, 3>(q, e))] = make_ulonglong2(nlo[e], nhi[e]);
    }
    __syncthreads();  // B6
    if (PRIO == 0) __builtin_amdgcn_s_setprio(3);
  }

  // ---- sample extract from the image of -acc: a'_0 = a_0, a'_j = -a_{N-j}, b' = b_0 ----
  uint64_t *out = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
  const uint64_t *img = reinterpret_cast<const uint64_t *>(smem_raw);  // polynomial 0 (polynomial 1 follows at word 2M)
  for (int j = tid; j < N; j += blockDim.x) {
    const int jj = (j == 0) ? 0 : N - j;
    const uint64_t v = img[image_word<LOGM>(jj)];
    out[j] = (j == 0) ? (uint64_t)0 - v : v;
  }
  if (tid == 0) out[N] = (uint64_t)0 - img[(size_t)2 * M + image_word<LOGM>(0)];
}

#ifdef FHES_LAB  // variant 8 (three workgroups per CU: measured 28 % slower, profiles/r04_probe5_variant8.txt): lab only since round 5
// ------------------------------------------------------------------------------------------
// K2+K3 (variant 8, round 4): variant 5 on a diet -- THREE workgroups per CU.  The synthetic kernel with variant 5's
// instruction counts (scripts/micro/lds_valu_gate.hip, structure "K") says what a third resident workgroup is worth: 4.3-4.6 k
// cycles per CMUX and CU in steady state against 6.4 k with two.  A third workgroup needs <= 168 registers and <= 52 KB of LDS:
//   * LDS: Y (image of -acc AND the only exchange buffer, 32 KB) | inverse stage tables (8 KB) | forward twiddles (8 KB) = 48 KB.
//     Without variant 5's second buffer X two more workgroup barriers are needed: Ba "every rotated read of the image is done"
//     before pass A's output goes into Y, and B4 "the other half has read my spectrum" before the inverse exchanges overwrite
//     it -- six barriers per iteration (the synthetic: 6 457 against 6 398 cycles at two residents, 4 298 against 4 237 at three);
//   * registers: own accumulator words (32) + points (32) + GGSW rows (64) are the floor; a pass's twiddles cost four table loads
//     with the quarter-turn partners derived (16); the lane-pair constants and the untwist factors are requested where the
//     registers of the previous pass's twiddles have died.
// Arithmetic, layouts, swizzle, integer tricks: variant 5's, statement by statement (same bits).
// ------------------------------------------------------------------------------------------
// The lane-pair stage of the forward transform with HALF the constants (variant 8): the last-stage twiddles of consecutive blocks
// are quarter turns of each other, t' = (t.im, -t.re) bit for bit, so the constant of point e + 1 follows from the
// one of point e (e even): with c = (s t.re, t.im), s = +-1 by lane parity, c' = (s c.y, -s c.x), and the sign s moves onto the
// partner's value (a sign flip of the high dword on odd lanes): fma(s c.y, px, own) == fma(c.y, s px, own) -- the same product,
// the same rounding, the same bits as fft_fwd_stage0_dpp with all eight constants.
__device__ __forceinline__ void fft_fwd_stage0_dpp_half(double2 (&x)[8], const double2 (&c)[4], int sign_mask /* q odd: 0x80000000 */) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const double2 ce = c[e >> 1];
    {
      const double px = dpp_swap1(x[e].x), py = dpp_swap1(x[e].y);
      double re = __builtin_fma(ce.x, px, x[e].x);
      re = __builtin_fma(-ce.y, py, re);
      double im = __builtin_fma(ce.x, py, x[e].y);
      im = __builtin_fma(ce.y, px, im);
      x[e] = make_double2(re, im);
    }
    {
      int lo = __double2loint(x[e + 1].x), hi = __double2hiint(x[e + 1].x);
      lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
      hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true) ^ sign_mask;
      const double spx = __hiloint2double(hi, lo);
      lo = __double2loint(x[e + 1].y), hi = __double2hiint(x[e + 1].y);
      lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
      hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true) ^ sign_mask;
      const double spy = __hiloint2double(hi, lo);
      // c' = (s c.y, -s c.x):  re = fma(c'.x, px, own.x) = fma(c.y, s px, .);  re = fma(-c'.y, py, re) = fma(c.x, s py, re)
      //                        im = fma(c'.x, py, own.y) = fma(c.y, s py, .);  im = fma(c'.y, px, im) = fma(-c.x, s px, im)
      double re = __builtin_fma(ce.y, spx, x[e + 1].x);
      re = __builtin_fma(ce.x, spy, re);
      double im = __builtin_fma(ce.y, spy, x[e + 1].y);
      im = __builtin_fma(-ce.x, spx, im);
      x[e + 1] = make_double2(re, im);
    }
  }
}

typedef double __attribute__((ext_vector_type(2))) fhes_f64x2;
typedef fhes_f64x2 __attribute__((address_space(3))) *lds_f64x2_ptr;
// one complex slot at an LDS byte address (ds_read_b128 / ds_write_b128)
__device__ __forceinline__ double2 lds_get_d2(unsigned a) {
  const fhes_f64x2 v = *(lds_f64x2_ptr)(uintptr_t)a;
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ void lds_put_d2(unsigned a, double2 v) {
  fhes_f64x2 t;
  t.x = v.x;
  t.y = v.y;
  *(lds_f64x2_ptr)(uintptr_t)a = t;
}

// ABL: timing-only ablations (WRONG results): 1 = the own words neither stored nor reloaded, 2 = also no key loads
template <int LOGN, int BL, int ABL = 0>
__global__ __launch_bounds__(((1 << LOGN) / 8) < 64 ? 64 : ((1 << LOGN) / 8), 3) void blind_rotate_v8_kernel(
    NodeDesc d, uint64_t *__restrict__ arena, int ct_stride, const uint64_t *__restrict__ ks_out,
    int ks_stride, int n_small, const double2 *__restrict__ bsk_e, const uint64_t *__restrict__ lut_polys,
    const double2 *__restrict__ twist_inv /* twist / (M * 2^64): torus_from_f64_pair */, const double2 *__restrict__ tw,
    const double2 *__restrict__ psi /* forward twiddles [M] */, const double2 *__restrict__ cdpp /* [8][M/8] */,
    ulonglong2 *__restrict__ own_scratch /* [workgroups][8][256]: the threads' own accumulator words between two updates */) {
  constexpr int N = 1 << LOGN, M = N / 2, LOGM = LOGN - 1, TP = M / 8;
  constexpr int NP = (LOGM - 1) / 3;
  static_assert(LOGM % 3 == 1 && NP == 3, "written for N = 2048 (three 3-bit passes + the lane-pair stage)");
  static_assert(BL >= 1 && BL <= 31, "single-level digit must fit the high dword");
  static_assert(TP >= 128, "a wave lies inside one half and the swizzle term is common to a thread's points");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double2 *ybuf = reinterpret_cast<double2 *>(smem_raw);  // [2][M]  image of -acc and exchange buffer
  double2 *T = ybuf + (size_t)2 * M;                      // [M/2]   inverse stage tables
  double2 *TF = T + M / 2;                                // [M/2]   forward twiddles psi[0, M/2)

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw;
  if (lds0 & (unsigned)(2 * N * 8 - 1)) __builtin_trap();  // the rotated reads OR the block's address into their offsets
  const int tid = threadIdx.x;
  const int r = __builtin_amdgcn_readfirstlane(tid / TP);  // the half (output polynomial): wave-uniform
  const int q = tid % TP;
  const double sgn = (q & 1) ? -1.0 : 1.0;
  const int vidx = d.v0 + blockIdx.x;
  const int inst = vidx / d.count, node = d.first + vidx % d.count;

  const uint64_t *ks = ks_out + (size_t)blockIdx.x * ks_stride;
  auto modswitch_of = [&](uint64_t w) __attribute__((always_inline)) {
    return (int)((((w >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
  };
  for (int idx = 2 + tid; idx < M / 2; idx += blockDim.x) {
    const int beta = 31 - __builtin_clz(idx);  // idx = 2^beta + j
    T[idx] = tw[(idx - (1 << beta)) << (LOGM - 1 - beta)];
  }
  for (int idx = tid; idx < M / 2; idx += blockDim.x) TF[idx] = psi[idx];
  __syncthreads();

  constexpr int LO_A = LOGM - 3, LO_B = LOGM - 6, LO_C = LOGM - 9;
  double2 wfa[7];  // forward pass A: the same 7 twiddles for every thread (SGPRs)
#pragma unroll
  for (int kk = 0; kk < 7; ++kk) wfa[kk] = psi[1 + kk];
  const double2 wnone[4] = {};
  const __amdgpu_buffer_rsrc_t rs_twg = __builtin_amdgcn_make_buffer_rsrc((void *)tw, 0, (M / 2) * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g =
      __builtin_amdgcn_make_buffer_rsrc((void *)bsk_e, 0, (unsigned)((size_t)n_small * 4 * M * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cd = __builtin_amdgcn_make_buffer_rsrc((void *)cdpp, 0, M * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ti = __builtin_amdgcn_make_buffer_rsrc((void *)twist_inv, 0, M * 16, 0x00020000);
  const unsigned voff_q = (unsigned)q * 16u;
  const unsigned voff_g = (unsigned)(r * 8 * TP + q) * 16u;
  // The own accumulator words (32 registers) are needed twice per iteration only -- when the digits are formed and when the
  // update is added -- and the LDS image is overwritten in between.  They wait in a per-workgroup global scratch (L2-resident:
  // 32 KB per workgroup, written once and read once per iteration, coalesced) instead of in registers the third workgroup
  // does not have: stored at the update, kept in registers across B6 for the next iteration's digits, dropped, and requested
  // again before B5.
  const __amdgpu_buffer_rsrc_t rs_own = __builtin_amdgcn_make_buffer_rsrc((void *)(own_scratch + (size_t)blockIdx.x * 8 * 256), 0, 8 * 256 * 16, 0x00020000);
  const unsigned voff_t = (unsigned)tid * 16u;

  // nacc: NEGATED coefficients p = q + e*TP (lo) and p + M (hi) of polynomial r
  uint64_t nlo[8], nhi[8];
  {
    const uint64_t *lut = lut_polys + (size_t)d.lut_ids[node] * N;
    const int t = (2 * N - modswitch_of(ks[n_small])) & (2 * N - 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = fft_point<LOGM, LO_A, 3>(q, e);
      uint64_t v0 = 0, v1 = 0;
      if (r == 1) {
        v0 = (t == 0) ? lut[p] : rot_read<N>(lut, t, p);
        v1 = (t == 0) ? lut[p + M] : rot_read<N>(lut, t, p + M);
      }
      nlo[e] = (uint64_t)0 - v0;
      nhi[e] = (uint64_t)0 - v1;
      reinterpret_cast<uint64_t *>(ybuf + (size_t)r * M)[image_word<LOGM>(p)] = nlo[e];
      reinterpret_cast<uint64_t *>(ybuf + (size_t)r * M)[image_word<LOGM>(p + M)] = nhi[e];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, make_ulonglong2(nlo[e], nhi[e])),
                                             rs_own, voff_t, (unsigned)(e * 256 * 16), 0);
    }
  }
  __syncthreads();

  // Byte addresses of this thread's slots in the three pass layouts, fswz spelled out.  Pass A: one base, the points 2 KB apart
  // (immediate offsets).  Passes B and C: the swizzle XORs the register index e into bank bits that hold thread bits, so the eight
  // addresses are base ^ constant(e) -- the compiler keeps all of them in registers across the loop (25 with the other half's),
  // which the third workgroup cannot afford.  The bases are made opaque before every use: one v_xor per access instead.
  // (Y starts at LDS offset 0 and a half is 16 KB: the XORs below stay inside the half, and bit 14 selects it.)
  const unsigned yr0 = lds0 + (unsigned)r * (unsigned)(M * 16);
  const unsigned aA = yr0 + 16u * (unsigned)fswz(q);
  unsigned aB = yr0 + 16u * (unsigned)(((q >> 4) << 7) | (q & 15));
  unsigned aC = yr0 + 16u * (unsigned)(((q >> 1) << 4) | (((q >> 1) & 7) << 1) | (q & 1));
  auto fresh = [](int v) { asm volatile("" : "+v"(v)); return v; };  // a copy the compiler cannot hoist address arithmetic from
  constexpr uint64_t ROUND = (uint64_t)1 << (63 - BL);

  int a_next = modswitch_of(ks[0]);
  for (int i = 0; i < n_small; ++i) {
    const int a = a_next;
    double2 x[8];
    {
      const unsigned sb = ((unsigned)(q - a) << 4) ^ (1u << (LOGM + 5));
      const unsigned img_base = (unsigned)r * (unsigned)(M * 16) + lds0;
      const unsigned swz_all = ((sb >> 3) & 0xE0u) | img_base;
      uint64_t va[8], vb[8];
      int hmv[8], m0v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned se = sb + (unsigned)(e * TP * 16);
        hmv[e] = __builtin_amdgcn_sbfe(se, LOGM + 4, 1);
        m0v[e] = __builtin_amdgcn_sbfe(se, LOGM + 5, 1);
        const unsigned off0 = ((se & (unsigned)((M - 1) << 4)) ^ swz_all) | ((unsigned)hmv[e] & 8u);
        va[e] = *(lds_u64_ptr)(uintptr_t)(off0);
        vb[e] = *(lds_u64_ptr)(uintptr_t)(off0 ^ 8u);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int m0 = m0v[e];
        const int m1 = m0 ^ hmv[e];
        uint64_t dd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m32 = h == 0 ? m0 : m1;
          const uint64_t v = h == 0 ? va[e] : vb[e];
          const uint64_t mm = ((uint64_t)(uint32_t)m32 << 32) | (uint64_t)(uint32_t)m32;
          const uint64_t kk = ROUND | (uint64_t)(uint32_t)(m32 & 1);
          dd[h] = add64(add64(v ^ mm, h == 0 ? nlo[e] : nhi[e]), kk);
        }
        const double d0 = (double)((int)(uint32_t)(dd[0] >> 32) >> (32 - BL));
        const double d1 = (double)((int)(uint32_t)(dd[1] >> 32) >> (32 - BL));
        x[e] = make_double2(d0, d1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t ks_next = ks[i + 1];
    // the key's two rows in two requests: the second goes out after the first pass, whose butterflies need the registers
    double2 g0[8], g1[8];
    const unsigned so = (unsigned)i * (unsigned)(4 * M * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) g0[e] = ABL < 2 ? buf_load_d2(rs_g, voff_g, so + (unsigned)((0 * 16 + e) * TP * 16)) : make_double2(x[e].y, wfa[e & 3].x);
    double2 w[7];
    fft_butterflies_reg<true>(x, wfa);
    a_next = modswitch_of(ks_next);
    asm volatile("" : "+s"(a_next));
    __syncthreads();  // Ba: every rotated read of the image is done -- Y may be overwritten
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 8; ++e) g1[e] = ABL < 2 ? buf_load_d2(rs_g, voff_g, so + (unsigned)((1 * 16 + e) * TP * 16)) : make_double2(x[e].x, wfa[e & 3].y);
#pragma unroll
    for (int e = 0; e < 8; ++e) lds_put_d2(aA + (unsigned)(e * TP * 16), x[e]);
    load_fwd_twiddles<LOGM, LO_B>(w, fresh(q), TF);
    __syncthreads();  // B1
    asm volatile("" : "+v"(aB));
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = lds_get_d2(aB ^ (unsigned)(e * 0x120));
    __builtin_amdgcn_sched_barrier(0);
    fft_butterflies_reg<true>(x, w);
    asm volatile("" : "+v"(aB));
#pragma unroll
    for (int e = 0; e < 8; ++e) lds_put_d2(aB ^ (unsigned)(e * 0x120), x[e]);  // B -> C stays inside a wave
    load_fwd_twiddles<LOGM, LO_C>(w, fresh(q), TF);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" : "+v"(aC));
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = lds_get_d2(aC ^ (unsigned)(e << 5));
    __builtin_amdgcn_sched_barrier(0);
    fft_butterflies_reg<true>(x, w);
    {
      // the lane-pair constants take the registers the pass twiddles have just left: four loads, the odd points' constants derived
      double2 cd[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) cd[e] = buf_load_d2(rs_cd, voff_q, (unsigned)(2 * e * TP * 16));
      fft_fwd_stage0_dpp_half(x, cd, (q & 1) << 31);
    }
    // the spectrum goes into the slots of the pass-C layout: this wave's own region of Y
    asm volatile("" : "+v"(aC));
#pragma unroll
    for (int e = 0; e < 8; ++e) lds_put_d2(aC ^ (unsigned)(e << 5), x[e]);
    __syncthreads();  // B3
    asm volatile("" : "+v"(aC));
    // the other half's spectrum in two bursts of four reads (variant 5: one of eight; sixteen registers fewer at the kernel's peak)
    if (r == 0) {
#pragma unroll
      for (int hb = 0; hb < 8; hb += 4) {
        double2 oth[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) oth[e] = lds_get_d2(aC ^ (unsigned)((M * 16) | ((hb + e) << 5)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[hb + e] = ext_mac(x[hb + e], oth[e], g0[hb + e], g1[hb + e]);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("; v8: end of form r = 0");
    } else {
#pragma unroll
      for (int hb = 0; hb < 8; hb += 4) {
        double2 oth[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) oth[e] = lds_get_d2(aC ^ (unsigned)((M * 16) | ((hb + e) << 5)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[hb + e] = ext_mac(oth[e], x[hb + e], g0[hb + e], g1[hb + e]);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("; v8: end of form r = 1");
    }
    // the first inverse pass's twiddles only now (variant 5: before B3): sixteen registers fewer under the products, the kernel's peak
    load_pass_twiddles_tab<LOGM, LO_C>(w, fresh(q), T, wnone);
    fft_stage_bit0_dpp(x, sgn);
    fft_butterflies_reg<false>(x, w);  // inverse pass C
    __syncthreads();  // B4: the other half has read my spectrum -- my region of Y may be overwritten
    asm volatile("" : "+v"(aC));
#pragma unroll
    for (int e = 0; e < 8; ++e) lds_put_d2(aC ^ (unsigned)(e << 5), x[e]);  // C -> B stays inside a wave
    load_pass_twiddles_tab<LOGM, LO_B>(w, fresh(q), T, wnone);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" : "+v"(aB));
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = lds_get_d2(aB ^ (unsigned)(e * 0x120));
    __builtin_amdgcn_sched_barrier(0);
    fft_butterflies_reg<false>(x, w);
    asm volatile("" : "+v"(aB));
#pragma unroll
    for (int e = 0; e < 8; ++e) lds_put_d2(aB ^ (unsigned)(e * 0x120), x[e]);
    double2 wtop[4] = {};
#pragma unroll
    for (int e = 0; e < 2; ++e) wtop[e] = buf_load_d2(rs_twg, voff_q, (unsigned)(e * TP * 16));
    load_pass_twiddles_tab<LOGM, LO_A>(w, fresh(q), T, wtop);
    double2 twi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) twi[e] = buf_load_d2(rs_ti, voff_q, (unsigned)(e * TP * 16));
    double2 own[8];  // the own accumulator words as stored at the previous update (bit patterns)
#pragma unroll
    for (int e = 0; e < 8; ++e) own[e] = ABL == 0 ? buf_load_d2(rs_own, voff_t, (unsigned)(e * 256 * 16)) : twi[e];
    __syncthreads();  // B5
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = lds_get_d2(aA + (unsigned)(e * TP * 16));
    __builtin_amdgcn_sched_barrier(0);
    fft_butterflies_reg<false>(x, w);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double2 v = cmul_conj(x[e], make_double2(-twi[e].x, -twi[e].y));
      nlo[e] = add64(torus_from_f64_pair(v.x), (uint64_t)__double_as_longlong(own[e].x));
      nhi[e] = add64(torus_from_f64_pair(v.y), (uint64_t)__double_as_longlong(own[e].y));
      lds_put_d2(aA + (unsigned)(e * TP * 16), make_double2(__longlong_as_double((long long)nlo[e]), __longlong_as_double((long long)nhi[e])));
      if (ABL == 0)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, make_ulonglong2(nlo[e], nhi[e])),
                                               rs_own, voff_t, (unsigned)(e * 256 * 16), 0);
    }
    __syncthreads();  // B6 (it also waits for the stores above: the words are in memory before they are requested again)
  }

  // ---- sample extract from the image of -acc: a'_0 = a_0, a'_j = -a_{N-j}, b' = b_0 ----
  uint64_t *out = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
  const uint64_t *img = reinterpret_cast<const uint64_t *>(smem_raw);
  for (int j = tid; j < N; j += blockDim.x) {
    const int jj = (j == 0) ? 0 : N - j;
    const uint64_t v = img[image_word<LOGM>(jj)];
    out[j] = (j == 0) ? (uint64_t)0 - v : v;
  }
  if (tid == 0) out[N] = (uint64_t)0 - img[(size_t)2 * M + image_word<LOGM>(0)];
}
#endif  // FHES_LAB

// ------------------------------------------------------------------------------------------
// K4: multi-bit blind rotate (grouping factor G = 2 or 3) + sample extract.  Same transforms,
// rounding and FMA chains as the classic kernels, different step:
//     bundle = G_0 + sum_{c=1}^{2^G-1} G_c * X^{deg_c}      (Fourier domain, c ascending)
//     acc   <- bundle (x) acc                                (digits of acc itself; replaced)
// with deg_c = modswitch(sum_t bit_{G-1-t}(c) * a_{jG+t}) -- the u64 mask elements are summed
// first, then switched.  The accumulator never needs an LDS image (no rotation inside the loop):
// it lives in registers; LDS holds only the FFT exchange buffer, the stage twiddles and the degrees.
// Mapping as variant 3: 256 threads = two halves (output polynomial r) x 128 threads x 8 points,
// two workgroups per CU.  Every exchange is written in place (a thread writes the slots it read),
// so an iteration has 6 barriers and none at its end.
// The Fourier image of X^d at spectrum slot p is exp(i pi E(p) d / N), E(p) = 1 - 4*bitrev(p) (slot p holds the evaluation at
// twist * exp(-2*pi*i*bitrev(p)/M)); since round 5 it is taken as the PRODUCT of two entries of root[t] = exp(i*pi*t/N) whose
// exponents split E(p) d along fixed bits of bitrev(p) (oracle_blind_rotate_multibit): one gather per monomial
// and thread plus scalar loads, where the single-entry form cost eight gathers.
// LDS: fbuf double2[2][M] | stage twiddle tables double2[M/2] | degrees u16[(n/G) * (2^G - 1)]
// ------------------------------------------------------------------------------------------
// acc += g * w with the operation order of cmul
__device__ __forceinline__ double2 cfma(double2 g, double2 w, double2 acc) {
  double re = __builtin_fma(g.x, w.x, acc.x);
  re = __builtin_fma(-g.y, w.y, re);
  double im = __builtin_fma(g.x, w.y, acc.y);
  im = __builtin_fma(g.y, w.x, im);
  return make_double2(re, im);
}

// INVERSE butterflies of a 3-bit pass, every stage's twiddles read from the LDS stage tables (T[M] entries) at the point
// of use (the pipelined multi-bit build: no registers to hold them across a barrier)
template <int LOGM, int LO>
__device__ __forceinline__ void fft_inv_butterflies_tab_all(double2 (&x)[8], int q, const double2 *T) {
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    const int fb = st;
    const int beta = LO + fb;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e & (1 << fb)) continue;
      const int e1 = e | (1 << fb);
      inv_butterfly(x[e], x[e1], T[(1 << beta) + (fft_point<LOGM, LO, 3>(q, e) & ((1 << beta) - 1))]);
    }
  }
}
// the 7 twiddles of an INVERSE 3-bit pass from the LDS stage tables with M entries (every stage; the order of
// fft_butterflies_reg<false>): requested ahead of the barrier in front of the pass, like variant 5 does
template <int LOGM, int LO>
__device__ __forceinline__ void load_inv_twiddles_tab_all(double2 (&w)[7], int q, const double2 *T) {
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int fb = k < 4 ? 2 : (k < 6 ? 1 : 0);
    const int e = k < 4 ? k : (k < 6 ? (k - 4) : 0);
    const int beta = LO + fb;
    w[k] = T[(1 << beta) + (fft_point<LOGM, LO, 3>(q, e) & ((1 << beta) - 1))];
  }
}

// LAT = true: the latency build -- one workgroup per CU (512 registers per lane), pass twiddles and
// twist factors in registers (as variant 2), always with the bundle pipeline.
// ILV (round-4 layout A/B, g = 3 throughput build only): the 2^G GGSWs of a group interleaved row by row in the key
// SCAL: the workgroup-uniform monomial factors through the scalar cache (launches of at most one workgroup per CU: 2.34 against
// 2.66 ms per g = 3 bootstrap) instead of as one-address vector loads (wide launches: 157 against 127 k PBS/s for g = 2)
template <int LOGN, int BL, int G, bool PIPE, bool LAT, bool ILV = false, bool SCAL = false>
__global__ __launch_bounds__(((1 << LOGN) / 8) < 64 ? 64 : ((1 << LOGN) / 8), LAT ? 1 : 2) void blind_rotate_mb_kernel(
    NodeDesc d, uint64_t *__restrict__ arena, int ct_stride, const uint64_t *__restrict__ ks_out,
    int ks_stride, int n_small, const double2 *__restrict__ bsk_e, const uint64_t *__restrict__ lut_polys,
    const double2 *__restrict__ twist_inv /* twist / M */, const double2 *__restrict__ tw,
    const double2 *__restrict__ root /* [2N] */, const double2 *__restrict__ psi /* forward twiddles [M] */,
    const double2 *__restrict__ cdpp /* [8][M/8] */
#ifdef FHES_LAB   // timing-only probes (WRONG results unless both are -1): the monomial factors from a table cut down to
                  // `lab_root_mask + 1` entries, every step reading the key of group `j AND lab_group_mask` -- what would the
                  // gathers / the key stream cost if they hit close by?
    , int lab_root_mask, int lab_group_mask
#endif
    ) {
#ifndef FHES_LAB
  constexpr int lab_root_mask = -1, lab_group_mask = -1;
#endif
  constexpr bool ilv = ILV;
  constexpr int N = 1 << LOGN, M = N / 2, LOGM = LOGN - 1, TP = M / 8;
  constexpr int NP = (LOGM - 1) / 3, NC = 1 << G;
  static_assert(LOGM % 3 == 1 && (NP == 2 || NP == 3), "supported: N = 256, N = 2048");
  constexpr int SLOTS = (M >= 128 ? M / 128 : 1) * 142;  // padded slots per polynomial (fpad)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double2 *fbuf = reinterpret_cast<double2 *>(smem_raw);                                 // [2][SLOTS]
  double2 *T = reinterpret_cast<double2 *>(smem_raw + (size_t)2 * SLOTS * 16);           // [M] inverse stage tables
  double2 *TF = T + M;                                                                   // [M/2] forward twiddles psi[0, M/2)
  double2 *CDL = TF + M / 2;                                                             // [8][M/8] constants of the lane-pair stage
  uint16_t *degs = reinterpret_cast<uint16_t *>(CDL + M);

  const int tid = threadIdx.x;
  const int r = tid / TP;
  const int q = tid % TP;
  const double sgn = (q & 1) ? -1.0 : 1.0;
  const int vidx = d.v0 + blockIdx.x;
  const int inst = vidx / d.count, node = d.first + vidx % d.count;
  const int n_groups = n_small / G;

  const uint64_t *ks = ks_out + (size_t)blockIdx.x * ks_stride;
  for (int idx = tid; idx < n_groups * (NC - 1); idx += blockDim.x) {
    const int j = idx / (NC - 1), c = idx % (NC - 1) + 1;
    uint64_t sum = 0;
#pragma unroll
    for (int t = 0; t < G; ++t)
      if ((c >> (G - 1 - t)) & 1) sum += ks[j * G + t];
    degs[idx] = (uint16_t)((((sum >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
  }
  for (int idx = 2 + tid; idx < M; idx += blockDim.x) {
    const int beta = 31 - __builtin_clz(idx);  // idx = 2^beta + j
    T[idx] = tw[(idx - (1 << beta)) << (LOGM - 1 - beta)];
  }
  for (int idx = tid; idx < M / 2; idx += blockDim.x) TF[idx] = psi[idx];
  for (int idx = tid; idx < M; idx += blockDim.x) CDL[idx] = cdpp[idx];
  __syncthreads();

  constexpr int LO_A = LOGM - 3, LO_B = LOGM - 6, LO_C = NP == 3 ? LOGM - 9 : LOGM - 6;
  const double2 *cdpp_q = cdpp + q;
  const double2 *twist_inv_q = twist_inv + q;
  double2 wfa[7];  // forward pass A: the same 7 twiddles for every thread
#pragma unroll
  for (int kk = 0; kk < 7; ++kk) wfa[kk] = psi[1 + kk];
  // latency build only (dead otherwise): the inverse twiddles, the lane-pair constants and the untwist factors
  // stay in registers; the forward twiddles come from LDS in both builds (resident they pushed the latency
  // build to 502 registers, of which only 256 are VALU-addressable: 266 v_accvgpr_read per kernel)
  double2 wA[7], wB[7], wC[7], cdr[8], twir[8];
  if (LAT) {
    load_pass_twiddles<LOGM, LO_A>(wA, q, tw);
    load_pass_twiddles<LOGM, LO_B>(wB, q, tw);
    if (NP == 3) load_pass_twiddles<LOGM, LO_C>(wC, q, tw);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cdr[e] = cdpp_q[e * TP];
      twir[e] = twist_inv_q[e * TP];
    }
  }
  const __amdgpu_buffer_rsrc_t rs_g =
      __builtin_amdgcn_make_buffer_rsrc((void *)bsk_e, 0, (unsigned)((size_t)n_groups * NC * 4 * M * 16), 0x00020000);
  // G[((row*2 + r)*8 + e)*TP + q] inside GGSW c of a group; interleaved: rows of 2 KB ordered (row, r, e, c)
  const unsigned voff_g = ilv ? (unsigned)(r * 8 * NC * TP + q) * 16u : (unsigned)(r * 8 * TP + q) * 16u;
  auto goff = [&](unsigned so, int c, int row, int e) __attribute__((always_inline)) {
    return ilv ? so + (unsigned)(((row * 16 + e) * NC + c) * TP * 16) : so + (unsigned)c * (unsigned)(4 * M * 16) + (unsigned)((row * 16 + e) * TP * 16);
  };
  // E(p) of the slots this thread holds when the external product runs: p = point_C(q, e) has e in
  // bits [LO_C, LO_C+3), so bitrev(p) = bitrev(point_C(q, 0)) + (bitrev3(e) << (LOGM - 3 - LO_C)) and
  // E(p) * d = E(point_C(q, 0)) * d - bitrev3(e) * (d << (LOGM - 1 - LO_C))   (mod 2N)
  const int sexp0 = (1 - 4 * (int)(__brev((unsigned)fft_point<LOGM, LO_C, 3>(q, 0)) >> (32 - LOGM))) & (2 * N - 1);

  // ---- acc = (0, X^{-b~} * LUT), in registers: coefficients p and p + M of polynomial r ----
  uint64_t own_lo[8], own_hi[8];
  {
    const uint64_t *lut = lut_polys + (size_t)d.lut_ids[node] * N;
    const int bt = (int)((((ks[n_small] >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
    const int t = (2 * N - bt) & (2 * N - 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = fft_point<LOGM, LO_A, 3>(q, e);
      uint64_t v0 = 0, v1 = 0;
      if (r == 1) {
        v0 = (t == 0) ? lut[p] : rot_read<N>(lut, t, p);
        v1 = (t == 0) ? lut[p + M] : rot_read<N>(lut, t, p + M);
      }
      own_lo[e] = v0;
      own_hi[e] = v1;
    }
  }

  // one base slot per pass layout; element e lives at base + e * stride (immediate offsets)
  constexpr int SA = fpad(1 << LO_A), SB = fpad(1 << LO_B), SC = fpad(1 << LO_C);
  double2 *bufA = fbuf + (size_t)r * SLOTS + fpad(fft_point<LOGM, LO_A, 3>(q, 0));
  double2 *bufB = fbuf + (size_t)r * SLOTS + fpad(fft_point<LOGM, LO_B, 3>(q, 0));
  double2 *bufC = fbuf + (size_t)r * SLOTS + fpad(fft_point<LOGM, LO_C, 3>(q, 0));
  const double2 *bufCo = fbuf + (size_t)(1 - r) * SLOTS + fpad(fft_point<LOGM, LO_C, 3>(q, 0));

  // ---- bundle pipeline.  The bundle of group j+1 (column r: b0 = row 0, b1 = row 1) is built while
  // group j runs its inverse FFT: b0/b1 are dead after the external product, and the accumulator
  // registers are dead between the digits and the rounding, so the loads of a batch (two points for
  // G = 2, one point and half of c for G = 3) stay in flight across a whole FFT pass.  A batch is
  // issued, and consumed at the next slot; the last batch is consumed right before the product. ----
  constexpr int EB = G == 2 ? 2 : 1, CB = G == 2 ? NC - 1 : 4, NCB = (NC - 1 + CB - 1) / CB;
  // PIPE (chosen by the launcher): build the next bundle inside the inverse FFT.  Measured: pays for
  // G = 3 always (72 -> 84 k PBS/s) and for G = 2 when a launch leaves one workgroup per CU (2.83 vs
  // 3.03 ms per PBS); in wide G = 2 launches the key stream is the limit either way and the plain
  // order keeps the kernel spill-free (122 vs 105 k PBS/s)
  constexpr int NBATCH = (8 / EB) * NCB, QB = NBATCH / 4, NEB = 8 / EB;
  double2 b0[8], b1[8];
  double2 tg0[EB][CB], tg1[EB][CB];
  // Monomial factors: W_c[p] = root[deg_c (1 - 4 b_S)] * root[-4 deg_c b_L] with b = bitrev(p) split
  // into b_S (the three bits a thread's eight points differ in: the same for the whole workgroup -> a SCALAR load per point
  // and monomial) and b_L (the thread's own: ONE gather per monomial, `rw`, where the single-entry form root[E(p) deg_c] took
  // eight).  The gathers were what bound these kernels: 64 lanes x 64 lines of a 64 KB table, two thirds of them L2 requests
  // (profiles/r05_probe8_multibit_bounds.txt: g = 3 84 -> 153 k PBS/s with the gathers hitting close by).
  double2 rw[CB];   // batches run c-range by c-range (all points of c in [cb, cb + CB) before the next range): CB factors live
  int deg[NC];
  const int lexp = (sexp0 - 1) & (2 * N - 1);   // -4 b_L of this thread (e = 0: the e bits of bitrev(p) are b_S)
  auto bundle_degrees = [&](int jn) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 1; c < NC; ++c) deg[c] = __builtin_amdgcn_readfirstlane((int)degs[jn * (NC - 1) + (c - 1)]);  // uniform -> SGPR
  };
  // the workgroup-uniform factor of monomial c at register e: root[deg_c (1 - 4 (bitrev3(e) << (LOGM - 4)))]
  // (requested as a VECTOR load of one address -- a single line for the texture unit, counted by vmcnt like the key rows it
  // travels with: as a scalar load it drained the LDS counter at every use, 127 instead of 134 k PBS/s for g = 2.  `zlane` is a
  // zero the compiler cannot see through.)
  int zlane = 0;
  if (!SCAL) asm volatile("" : "+v"(zlane));
  auto mono_scalar = [&](int c, int e) __attribute__((always_inline)) {
    const int br3 = ((e & 1) << 2) | (e & 2) | ((e >> 2) & 1);  // constant after unrolling
    return root[((deg[c] - br3 * (deg[c] << (LOGM - 1 - LO_C))) & (2 * N - 1)) + zlane];
  };
  auto bundle_issue = [&](int jn, int k) __attribute__((always_inline)) {
    const int eb = (k % NEB) * EB, cb = 1 + (k / NEB) * CB;
    const unsigned so = (unsigned)(jn & lab_group_mask) * (unsigned)(NC * 4 * M * 16);
    if (k % NEB == 0) {   // first batch of a c-range: its lane factors (the previous range's last batch has been consumed)
#pragma unroll
      for (int cc = 0; cc < CB; ++cc)
        if (cb + cc < NC) rw[cc] = root[(lexp * deg[cb + cc]) & (2 * N - 1) & lab_root_mask];
    }
#pragma unroll
    for (int ee = 0; ee < EB; ++ee) {
      const int e = eb + ee;
      if (cb == 1) {
        b0[e] = buf_load_d2(rs_g, voff_g, goff(so, 0, 0, e));
        b1[e] = buf_load_d2(rs_g, voff_g, goff(so, 0, 1, e));
      }
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) {
        const int c = cb + cc;
        if (c >= NC) continue;
        tg0[ee][cc] = buf_load_d2(rs_g, voff_g, goff(so, c, 0, e));
        tg1[ee][cc] = buf_load_d2(rs_g, voff_g, goff(so, c, 1, e));
      }
    }
  };
  auto bundle_consume = [&](int k) __attribute__((always_inline)) {
    const int eb = (k % NEB) * EB, cb = 1 + (k / NEB) * CB;
#pragma unroll
    for (int ee = 0; ee < EB; ++ee) {
      const int e = eb + ee;
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) {
        if (cb + cc >= NC) continue;
        const double2 w = cmul(mono_scalar(cb + cc, e), rw[cc]);
        b0[e] = cfma(tg0[ee][cc], w, b0[e]);
        b1[e] = cfma(tg1[ee][cc], w, b1[e]);
      }
    }
  };
  // batches [first, first + count) of group jn; on entry batch first-1 is in flight (if first > 0),
  // on exit batch first+count-1 is
  auto bundle_advance = [&](int jn, int first, int count) __attribute__((always_inline)) {
#pragma unroll
    for (int k = first; k < first + count; ++k) {
      __builtin_amdgcn_sched_barrier(0);
      if (k > 0) bundle_consume(k - 1);
      bundle_issue(jn, k);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (PIPE) {
    bundle_degrees(0);
    bundle_advance(0, 0, NBATCH);
  }

  for (int j = 0; j < n_groups; ++j) {
    const int jn = j + 1 < n_groups ? j + 1 : j;  // the last step rebuilds its own bundle (unused)
    if (!PIPE) {  // plain order: two points at a time, loads and multiply-adds of a batch together
      bundle_degrees(j);
      const unsigned so = (unsigned)(j & lab_group_mask) * (unsigned)(NC * 4 * M * 16);
      double2 rall[NC];   // the lane factors of all monomials of the group: one gather each
#pragma unroll
      for (int c = 1; c < NC; ++c) rall[c] = root[(lexp * deg[c]) & (2 * N - 1) & lab_root_mask];
#pragma unroll
      for (int eb = 0; eb < 8; eb += 2) {
#pragma unroll
        for (int e = eb; e < eb + 2; ++e) {
          b0[e] = buf_load_d2(rs_g, voff_g, goff(so, 0, 0, e));
          b1[e] = buf_load_d2(rs_g, voff_g, goff(so, 0, 1, e));
#pragma unroll
          for (int c = 1; c < NC; ++c) {
            const double2 w = cmul(mono_scalar(c, e), rall[c]);
            const double2 g0 = buf_load_d2(rs_g, voff_g, goff(so, c, 0, e));
            const double2 g1 = buf_load_d2(rs_g, voff_g, goff(so, c, 1, e));
            b0[e] = cfma(g0, w, b0[e]);
            b1[e] = cfma(g1, w, b1[e]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- digits of the accumulator (no twist: it lives in the forward twiddles) ----
    double2 x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = make_double2(digit1_f64<BL>(own_lo[e]), digit1_f64<BL>(own_hi[e]));

    // ---- forward FFT ----  (round 3, EARLY: the twiddles of a pass are requested ahead of the barrier in front of it and
    // the reads of an exchange go out as one burst before the first butterfly -- the latency build only: 2.50 against
    // 2.52 ms per bootstrap.  The throughput builds keep "reads, twiddles at the point of use, butterflies" and the
    // compiler's interleaving: g = 2 measured 125.5 against 121.2 ms per 16 384 with EARLY, g = 3 has no registers for it)
    constexpr bool EARLY = LAT;
    double2 w[7];
    fft_butterflies_reg<true>(x, wfa);
#pragma unroll
    for (int e = 0; e < 8; ++e) bufA[e * SA] = x[e];
    if (EARLY) load_fwd_twiddles<LOGM, LO_B, false>(w, q, TF);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = bufB[e * SB];
    if (EARLY) __builtin_amdgcn_sched_barrier(0); else load_fwd_twiddles<LOGM, LO_B, false>(w, q, TF);
    fft_butterflies_reg<true>(x, w);
    if (NP == 3) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bufB[e * SB] = x[e];
      if (EARLY) load_fwd_twiddles<LOGM, LO_C, false>(w, q, TF);
      __builtin_amdgcn_wave_barrier();  // B -> C stays inside a wave (q >> 6 is index bit LOGM-1 in both layouts)
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = bufC[e * SC];
      if (EARLY) __builtin_amdgcn_sched_barrier(0); else load_fwd_twiddles<LOGM, LO_C, false>(w, q, TF);
      fft_butterflies_reg<true>(x, w);
    }
    {
      double2 cd[8];  // constants of the lane-pair stage (latency build: registers; otherwise the LDS copy)
#pragma unroll
      for (int e = 0; e < 8; ++e) cd[e] = LAT ? cdr[e] : CDL[e * TP + q];
      fft_fwd_stage0_dpp(x, cd);
    }

    // ---- spectra exchanged in place, external product with the bundle ----
#pragma unroll
    for (int e = 0; e < 8; ++e) bufC[e * SC] = x[e];
    __builtin_amdgcn_sched_barrier(0);
    if (PIPE) bundle_consume(NBATCH - 1);  // last batch of this group's bundle
    if (EARLY && !LAT) load_inv_twiddles_tab_all<LOGM, LO_C>(w, q, T);  // first inverse pass (LO_C == LO_B when there are two)
    __syncthreads();
    if (!EARLY) {
      if (r == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = ext_mac(x[e], bufCo[e * SC], b0[e], b1[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = ext_mac(bufCo[e * SC], x[e], b0[e], b1[e]);
      }
    } else if (r == 0) {
      double2 oth[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) oth[e] = bufCo[e * SC];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = ext_mac(x[e], oth[e], b0[e], b1[e]);
      asm volatile("; end of form r = 0");  // two DIFFERENT closing statements: the forms are not merged behind register shuffles
    } else {
      double2 oth[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) oth[e] = bufCo[e * SC];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = ext_mac(oth[e], x[e], b0[e], b1[e]);
      asm volatile("; end of form r = 1");
    }
    double2 twi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) twi[e] = LAT ? twir[e] : twist_inv_q[e * TP];
    if (PIPE) {
      bundle_degrees(jn);
      bundle_advance(jn, 0, QB);
    }
    __syncthreads();  // the other half has read my spectrum

    // ---- inverse FFT ----
    fft_stage_bit0_dpp(x, sgn);
    if (NP == 3) {
      if (!EARLY) fft_inv_butterflies_tab_all<LOGM, LO_C>(x, q, T); else fft_butterflies_reg<false>(x, LAT ? wC : w);
#pragma unroll
      for (int e = 0; e < 8; ++e) bufC[e * SC] = x[e];
      if (EARLY && !LAT) load_inv_twiddles_tab_all<LOGM, LO_B>(w, q, T);
      if (PIPE) bundle_advance(jn, QB, QB);
      __builtin_amdgcn_wave_barrier();  // C -> B stays inside a wave
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = bufB[e * SB];
      if (EARLY) __builtin_amdgcn_sched_barrier(0);
    } else {
      if (PIPE) bundle_advance(jn, QB, QB);
    }
    if (!EARLY) fft_inv_butterflies_tab_all<LOGM, LO_B>(x, q, T); else fft_butterflies_reg<false>(x, LAT ? wB : w);
#pragma unroll
    for (int e = 0; e < 8; ++e) bufB[e * SB] = x[e];
    if (EARLY && !LAT) load_inv_twiddles_tab_all<LOGM, LO_A>(w, q, T);
    if (PIPE) bundle_advance(jn, 2 * QB, QB);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = bufA[e * SA];
    if (EARLY) {
      __builtin_amdgcn_sched_barrier(0);
      fft_butterflies_reg<false>(x, LAT ? wA : w);
    } else {
      fft_inv_butterflies_tab_all<LOGM, LO_A>(x, q, T);
    }
    if (PIPE) bundle_advance(jn, 3 * QB, NBATCH - 3 * QB);

    // ---- untwist, round to the torus: the new accumulator ----
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double2 v = cmul_conj(x[e], twi[e]);
      own_lo[e] = torus_from_f64_fast(v.x);
      own_hi[e] = torus_from_f64_fast(v.y);
    }
    // no barrier: the next write into fbuf goes to the slots this thread has just read
  }

  // ---- sample extract from registers: a'_0 = a_0, a'_j = -a_{N-j}, b' = b_0 ----
  uint64_t *out = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int p = fft_point<LOGM, LO_A, 3>(q, e);
    if (r == 0) {
      if (p == 0) out[0] = own_lo[e];
      else out[N - p] = (uint64_t)0 - own_lo[e];
      out[N - (p + M)] = (uint64_t)0 - own_hi[e];
    } else if (p == 0) {
      out[N] = own_lo[e];
    }
  }
}

// engine layout of the Fourier BSK for variant 2: [poly][e][q] with the point
// p = fft_point<LOGM, LO_last3, 3>(q, e) of the last 3-bit pass (the layout the spectrum
// is in when the external product runs).  grid = polynomials, block = M/8.
template <int LOGN>
__global__ void bsk_to_engine_layout_kernel(const double2 *__restrict__ canon, double2 *__restrict__ out, int nc_ilv = 0) {
  constexpr int M = (1 << LOGN) / 2, LOGM = LOGN - 1, TP = M / 8;
  constexpr int NP = (LOGM - 1) / 3;
  constexpr int LO_C = LOGM - 3 * NP;
  const int q = threadIdx.x;
  const double2 *src = canon + (size_t)blockIdx.x * M;
  if (nc_ilv > 1) {  // multi-bit, interleaved: polynomial ((group * NC + c) * 4 + rc) -> rows (rc, e, c) of the group's block
    const int rc = blockIdx.x & 3, c = (blockIdx.x >> 2) % nc_ilv, grp = (blockIdx.x >> 2) / nc_ilv;
    double2 *base = out + (size_t)grp * nc_ilv * 4 * M;
#pragma unroll
    for (int e = 0; e < 8; ++e) base[((size_t)(rc * 8 + e) * nc_ilv + c) * TP + q] = src[fft_point<LOGM, LO_C, 3>(q, e)];
    return;
  }
  double2 *dst = out + (size_t)blockIdx.x * M;
#pragma unroll
  for (int e = 0; e < 8; ++e) dst[(size_t)e * TP + q] = src[fft_point<LOGM, LO_C, 3>(q, e)];
}

}  // namespace dev
}  // namespace fhes
