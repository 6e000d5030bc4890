// kernel_br_wpp.hip.h -- the WAVE-PER-POLYNOMIAL build of the classic blind rotate (round 5, VERDICT r4 item 7: a different
// decomposition of the throughput kernel, chosen by arithmetic; k = 1, one level, N = 2048).
//
// The throughput kernel (blind_rotate_v5_kernel) spreads a polynomial over two waves: 8 points per lane, register passes of
// 3 | 3 | 3 index bits and one stage across lane pairs, four workgroup barriers per CMUX because two of a transform's exchanges
// cross waves.  Its rate is what two resident workgroups overlap of their LDS and VALU phases BETWEEN those barriers
// (1.37 x max(VALU, LDS); profiles/EXPERIMENTS.md, round 4).  Here ONE wave owns a polynomial:
//
//   * 16 points per lane, register passes of 4 | 4 | 2 index bits: ten stages in three passes, NO lane-pair stage (the last
//     forward stage and the first inverse stage are register butterflies: 24 f64 operations instead of 2 x 32 + 64 DPP moves);
//   * both exchanges of a transform are local to the wave: the LDS operations of one wave execute in order, so an exchange
//     needs no barrier at all -- a wave runs from its digits to its spectrum, and from the external product to its image,
//     without waiting for anybody;
//   * two barriers per CMUX: the spectra meet in the external product (hand-over written / hand-over read).  TWO bootstraps share
//     a workgroup (BOOTS = 2: 256 threads, two workgroups per CU), so four waves meet at each barrier although only pairs
//     depend on each other -- on purpose: the first form (BOOTS = 1: 128 threads, four workgroups per CU, eight waves free of
//     each other except in pairs) ran at the throughput build's rate, 126 k PBS/s, because free-running waves collide in the
//     LDS store path at random, while two barrier-synchronised workgroups fall into anti-phase -- one computes while the other
//     exchanges: 136 k in the same run (profiles/r05_probe32_wpp_two_per_workgroup.txt);
//   * the accumulator's image, both exchanges and the hand-over share ONE 17 KB buffer per wave (the own accumulator words live
//     in registers, so the image is dead while the transforms run): 34 KB per bootstrap + 5.5 KB of twiddle tables per
//     workgroup, FOUR bootstraps per CU;
//   * the external product reads BOTH spectra from LDS (the own one from the slots the wave has just written): one instruction
//     sequence for both waves -- a branch on the wave's role got folded into register moves and 96 spilled registers -- and the
//     points' 64 registers are free while it runs;
//   * twiddles: pass 1 of the forward transform has uniform twiddles (scalar registers).  The 8 + 6 + 8 + 8 per-lane twiddles of
//     the other passes and the 16 untwist factors do not fit next to 64 + 64 registers of points and accumulator words, and a
//     per-lane request to L2 costs 0.3 % of the rate EACH (timing-only ablation: 140 k PBS/s without any, 122 k with all 46):
//     forward pass 3 (6) and five of inverse pass 1' stay in registers for the whole loop, forward pass 2 (8), inverse pass 2'
//     (8) and three of pass 1' come from per-workgroup LDS tables read a pass AHEAD (read where they are used they cost 2 %),
//     the untwist factors are the only table requests left in the loop.  A constant the compiler has to spill is reloaded
//     where it is used, latency and all (102 k PBS/s with ten of them spilled): 252 registers, none spilled.
//
// The synthetic twin (scripts/micro/lds_valu_gate.hip, GATE_WAVE_PER_POLY=1; profiles/r05_gate_wave_per_poly.jsonl) priced this
// structure at 5.6 k cycles per CMUX and CU by wall-clock against 6.6 k for the shipped structure's twin (-14 %).  Measured with two
// bootstraps per workgroup: 8-10 % above the throughput build per full round of 1024 bootstraps (136 k against 123 k PBS/s at
// 16 384 in the same run), behind it where the last round is a small remainder (profiles/r05_probe33_wpp2_sizes.txt; DESIGN.md
// section 7 has the reading).  launch_blind_rotate (engine.hip) compares the two step functions of the launch size.
//
// Arithmetic: DESIGN.md section 4 -- the butterflies of blind_rotate_v5_kernel on the same operands in the same stage order, the
// same bits.  Index model of the layouts, the padded slots and every twiddle formula (with the quarter-turn derivations):
// scripts/model/wpp_layouts.py.
//
// Layouts (p = point index p9..p0; e = register e3..e0; l = lane):
//   L1  e = (p9 p8 p7 p6)   l = p5..p0                          digits, forward pass 1, inverse pass 1', untwist, image
//   L2  e = (p5 p4 p3 p2)   l5..l2 = p9..p6, l1 l0 = p1 p0      forward pass 2, inverse pass 2'
//   L3  e = (p3 p2 p1 p0)   l = p9..p4                          forward pass 3, external product, inverse pass 3'
// Slots (16 bytes): exchange L1 <-> L2 at slot1(p) = p + 4 (p >> 6); exchange L2 <-> L3 and the hand-over at slot2(p) = p + (p >> 4):
// conflict-free ds_read / ds_write_b128 in both layouts of each exchange, and every address is base(lane) + constant(e).
#pragma once
#include "kernel_br_lat.hip.h"

namespace fhes {
namespace dev {

constexpr int kWppThreads = 128;
constexpr int kWppSlots = 1088;                                  // padded slots per wave
// two buffers (34 KB) | F: the twiddles of forward pass 2, [8][16 blocks] | I: those of inverse pass 2', [8][4] | J: three of inverse
// pass 1', [3][64 lanes]  = 40 448 bytes: four workgroups per CU (160 KB)
constexpr int kWppTabI = 8 * 16, kWppTabJ = kWppTabI + 8 * 4, kWppTabEnd = kWppTabJ + 3 * 64;   // in slots of 16 bytes, behind the buffers
// (lab, TWL) T: the 1024 untwist factors behind the tables -- 16 KB that only fit with FOUR bootstraps per workgroup (161 792 bytes)
constexpr size_t wpp_lds_bytes(int boots, bool twl = false) { return (size_t)(boots * 2 * kWppSlots + kWppTabEnd + (twl ? 1024 : 0)) * 16; }
constexpr size_t kWppLdsBytes = wpp_lds_bytes(1);

// forward pass over the four register bits (highest first).  t3: the stage of e3; t2: its successor's twiddle for the EVEN block
// (the odd block's is -i times that); t1[2], t0[4]: the even blocks of the last two stages.  `out(e)` is called
// for both points of a last-stage butterfly right behind it: the exchange's stores leave while the next butterflies compute
// (two ds_write_b128 = 26 cycles of the LDS store path, one butterfly = 6 x 4 cycles of the SIMD).
template <class Out>
__device__ __forceinline__ void wpp_fwd_pass4(double2 (&x)[16], double2 t3, double2 t2, const double2 (&t1)[2], const double2 (&t0)[4], Out out) {
#pragma unroll
  for (int e = 0; e < 8; ++e) fwd_butterfly(x[e], x[e + 8], t3);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (e & 4) continue;
    fwd_butterfly(x[e], x[e + 4], (e >> 3) ? mul_neg_i(t2) : t2);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (e & 2) continue;
    const int blk = e >> 2;
    fwd_butterfly(x[e], x[e + 2], (blk & 1) ? mul_neg_i(t1[blk >> 1]) : t1[blk >> 1]);
  }
#pragma unroll
  for (int e = 0; e < 16; e += 2) {
    const int blk = e >> 1;
    fwd_butterfly(x[e], x[e + 1], (blk & 1) ? mul_neg_i(t0[blk >> 1]) : t0[blk >> 1]);
    out(e);
    out(e + 1);
    if ((e & 2) != 0) __builtin_amdgcn_sched_barrier(0);
  }
}
// inverse pass over the four register bits (lowest first).  a: the stage of e0 (depends on the bits below only); b: the next
// stage for e0 = 0 (e0 = 1: a quarter turn further, -i times); c[e & 1], dd[e & 3] likewise.
template <class Out>
__device__ __forceinline__ void wpp_inv_pass4(double2 (&x)[16], double2 a, double2 b, const double2 (&c)[2], const double2 (&dd)[4], Out out) {
#pragma unroll
  for (int e = 0; e < 16; e += 2) inv_butterfly(x[e], x[e + 1], a);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (e & 2) continue;
    inv_butterfly(x[e], x[e + 2], (e & 1) ? mul_neg_i(b) : b);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (e & 4) continue;
    inv_butterfly(x[e], x[e + 4], (e & 2) ? mul_neg_i(c[e & 1]) : c[e & 1]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    inv_butterfly(x[e], x[e + 8], (e & 4) ? mul_neg_i(dd[e & 3]) : dd[e & 3]);
    out(e);
    out(e + 8);
    if ((e & 1) != 0) __builtin_amdgcn_sched_barrier(0);
  }
}

constexpr int kWppStampIter = 300;   // STAMP (lab): every wave records s_memtime at the marked points of ONE iteration
// ABL (lab, timing only, WRONG results): 1 = the per-lane table requests replaced by a register constant; 2 = the GGSW rows as well
// BOOTS (lab): bootstraps per workgroup -- 1: 128 threads, four workgroups per CU; 2: 256 threads, two workgroups per CU whose
// four waves meet at both barriers (the throughput build's occupancy shape: do two workgroups in anti-phase overlap better than
// eight free-running waves?); 4 (lab variants 31 / 32, NOT YET RUN on a device): 512 threads, ONE workgroup per CU, eight waves at
// both barriers -- the tables are shared by four bootstraps, which leaves room for
// TWL (lab): the untwist factors in an LDS table of the workgroup (the 16 requests per CMUX that are left in the loop cost 0.3 % of
// the rate each, profiles/r05_probe28_wpp_ablations.txt)
template <int BL, bool STAMP = false, int ABL = 0, int BOOTS = 1, bool TWL = false>
__global__ __launch_bounds__(kWppThreads * BOOTS, BOOTS >= 4 ? 1 : 2) void blind_rotate_wpp_kernel(
    NodeDesc d, uint64_t *__restrict__ arena, int ct_stride, const uint64_t *__restrict__ ks_out, int ks_stride, int n_small,
    const double2 *__restrict__ bsk_e, const uint64_t *__restrict__ lut_polys,
    const double2 *__restrict__ twist_inv /* twist / (M * 2^64): torus_from_f64_pair */, const double2 *__restrict__ tw,
    const double2 *__restrict__ psi, unsigned long long *__restrict__ stamps = nullptr, int v_count = 0) {
  constexpr int LOGN = 11, N = 1 << LOGN, M = N / 2;
  static_assert(BL >= 1 && BL <= 31, "single-level digit must fit the high dword");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  int it_now = -1;
#define FHES_WPP_STAMP(k)                                                                                  \
  do {                                                                                                     \
    if (STAMP && it_now == kWppStampIter && blockIdx.x * BOOTS < 1024) {                                        \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                          \
      if ((tid & 63) == 0) stamps[((size_t)blockIdx.x * 2 * BOOTS + (size_t)(tid >> 6)) * kLatStamps + (k)] = t_; \
    }                                                                                                      \
  } while (0)
  const int r = __builtin_amdgcn_readfirstlane((tid >> 6) & 1);  // wave = output polynomial
  const int boot = __builtin_amdgcn_readfirstlane(tid >> 7);     // bootstrap inside the workgroup (BOOTS = 2)
  const int l = tid & 63;
  unsigned char *boot_base = smem_raw + (size_t)boot * 2 * kWppSlots * 16;
  unsigned char *mine = boot_base + (size_t)r * kWppSlots * 16;
  uint64_t *img = reinterpret_cast<uint64_t *>(mine);            // [N] image of -acc, plain order (while no transform runs)
  double2 *xb = reinterpret_cast<double2 *>(mine);               // [kWppSlots] exchanges and the hand-over
  // BOOTS = 2 and an odd launch: the last workgroup's second half repeats the last bootstrap (same inputs, same words written twice)
  const int slot_in_launch = BOOTS == 1 ? (int)blockIdx.x : min((int)blockIdx.x * BOOTS + boot, v_count - 1);
  const int vidx = d.v0 + slot_in_launch;
  const int inst = vidx / d.count, node = d.first + vidx % d.count;

  // lane bases of the slots (scripts/model/wpp_layouts.py prints the constants)
  const int h4 = l >> 2, q2 = l & 3;
  double2 *s1 = xb + l;                      // L1 / slot1: + 68 e
  double2 *s2 = xb + h4 * 68 + q2;           // L2 / slot1: + 4 e;   L2 / slot2: + 4 e + (e >> 2)
  double2 *s3 = xb + l * 17;                 // L3 / slot2: + e
  // the hand-over as the external product reads it: BOTH spectra from LDS (the own one from the slots this wave has just
  // written), so the product is the same instruction sequence for both waves and the points' registers are free while it runs
  const double2 *d0s = reinterpret_cast<const double2 *>(boot_base) + l * 17;
  const double2 *d1s = reinterpret_cast<const double2 *>(boot_base + (size_t)kWppSlots * 16) + l * 17;

  // uniform twiddles of forward pass 1 and of the inverse stage of index bit 1
  const double2 u3 = psi[1], u2 = psi[2];
  const double2 u1[2] = {psi[4], psi[6]};
  const double2 u0[4] = {psi[8], psi[10], psi[12], psi[14]};
  const double2 iw0 = tw[0], iw1 = tw[M / 4];

  const __amdgpu_buffer_rsrc_t rs_g =
      __builtin_amdgcn_make_buffer_rsrc((void *)bsk_e, 0, (unsigned)((size_t)n_small * 4 * M * 16), 0x00020000);
  // Fourier key in the engine layout [i][row][col][e8][128]: point p of L3 sits at e8 = e >> 1, position 2 l + (e & 1)
  const unsigned voff_g = (unsigned)(r * 8 * 128 + l * 2) * 16u;
  // the tables through buffer descriptors: one lane offset per request group, the rest immediates (a 64-bit address per request
  // would cost two registers each)
  const __amdgpu_buffer_rsrc_t rs_psi = __builtin_amdgcn_make_buffer_rsrc((void *)psi, 0, (unsigned)(M * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_tw = __builtin_amdgcn_make_buffer_rsrc((void *)tw, 0, (unsigned)(M / 2 * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_twi = __builtin_amdgcn_make_buffer_rsrc((void *)twist_inv, 0, (unsigned)(M * 16), 0x00020000);
  const unsigned lb = (unsigned)l * 16u;
  double abl_c = 0.70710678;
  asm volatile("" : "+v"(abl_c));
  auto tab = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) __attribute__((always_inline)) {
    if (ABL >= 1) return make_double2(abl_c, -abl_c);
    return buf_load_d2(rs, voff, soff);
  };
  auto row = [&](unsigned soff) __attribute__((always_inline)) {
    if (ABL >= 2) return make_double2(abl_c, abl_c);
    return buf_load_d2(rs_g, voff_g, soff);
  };

  const uint64_t *ks = ks_out + (size_t)slot_in_launch * ks_stride;
  auto modswitch_of = [&](uint64_t v) __attribute__((always_inline)) {
    return (int)((((v >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
  };

  // ---- accumulator: acc = (0, X^{-b~} LUT); registers and image hold -acc ----
  uint64_t nlo[16], nhi[16];
  {
    const uint64_t *lut = lut_polys + (size_t)d.lut_ids[node] * N;
    const int t = (2 * N - modswitch_of(ks[n_small])) & (2 * N - 1);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int p = l + 64 * e;
      uint64_t v0 = 0, v1 = 0;
      if (r == 1) {
        v0 = (t == 0) ? lut[p] : rot_read<N>(lut, t, p);
        v1 = (t == 0) ? lut[p + M] : rot_read<N>(lut, t, p + M);
      }
      nlo[e] = (uint64_t)0 - v0;
      nhi[e] = (uint64_t)0 - v1;
      img[p] = nlo[e];
      img[p + M] = nhi[e];
    }
  }
  // (no barrier for the image: a wave rotates its own polynomial, and its own LDS operations execute in order)
  // ---- twiddle tables of the workgroup (both waves read them): per-lane requests to L2 cost 0.3 % of the rate EACH
  // (profiles/r05_probe28_wpp_ablations.txt), an LDS read next to the exchange's sixteen costs nothing measurable ----
  {
    double2 *tabs = reinterpret_cast<double2 *>(smem_raw) + BOOTS * 2 * kWppSlots;
    if (tid < 128) {   // F[k][h]: psi[16 + h], psi[32 + 2 h], psi[64 + 4 h], psi[66 + 4 h], psi[128 + 8 h + 2 j]
      const int k = tid >> 4, h = tid & 15;
      const int idx = k == 0 ? 16 + h : (k == 1 ? 32 + 2 * h : (k < 4 ? 64 + 4 * h + 2 * (k - 2) : 128 + 8 * h + 2 * (k - 4)));
      tabs[tid] = psi[idx];
    }
    if (tid < 32) {    // I[k][q]: tw[q << 7], tw[q << 6], tw[q << 5], tw[(4 + q) << 5], tw[(4 j + q) << 4]
      const int k = tid >> 2, q = tid & 3;
      const int idx = k == 0 ? q << 7 : (k == 1 ? q << 6 : (k < 4 ? (4 * (k - 2) + q) << 5 : (4 * (k - 4) + q) << 4));
      tabs[kWppTabI + tid] = tw[idx];
    }
    for (int t = tid; t < 192; t += kWppThreads * BOOTS) {   // J[k][l]: tw[l << 3], tw[l << 2], tw[l << 1]
      const int k = t >> 6, ll = t & 63;
      tabs[kWppTabJ + t] = tw[ll << (3 - k)];
    }
    if (TWL)
      for (int t = tid; t < M; t += kWppThreads * BOOTS) tabs[kWppTabEnd + t] = twist_inv[t];   // T[p]
  }
  __syncthreads();
  const double2 *tab0 = reinterpret_cast<const double2 *>(smem_raw) + BOOTS * 2 * kWppSlots;
  const double2 *tabF = tab0 + h4;               // + 16 k
  const double2 *tabI = tab0 + kWppTabI + q2;    // + 4 k
  const double2 *tabJ = tab0 + kWppTabJ + l;     // + 64 k
  const double2 *tabT = tab0 + kWppTabEnd + l;   // + 64 e (TWL)

  // the per-lane twiddles of forward pass 3 stay in registers for the whole loop (6 entries: their requests are the scattered ones --
  // psi[512 + 8 l + ..] is 64 cache lines per request; more do not fit: a spilled constant is reloaded where it is used, latency and
  // all -- 102 k PBS/s with 14 entries held and 10 of them spilled); the others are requested per iteration, a pass ahead
  const double2 g1t[2] = {tab(rs_psi, lb * 4, 256 * 16), tab(rs_psi, lb * 4, 258 * 16)};
  const double2 g0t[4] = {tab(rs_psi, lb * 8, 512 * 16), tab(rs_psi, lb * 8, 514 * 16), tab(rs_psi, lb * 8, 516 * 16), tab(rs_psi, lb * 8, 518 * 16)};
  // ... and so do five of inverse pass 1' (the registers that are left)
  const double2 jc1 = tab(rs_tw, lb << 1, (64u << 1) * 16);
  const double2 jd[4] = {tab(rs_tw, lb, 0), tab(rs_tw, lb, 64 * 16), tab(rs_tw, lb, 128 * 16), tab(rs_tw, lb, 192 * 16)};
  constexpr uint64_t ROUND = (uint64_t)1 << (63 - BL);
  int a_next = modswitch_of(ks[0]);
  for (int i = 0; i < n_small; ++i) {
    const int a = a_next;
    it_now = i;
    FHES_WPP_STAMP(0);
    // an opaque zero per iteration: the table requests below depend on it, so they stay inside the loop (hoisted, the 46
    // per-lane constants would need 184 registers)
    unsigned z = 0;
    asm volatile("" : "+s"(z));
    double2 x[16];
    // ---- digits of X^a acc - acc at the own coefficients p (re) and p + M (im), two halves of eight points ----
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint64_t va[8], vb[8];
      int m0v[8], m1v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = half * 8 + k;
        const unsigned s0 = (unsigned)(l + 64 * e - a) & (unsigned)(2 * N - 1);
        const unsigned s1r = (s0 + (unsigned)M) & (unsigned)(2 * N - 1);
        // the image holds -acc: the operand is +acc[s] = -image when the rotation did NOT wrap, -acc[s] = image when it did
        m0v[k] = (int)((s0 >> LOGN) & 1u) - 1;   // -1: negate
        m1v[k] = (int)((s1r >> LOGN) & 1u) - 1;
        va[k] = img[s0 & (unsigned)(N - 1)];
        vb[k] = img[s1r & (unsigned)(N - 1)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = half * 8 + k;
        uint64_t dd[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int m32 = hh == 0 ? m0v[k] : m1v[k];
          const uint64_t v = hh == 0 ? va[k] : vb[k];
          const uint64_t mm = ((uint64_t)(uint32_t)m32 << 32) | (uint64_t)(uint32_t)m32;
          const uint64_t kk = ROUND | (uint64_t)(uint32_t)(m32 & 1);
          dd[hh] = add64(add64(v ^ mm, hh == 0 ? nlo[e] : nhi[e]), kk);
        }
        const double d0 = (double)((int)(uint32_t)(dd[0] >> 32) >> (32 - BL));
        const double d1 = (double)((int)(uint32_t)(dd[1] >> 32) >> (32 - BL));
        x[e] = make_double2(d0, d1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t ks_next = ks[i + 1];
    // twiddles of forward pass 2 (LDS table), a pass ahead
    const double2 f3 = tabF[0], f2 = tabF[16];
    const double2 f1[2] = {tabF[32], tabF[48]};
    const double2 f0[4] = {tabF[64], tabF[80], tabF[96], tabF[112]};
    FHES_WPP_STAMP(1);
    // ---- forward transform ----
    // every rotated read of the image has returned (the digits are computed): the buffer is the exchange buffer from here on
    asm volatile("" ::: "memory");
    wpp_fwd_pass4(x, u3, u2, u1, u0, [&](int e) __attribute__((always_inline)) { s1[68 * e] = x[e]; });   // bits 9 8 7 6 (uniform twiddles)
    FHES_WPP_STAMP(2);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k) {   // in the order the first stage pairs them: (e, e + 8)
      const int e = (k >> 1) + 8 * (k & 1);
      x[e] = s2[4 * e];
    }
    FHES_WPP_STAMP(3);
    a_next = modswitch_of(ks_next);
    asm volatile("" : "+s"(a_next));
    __builtin_amdgcn_sched_barrier(0);
    wpp_fwd_pass4(x, f3, f2, f1, f0, [&](int e) __attribute__((always_inline)) { s2[4 * e + (e >> 2)] = x[e]; });   // bits 5 4 3 2
    FHES_WPP_STAMP(4);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k) {   // in the order the stage of bit 1 pairs them: (e, e + 2)
      const int e = (k & 1) * 2 + ((k >> 1) & 1) + (k >> 2) * 4;
      x[e] = s3[e];
    }
    // GGSW rows: four points' worth requested here, the next four behind the hand-over, then one group ahead of the products
    // (all sixteen in flight at once measured the same rate and takes 64 registers more)
    const unsigned so = (unsigned)i * (unsigned)(4 * M * 16);
    double2 g0[2][4], g1[2][4];
    auto load_rows = [&](int c) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = 4 * c + k;
        const unsigned eo = (unsigned)(((e >> 1) * 128 + (e & 1)) * 16);
        g0[c & 1][k] = row(so + (unsigned)(0 * 16 * 128 * 16) + eo);
        g1[c & 1][k] = row(so + (unsigned)(1 * 16 * 128 * 16) + eo);
      }
    };
    load_rows(0);
    __builtin_amdgcn_sched_barrier(0);
    FHES_WPP_STAMP(5);
#pragma unroll
    for (int e = 0; e < 16; ++e) {                                      // bit 1
      if (e & 2) continue;
      const int blk = e >> 2;
      fwd_butterfly(x[e], x[e + 2], (blk & 1) ? mul_neg_i(g1t[blk >> 1]) : g1t[blk >> 1]);
    }
#pragma unroll
    for (int e = 0; e < 16; e += 2) {                                   // bit 0: the internal form of DESIGN.md 4.3
      const int blk = e >> 1;
      fwd_butterfly_last(x[e], x[e + 1], (blk & 1) ? mul_neg_i(g0t[blk >> 1]) : g0t[blk >> 1]);
      // hand-over: the spectrum where the other wave finds it (the slots this wave has just read), behind each butterfly
      s3[e] = x[e];
      s3[e + 1] = x[e + 1];
      if ((e & 2) != 0) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_rows(1);
    FHES_WPP_STAMP(6);
    __syncthreads();  // B1: both spectra are written
    FHES_WPP_STAMP(7);
    // ---- external product: D0 G[0][r] + D1 G[1][r], four points at a time; both spectra from LDS ----
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double2 da[4], db[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        da[k] = d0s[4 * c + k];
        db[k] = d1s[4 * c + k];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 4; ++k) x[4 * c + k] = ext_mac(da[k], db[k], g0[c & 1][k], g1[c & 1][k]);
      __builtin_amdgcn_sched_barrier(0);
      if (c < 2) load_rows(c + 2);
    }
    // twiddles of inverse pass 2' (LDS table), a pass ahead
    const double2 ia = tabI[0], ib = tabI[4];
    const double2 ic[2] = {tabI[8], tabI[12]};
    const double2 id[4] = {tabI[16], tabI[20], tabI[24], tabI[28]};
    // ---- inverse transform ----
    FHES_WPP_STAMP(8);
#pragma unroll
    for (int e = 0; e < 16; e += 2) {                                   // bit 0: twiddle exactly 1
      const double2 u = x[e], v = x[e + 1];
      x[e] = make_double2(u.x + v.x, u.y + v.y);
      x[e + 1] = make_double2(u.x - v.x, u.y - v.y);
    }
    FHES_WPP_STAMP(9);
    __syncthreads();  // B2: both waves have read both spectra: the buffers are their owners' again (in front of the stage whose
                      // results go back into them, so that its stores can leave butterfly by butterfly)
    FHES_WPP_STAMP(10);
#pragma unroll
    for (int e = 0; e < 16; ++e) {                                      // bit 1: tw[p0 << 8]
      if (e & 2) continue;
      inv_butterfly(x[e], x[e + 2], (e & 1) ? iw1 : iw0);
      s3[e] = x[e];
      s3[e + 2] = x[e + 2];
      if ((e & 1) != 0) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = s2[4 * e + (e >> 2)];
    FHES_WPP_STAMP(11);
    // twiddles of inverse pass 1'
    const double2 ja = tabJ[0], jb = tabJ[64];
    const double2 jc[2] = {tabJ[128], jc1};
    __builtin_amdgcn_sched_barrier(0);
    wpp_inv_pass4(x, ia, ib, ic, id, [&](int e) __attribute__((always_inline)) { s2[4 * e] = x[e]; });   // bits 2 3 4 5
    FHES_WPP_STAMP(12);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = s1[68 * e];
    FHES_WPP_STAMP(13);
    double2 tf[2][4];
    auto load_tf = [&](int c) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) tf[c & 1][k] = TWL ? tabT[64 * (4 * c + k)] : tab(rs_twi, lb, z + (unsigned)(64 * (4 * c + k) * 16));
    };
    load_tf(0);
    load_tf(1);
    __builtin_amdgcn_sched_barrier(0);
    wpp_inv_pass4(x, ja, jb, jc, jd, [](int) {});                       // bits 6 7 8 9
    FHES_WPP_STAMP(14);
    // every exchange read has returned: the buffer is the image again
    asm volatile("" ::: "memory");
    // ---- untwist (negated factors: -y), round, nacc += rint(-y); the image goes back ----
    {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = 4 * c + k;
          const double2 v = cmul_conj(x[e], make_double2(-tf[c & 1][k].x, -tf[c & 1][k].y));
          nlo[e] = add64(torus_from_f64_pair(v.x), nlo[e]);
          nhi[e] = add64(torus_from_f64_pair(v.y), nhi[e]);
          img[l + 64 * e] = nlo[e];
          img[l + 64 * e + M] = nhi[e];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c < 2) load_tf(c + 2);
      }
    }
    asm volatile("" ::: "memory");
    FHES_WPP_STAMP(15);
  }
#undef FHES_WPP_STAMP
  __syncthreads();

  // ---- sample extract from the images of -acc: a'_0 = a_0, a'_j = -a_{N-j}, b' = b_0 ----
  uint64_t *out = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
  const uint64_t *img0 = reinterpret_cast<const uint64_t *>(boot_base);
  const uint64_t *img1 = reinterpret_cast<const uint64_t *>(boot_base + (size_t)kWppSlots * 16);
  for (int j = tid & 127; j < N; j += kWppThreads) {
    const int jj = (j == 0) ? 0 : N - j;
    const uint64_t v = img0[jj];
    out[j] = (j == 0) ? (uint64_t)0 - v : v;
  }
  if ((tid & 127) == 0) out[N] = (uint64_t)0 - img1[0];
}

}  // namespace dev
}  // namespace fhes
