// kernel_br_lat.hip.h -- the LATENCY build of the classic blind rotate (round 5; k = 1, one level, N = 2048).
//
// One call of the reference's API is a chain of dependent levels of 3-4 bootstraps (the serial loops mod.rs:341-349,
// search.rs:91-109, split.rs:274-319): every level costs one whole blind rotate on 3-4 of the 256 CUs.  The throughput kernel
// (blind_rotate_v5_kernel: 256 threads, 8 points per thread, two workgroups per CU) alone on a CU has ONE wave per SIMD, and
// a wave that has issued its LDS stores issues no arithmetic until they are through: the CU's LDS pipe and its VALUs take turns
// (3.8 k + 4.0 k cycles per CMUX + barriers = 9.2 k).  Here ONE bootstrap runs on 512 threads -- two waves per SIMD, each with
// half the work, covering each other's LDS time -- with 4 points per thread:
//
//   * 256 threads per output polynomial (half r = tid >> 8), wave w = (tid >> 6) & 3, lane l; a transform is five 2-bit register
//     passes A..E (E: one stage) and the lane-pair stage.  Between them the data is re-laid out
//         A -> B   through LDS across the four waves of a half (workgroup barrier),
//         B -> C   inside a wave WITHOUT LDS: v_permlane32_swap / v_permlane16_swap transpose a register bit with lane bit 5 / 4,
//         C -> D   through LDS inside a wave (no barrier: the LDS operations of one wave execute in order),
//         D -> E   v_permlane32_swap,
//     so a transform costs the same two LDS round trips per point as the throughput kernel's three 3-bit passes;
//   * with 4 points per thread the state is 16 + 16 + 32 registers (own accumulator words, points, GGSW rows), so EVERY
//     twiddle of every pass, the lane-pair constants and the untwist factors stay in registers for the whole loop: no table
//     reads from LDS at all;
//   * one workgroup per CU has 128 KB of LDS to itself: the image of -acc, two exchange buffers and the spectrum hand-over
//     buffer are separate, the image is in plain coefficient order (rotated 8-byte reads of consecutive lanes are consecutive
//     words: conflict-free, where the throughput kernel pays 76 conflict cycles per wave for its in-place image).
//   * four workgroup barriers per CMUX, as in the throughput kernel (each is a true all-to-all dependence).
//
// Arithmetic: DESIGN.md section 4, statement by statement the butterflies of blind_rotate_v5_kernel on the same operands in
// the same stage order -- the same bits (tests/test_gpu_pbs.py compares every variant with the oracle word for word).
// Index model of the layouts, swizzles and transposes, checked against a plain stage-by-stage transform and against the LDS
// bank rules: scripts/model/lat_layouts.py.
//
// Layouts (p = point index, bits p9..p0; e = register, bits e1 e0; l = lane, bits l5..l0):
//   L_A  e = (p9, p8)   q = tid & 255 = p7..p0
//   L_B  w = (p9, p8)   e = (p7, p6)   l5 = p4, l4 = p5, l3..l0 = p3..p0
//   L_C  w = (p9, p8)   e = (p5, p4)   l5 = p6, l4 = p7, l3..l0 = p3..p0          (L_B with e0 <-> l5, e1 <-> l4)
//   L_D  w = (p9, p8)   e = (p3, p2)   l5 = p1, l4..l1 = p7..p4, l0 = p0
//   L_E  w = (p9, p8)   e = (p3, p1)   l5 = p2, l4..l1 = p7..p4, l0 = p0          (L_D with e0 <-> l5)
#pragma once
#include "kernels.hip.h"

namespace fhes {
namespace dev {

__device__ __forceinline__ void lat_swap32(double &a, double &b) {  // register bit <-> lane bit 5
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void lat_swap16(double &a, double &b) {  // register bit <-> lane bit 4
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// e0 <-> l5 on both register pairs
__device__ __forceinline__ void lat_transpose_e0_l5(double2 (&x)[4]) {
  lat_swap32(x[0].x, x[1].x); lat_swap32(x[0].y, x[1].y);
  lat_swap32(x[2].x, x[3].x); lat_swap32(x[2].y, x[3].y);
}
// e1 <-> l4
__device__ __forceinline__ void lat_transpose_e1_l4(double2 (&x)[4]) {
  lat_swap16(x[0].x, x[2].x); lat_swap16(x[0].y, x[2].y);
  lat_swap16(x[1].x, x[3].x); lat_swap16(x[1].y, x[3].y);
}

// Register bit <-> lane bit 3 / 2 WITHOUT LDS (layout v2): a row shift by 8 / 4 lanes whose bank mask lets only the lanes take the
// partner's value that need it (banks = groups of four lanes: lane bit 3 = banks 2, 3; lane bit 2 = banks 1, 3); the other lanes keep
// `old`.  Same exchange as the permlane swaps: the first register of a pair ends up with the data of the lanes whose bit was 0.
template <int CTRL, int BANKS>
__device__ __forceinline__ double lat_dpp_upd(double old, double src) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANKS, false);
  return __hiloint2double(hi, lo);
}
template <int SHIFT>   // SHIFT = 8: lane bit 3, SHIFT = 4: lane bit 2
__device__ __forceinline__ void lat_swap_row(double &a, double &b) {
  constexpr int SHR = 0x110 + SHIFT, SHL = 0x100 + SHIFT;
  constexpr int HI = SHIFT == 8 ? 0xC : 0xA, LO = SHIFT == 8 ? 0x3 : 0x5;   // banks whose lanes have the bit set / clear
  const double a2 = lat_dpp_upd<SHR, HI>(a, b);   // lanes with the bit set: b of the lane SHIFT below
  const double b2 = lat_dpp_upd<SHL, LO>(b, a);   // lanes with the bit clear: a of the lane SHIFT above
  a = a2;
  b = b2;
}
__device__ __forceinline__ void lat_transpose_e1_l3(double2 (&x)[4]) {
  lat_swap_row<8>(x[0].x, x[2].x); lat_swap_row<8>(x[0].y, x[2].y);
  lat_swap_row<8>(x[1].x, x[3].x); lat_swap_row<8>(x[1].y, x[3].y);
}
__device__ __forceinline__ void lat_transpose_e0_l2(double2 (&x)[4]) {
  lat_swap_row<4>(x[0].x, x[1].x); lat_swap_row<4>(x[0].y, x[1].y);
  lat_swap_row<4>(x[2].x, x[3].x); lat_swap_row<4>(x[2].y, x[3].y);
}
// register bit e0 <-> lane bit 1: quad permute [2, 3, 0, 1] and a per-lane select (`odd2`: lane bit 1 set)
__device__ __forceinline__ double lat_dpp_xor2(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x4E, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x4E, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void lat_swap_quad(double &a, double &b, bool odd2) {
  const double pb = lat_dpp_xor2(b), pa = lat_dpp_xor2(a);
  const double a2 = odd2 ? pb : a, b2 = odd2 ? b : pa;
  a = a2;
  b = b2;
}
__device__ __forceinline__ void lat_transpose_e0_l1(double2 (&x)[4], bool odd2) {
  lat_swap_quad(x[0].x, x[1].x, odd2); lat_swap_quad(x[0].y, x[1].y, odd2);
  lat_swap_quad(x[2].x, x[3].x, odd2); lat_swap_quad(x[2].y, x[3].y, odd2);
}

// forward 2-bit pass: the stage of the higher index bit (e1) first; hi = its twiddle (the block does not depend on e), lo0 / lo1
// the twiddles of the lower stage for e1 = 0 / 1
__device__ __forceinline__ void lat_fwd_pass(double2 (&x)[4], double2 hi, double2 lo0, double2 lo1) {
  fwd_butterfly(x[0], x[2], hi);
  fwd_butterfly(x[1], x[3], hi);
  fwd_butterfly(x[0], x[1], lo0);
  fwd_butterfly(x[2], x[3], lo1);
}
// inverse 2-bit pass: the stage of the lower index bit (e0) first; lo = its twiddle (depends on the bits below only), hi0 / hi1
// the twiddles of the higher stage for e0 = 0 / 1
__device__ __forceinline__ void lat_inv_pass(double2 (&x)[4], double2 lo, double2 hi0, double2 hi1) {
  inv_butterfly(x[0], x[1], lo);
  inv_butterfly(x[2], x[3], lo);
  inv_butterfly(x[0], x[2], hi0);
  inv_butterfly(x[1], x[3], hi1);
}
__device__ __forceinline__ void lat_fwd_stage0_dpp(double2 (&x)[4], const double2 (&c)[4]) {  // fft_fwd_stage0_dpp on 4 points
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double px = dpp_swap1(x[e].x), py = dpp_swap1(x[e].y);
    double re = __builtin_fma(c[e].x, px, x[e].x);
    re = __builtin_fma(-c[e].y, py, re);
    double im = __builtin_fma(c[e].x, py, x[e].y);
    im = __builtin_fma(c[e].y, px, im);
    x[e] = make_double2(re, im);
  }
}
__device__ __forceinline__ void lat_inv_stage0_dpp(double2 (&x)[4], double sgn) {  // fft_stage_bit0_dpp on 4 points
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double px = dpp_swap1(x[e].x), py = dpp_swap1(x[e].y);
    x[e].x = __builtin_fma(x[e].x, sgn, px);
    x[e].y = __builtin_fma(x[e].y, sgn, py);
  }
}

constexpr int kLatThreads = 512;
// image (32 KB) | X: both exchanges of a half (32 KB; the wave-local C <-> D exchange re-uses the region the wave has just read or
// is about to write) | S: spectra for the other half, double-buffered by iteration parity (64 KB) | flags of the half barriers (2 KB)
constexpr size_t kLatLdsBytes = (size_t)2 * 2048 * 8 + (size_t)2 * 1024 * 16 + (size_t)2 * 2 * 1024 * 16 + 2048;

// Barrier among the four waves of ONE half (SYNC = 1).  Of the four barriers of a CMUX only one is a dependence between the
// halves (the spectra meet in the external product); the exchanges A <-> B and the image of -acc are a half's own.  gfx950 has one
// s_barrier per workgroup, so a half's barrier is a flag per wave in LDS: the LDS operations of a wave execute in order, so
// a wave's flag store lands after its data stores, and a wave that has read every flag >= epoch reads the data behind them.
// Every lane keeps its own flag word (slot [lane][wave]: no EXEC games, the poll is one conflict-free ds_read_b128).
__device__ __forceinline__ void lat_half_barrier(unsigned flags_lane /* byte address of this lane's 4 flags */, unsigned w, unsigned epoch) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(flags_lane + 4u * w), "v"(epoch) : "memory");
  for (int spin = 0; spin < (1 << 22); ++spin) {   // bounded: a broken build ends with wrong bits, never with a hung GPU
    fhes_u32x4 f;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(flags_lane) : "memory");
    const unsigned m = min(min(f.x, f.y), min(f.z, f.w));
    if (__builtin_amdgcn_readfirstlane(m) >= epoch) break;   // every lane of a wave stores the same epoch: lane 0 speaks for all
    __builtin_amdgcn_s_sleep(1);
  }
}

// LAT_ABL: timing-only ablations for the lab book (FHES_LAB builds; WRONG results): 1 = without the LDS exchanges and the
// spectrum hand-over, 2 = without the butterflies
// SYNC: 0 = four workgroup barriers per CMUX; 1 = B1 / B5 / B6 among the waves of a half only (lat_half_barrier), B3 for all;
// 2 = timing-only ablation WITHOUT B1 / B5 / B6 (racy: wrong results)
// OPT packs the lab switches: bits 0-1 LAT_ABL, bits 2-3 SYNC, bits 4-5 PRIO (1: the waves of half 0 at high priority for the
// whole kernel, 2: those of half 1), bit 6 GPF (the GGSW rows of iteration i + 1 requested right behind the external product of
// iteration i -- a whole iteration ahead, into the registers that have just died -- instead of at the top of iteration i + 1)
constexpr int lat_opt(int abl, int sync, int prio, int gpf, int stamp = 0, int ils = 0) { return abl | (sync << 2) | (prio << 4) | (gpf << 6) | (stamp << 7) | (ils << 8); }   // + bit 9 RPRE, below
// what the product launches: GGSW rows a whole iteration ahead, stores pair by pair behind their butterflies (-3 %), the next
// iteration's rotated-read addresses in front of the closing barrier (-2.5 %): 2.47 -> 2.33 ms per bootstrap in the same run
// (profiles/r05_probe14_lat_ils_rpre_v2.txt; layout v2 -- no LDS inside a wave -- is 3 % SLOWER: 80 more VALU slots per transform)
constexpr int kLatDefaultOpt = lat_opt(0, 0, 0, 1, 0, 1) | (1 << 9);
constexpr int kLatStamps = 32, kLatStampIter = 300;   // bit 7 STAMP: every wave records s_memtime at the marked points of ONE iteration (lab)
template <int BL, int OPT = 0>
__global__ __launch_bounds__(kLatThreads, 1) void blind_rotate_lat_kernel(
    NodeDesc d, uint64_t *__restrict__ arena, int ct_stride, const uint64_t *__restrict__ ks_out, int ks_stride, int n_small,
    const double2 *__restrict__ bsk_e, const uint64_t *__restrict__ lut_polys,
    const double2 *__restrict__ twist_inv /* twist / (M * 2^64): torus_from_f64_pair */, const double2 *__restrict__ tw,
    const double2 *__restrict__ psi, unsigned long long *__restrict__ stamps = nullptr) {
  constexpr int LOGN = 11, N = 1 << LOGN, M = N / 2;
  constexpr int LAT_ABL = OPT & 3, SYNC = (OPT >> 2) & 3, PRIO = (OPT >> 4) & 3, GPF = (OPT >> 6) & 1, STAMP = (OPT >> 7) & 1;
  // bit 8 ILS: the stores of an exchange go out pair by pair, each behind the last butterfly that produces it, instead of as one
  // burst behind the pass
  constexpr bool ILS = ((OPT >> 8) & 1) && LAT_ABL == 0;
  // bit 9 RPRE: the addresses and sign masks of the NEXT iteration's rotated reads are formed in front of the closing barrier (the
  // rotation is known since B1), where the wave waits for its image stores anyway -- not behind it, on every wave's critical path
  constexpr bool RPRE = ((OPT >> 9) & 1) != 0;
  // bit 10 V2: the in-wave part without LDS -- C <-> D as transposes with lane bits 3 / 2, D <-> E with lane bit 1 (layouts L_D', L_E':
  // e = (p3, p2) / (p3, p1); l5 = p6, l4 = p7, l3 = p5, l2 = p4, l1 = p1 / p2, l0 = p0; scripts/model/lat_layouts_v2.py)
  constexpr bool V2 = ((OPT >> 10) & 1) != 0 && LAT_ABL == 0;
  static_assert(BL >= 1 && BL <= 31, "single-level digit must fit the high dword");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t *img_all = reinterpret_cast<uint64_t *>(smem_raw);                       // [2][N]   image of -acc, plain order
  double2 *x1_all = reinterpret_cast<double2 *>(smem_raw + (size_t)2 * N * 8);      // [2][M]   A <-> B exchanges ...
  double2 *x2_all = x1_all;                                                         //          ... and C <-> D (wave-local, same region)
  double2 *s_all = x1_all + 2 * M;                                                  // [2 parities][2][M]   spectra for the other half
  const unsigned flags0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw + (unsigned)(2 * N * 8 + 6 * M * 16);

  const int tid = threadIdx.x;
  const int r = __builtin_amdgcn_readfirstlane(tid >> 8);        // half = output polynomial
  const int w = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);  // wave inside the half = (p9, p8) in layouts B..E
  const int q = tid & 255;
  const int l = tid & 63;
  const int l0 = l & 1, l4 = (l >> 4) & 1, l5 = (l >> 5) & 1, lo4 = l & 15;
  const int vidx = d.v0 + blockIdx.x;
  const int inst = vidx / d.count, node = d.first + vidx % d.count;

  uint64_t *img = img_all + (size_t)r * N;
  double2 *x1 = x1_all + (size_t)r * M;
  double2 *x2 = x2_all + (size_t)r * M;
  double2 *s_r = s_all + (size_t)r * M;              // + parity * 2 M
  const double2 *s_o = s_all + (size_t)(1 - r) * M;
  const unsigned flags_lane = flags0 + (unsigned)r * 1024u + (unsigned)l * 16u;
  if (SYNC == 1) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(flags_lane + 4u * (unsigned)w), "v"(0u) : "memory");
  }
  unsigned epoch = 0;
  int it_now = -1;
#define FHES_LAT_STAMP(k)                                                                                         \
  do {                                                                                                            \
    if (STAMP && it_now == kLatStampIter) {                                                                       \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                 \
      if (l == 0) stamps[((size_t)blockIdx.x * 8 + (size_t)(tid >> 6)) * kLatStamps + (k)] = t_;                   \
    }                                                                                                             \
  } while (0)
#define FHES_LAT_HALF_BARRIER()                                           \
  do {                                                                    \
    if (SYNC == 0) __syncthreads();                                       \
    else if (SYNC == 1) lat_half_barrier(flags_lane, (unsigned)w, ++epoch); \
  } while (0)

  // ---- slots of this thread in every layout (element indices inside a half's buffer) ----
  const int l1 = (l >> 1) & 1, l2 = (l >> 2) & 1, l3 = (l >> 3) & 1;
  const bool odd2 = l1 != 0;
  const int blk16 = V2 ? w * 16 + l4 * 8 + l5 * 4 + l3 * 2 + l2 : w * 16 + ((l >> 1) & 15);   // p >> 4 in layouts D / E
  const int lE = V2 ? l1 : l5;                                 // the lane bit that holds p1 (layout D) / p2 (layout E)
  int sA[4], sB[4], sC[4], sD[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sA[e] = q + 256 * e;
    sB[e] = w * 256 + e * 64 + l4 * 32 + l5 * 16 + lo4;
    sC[e] = fswz(w * 256 + l4 * 128 + l5 * 64 + e * 16 + lo4);
    sD[e] = fswz(blk16 * 16 + e * 4 + l5 * 2 + l0);   // (layout v1 only)
  }

  // ---- every twiddle of both transforms, once (registers for the whole loop) ----
  // forward: psi[2^s + block], s = number of stages done so far, block = the index bits above the stage's bit
  const double2 fA_hi = psi[1], fA_lo0 = psi[2], fA_lo1 = psi[3];
  const double2 fB_hi = psi[4 + w], fB_lo0 = psi[8 + 2 * w], fB_lo1 = psi[8 + 2 * w + 1];
  const int blkC = w * 4 + l4 * 2 + l5;                       // p >> 6 in layout C
  const double2 fC_hi = psi[16 + blkC], fC_lo0 = psi[32 + 2 * blkC], fC_lo1 = psi[32 + 2 * blkC + 1];
  const double2 fD_hi = psi[64 + blk16], fD_lo0 = psi[128 + 2 * blk16], fD_lo1 = psi[128 + 2 * blk16 + 1];
  const double2 fE_0 = psi[256 + blk16 * 4 + lE], fE_1 = psi[256 + blk16 * 4 + 2 + lE];   // stage of p1: block = (p >> 2)
  double2 cd[4];                                              // lane-pair stage: t on even lanes, (-t.x, t.y) on odd lanes
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double2 t = psi[512 + blk16 * 8 + (e >> 1) * 4 + lE * 2 + (e & 1)];
    cd[e] = l0 ? make_double2(-t.x, t.y) : t;
  }
  // inverse: tw[(index bits below the stage's bit) << (9 - bit)]
  const double sgn = l0 ? -1.0 : 1.0;
  const double2 iE = tw[l0 << 8];
  const int low2 = lE * 2 + l0;                               // (p1, p0) in layout D
  const double2 iD_lo = tw[low2 << 7], iD_hi0 = tw[low2 << 6], iD_hi1 = tw[(4 + low2) << 6];
  const double2 iC_lo = tw[lo4 << 5], iC_hi0 = tw[lo4 << 4], iC_hi1 = tw[(16 + lo4) << 4];
  const int low6 = l4 * 32 + l5 * 16 + lo4;                   // p5..p0 in layout B
  const double2 iB_lo = tw[low6 << 3], iB_hi0 = tw[low6 << 2], iB_hi1 = tw[(64 + low6) << 2];
  const double2 iA_lo = tw[q << 1], iA_hi0 = tw[q], iA_hi1 = tw[256 + q];
  double2 twi[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) twi[e] = twist_inv[q + 256 * e];

  // ---- Fourier key in the engine layout [i][row][col][e8][128] (bsk_to_engine_layout_kernel): the point p of layout E
  // sits at e8 = (p3 p2 p1) = e1 * 4 + l5 * 2 + e0, position (p >> 4) * 2 + p0 = w * 32 + (l & 31) ----
  const __amdgpu_buffer_rsrc_t rs_g =
      __builtin_amdgcn_make_buffer_rsrc((void *)bsk_e, 0, (unsigned)((size_t)n_small * 4 * M * 16), 0x00020000);
  const unsigned voff_g = V2 ? (unsigned)((r * 8 + l1 * 2) * 128 + blk16 * 2 + l0) * 16u : (unsigned)((r * 8 + l5 * 2) * 128 + w * 32 + (l & 31)) * 16u;

  const uint64_t *ks = ks_out + (size_t)blockIdx.x * ks_stride;
  auto modswitch_of = [&](uint64_t v) __attribute__((always_inline)) {
    return (int)((((v >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
  };

  // ---- accumulator: acc = (0, X^{-b~} LUT); registers and image hold -acc ----
  uint64_t nlo[4], nhi[4];
  {
    const uint64_t *lut = lut_polys + (size_t)d.lut_ids[node] * N;
    const int t = (2 * N - modswitch_of(ks[n_small])) & (2 * N - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = q + 256 * e;
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
  __syncthreads();

  constexpr uint64_t ROUND = (uint64_t)1 << (63 - BL);
  int a_next = modswitch_of(ks[0]);
  if (PRIO == 1) { if (r == 0) __builtin_amdgcn_s_setprio(3); }
  if (PRIO == 2) { if (r == 1) __builtin_amdgcn_s_setprio(3); }
  double2 g0[4], g1[4];
  auto load_ggsw = [&](int it) __attribute__((always_inline)) {
    const unsigned so = (unsigned)it * (unsigned)(4 * M * 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned eo = (unsigned)(((e >> 1) * 4 + (e & 1)) * 128 * 16);
      g0[e] = buf_load_d2(rs_g, voff_g, so + (unsigned)(0 * 16 * 128 * 16) + eo);
      g1[e] = buf_load_d2(rs_g, voff_g, so + (unsigned)(1 * 16 * 128 * 16) + eo);
    }
  };
  if (GPF) load_ggsw(0);
  // rotated reads: word indices into the image and sign masks (-1: negate) of the coefficients p - a and p + M - a
  unsigned ri0[4], ri1[4];
  int m0v[4], m1v[4];
  auto rot_prepare = [&](int a) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned s0 = (unsigned)(q + 256 * e - a) & (unsigned)(2 * N - 1);
      const unsigned s1 = (s0 + (unsigned)M) & (unsigned)(2 * N - 1);
      // the image holds -acc: the operand is +acc[s] = -image when the rotation did NOT wrap, -acc[s] = image when it did
      m0v[e] = (int)((s0 >> LOGN) & 1u) - 1;   // -1: negate
      m1v[e] = (int)((s1 >> LOGN) & 1u) - 1;
      ri0[e] = s0 & (unsigned)(N - 1);
      ri1[e] = s1 & (unsigned)(N - 1);
    }
  };
  if (RPRE) rot_prepare(a_next);
  for (int i = 0; i < n_small; ++i) {
    const int a = a_next;
    it_now = i;
    FHES_LAT_STAMP(0);
    double2 x[4];
    // ---- digits of X^a acc - acc at the own coefficients p (re) and p + M (im) ----
    {
      uint64_t va[4], vb[4];
      if (!RPRE) rot_prepare(a);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        va[e] = img[ri0[e]];
        vb[e] = img[ri1[e]];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint64_t dd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m32 = h == 0 ? m0v[e] : m1v[e];
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
    // GGSW rows of this iteration: requested a whole forward transform ahead (GPF: a whole iteration ahead, below)
    if (!GPF) load_ggsw(i);
    // ---- forward transform ----
    FHES_LAT_STAMP(1);
    if (ILS) {
      fwd_butterfly(x[0], x[2], fA_hi);
      fwd_butterfly(x[1], x[3], fA_hi);
      fwd_butterfly(x[0], x[1], fA_lo0);
      x1[sA[0]] = x[0]; x1[sA[1]] = x[1];
      __builtin_amdgcn_sched_barrier(0);
      fwd_butterfly(x[2], x[3], fA_lo1);
      x1[sA[2]] = x[2]; x1[sA[3]] = x[3];
    } else {
    if (LAT_ABL != 2) lat_fwd_pass(x, fA_hi, fA_lo0, fA_lo1);                  // bits 9, 8
    FHES_LAT_STAMP(2);
    if (LAT_ABL != 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x1[sA[e]] = x[e];
    }
    }
    a_next = modswitch_of(ks_next);
    asm volatile("" : "+s"(a_next));
    FHES_LAT_STAMP(3);
    FHES_LAT_HALF_BARRIER();  // B1: the half's own (also: every rotated read of its image is done)
    FHES_LAT_STAMP(4);
    if (LAT_ABL != 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = x1[sB[e]];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (LAT_ABL != 2) lat_fwd_pass(x, fB_hi, fB_lo0, fB_lo1);                  // bits 7, 6
    lat_transpose_e0_l5(x);
    lat_transpose_e1_l4(x);                                                     // -> L_C
    if (ILS && !V2) {
      fwd_butterfly(x[0], x[2], fC_hi);
      fwd_butterfly(x[1], x[3], fC_hi);
      fwd_butterfly(x[0], x[1], fC_lo0);
      x2[sC[0]] = x[0]; x2[sC[1]] = x[1];
      __builtin_amdgcn_sched_barrier(0);
      fwd_butterfly(x[2], x[3], fC_lo1);
      x2[sC[2]] = x[2]; x2[sC[3]] = x[3];
    } else if (LAT_ABL != 2) lat_fwd_pass(x, fC_hi, fC_lo0, fC_lo1);                  // bits 5, 4
    FHES_LAT_STAMP(5);
    if (V2) {
      lat_transpose_e1_l3(x);
      lat_transpose_e0_l2(x);                                                   // -> L_D' (no LDS)
    } else if (LAT_ABL != 1) {
      if (!ILS) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x2[sC[e]] = x[e];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = x2[sD[e]];
      __builtin_amdgcn_sched_barrier(0);
    }
    FHES_LAT_STAMP(6);
    if (LAT_ABL != 2) lat_fwd_pass(x, fD_hi, fD_lo0, fD_lo1);                  // bits 3, 2
    if (V2) lat_transpose_e0_l1(x, odd2); else lat_transpose_e0_l5(x);          // -> L_E
    if (ILS) {
      auto dpp1 = [&](double2 &v, const double2 c) __attribute__((always_inline)) {   // lat_fwd_stage0_dpp on one point
        const double px = dpp_swap1(v.x), py = dpp_swap1(v.y);
        double re = __builtin_fma(c.x, px, v.x);
        re = __builtin_fma(-c.y, py, re);
        double im = __builtin_fma(c.x, py, v.y);
        im = __builtin_fma(c.y, px, im);
        v = make_double2(re, im);
      };
      double2 *sw = s_r + (i & 1) * 2 * M + q;
      fwd_butterfly(x[0], x[1], fE_0);
      dpp1(x[0], cd[0]); dpp1(x[1], cd[1]);
      sw[0] = x[0]; sw[256] = x[1];
      __builtin_amdgcn_sched_barrier(0);
      fwd_butterfly(x[2], x[3], fE_1);
      dpp1(x[2], cd[2]); dpp1(x[3], cd[3]);
      sw[512] = x[2]; sw[768] = x[3];
    } else {
    if (LAT_ABL != 2) {
      fwd_butterfly(x[0], x[1], fE_0);                                          // bit 1
      fwd_butterfly(x[2], x[3], fE_1);
      lat_fwd_stage0_dpp(x, cd);                                                // bit 0
    }
    FHES_LAT_STAMP(7);
    if (LAT_ABL != 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s_r[(i & 1) * 2 * M + e * 256 + q] = x[e];
    }
    }
    FHES_LAT_STAMP(8);
    __syncthreads();  // B3: the one dependence between the halves.  S alternates between two buffers: with the halves free of
                      // each other in between, a half may write its next spectrum while the other still reads this one
    FHES_LAT_STAMP(9);
    // ---- external product: D0 G[0][r] + D1 G[1][r] ----
    if (LAT_ABL == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = ext_mac(x[e], x[e], g0[e], g1[e]);
    } else if (r == 0) {
      double2 oth[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) oth[e] = s_o[(i & 1) * 2 * M + e * 256 + q];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = ext_mac(x[e], oth[e], g0[e], g1[e]);
      asm volatile("; end of form r = 0");
    } else {
      double2 oth[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) oth[e] = s_o[(i & 1) * 2 * M + e * 256 + q];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = ext_mac(oth[e], x[e], g0[e], g1[e]);
      asm volatile("; end of form r = 1");
    }
    FHES_LAT_STAMP(10);
    if (GPF) load_ggsw(i + 1 < n_small ? i + 1 : i);   // (the last request is a repeat: in range, unused)
    // ---- inverse transform ----
    if (LAT_ABL != 2) {
      lat_inv_stage0_dpp(x, sgn);                                               // bit 0
      inv_butterfly(x[0], x[1], iE);                                            // bit 1
      inv_butterfly(x[2], x[3], iE);
    }
    if (V2) lat_transpose_e0_l1(x, odd2); else lat_transpose_e0_l5(x);          // -> L_D
    if (ILS && !V2) {
      inv_butterfly(x[0], x[1], iD_lo);
      inv_butterfly(x[2], x[3], iD_lo);
      inv_butterfly(x[0], x[2], iD_hi0);
      x2[sD[0]] = x[0]; x2[sD[2]] = x[2];
      __builtin_amdgcn_sched_barrier(0);
      inv_butterfly(x[1], x[3], iD_hi1);
      x2[sD[1]] = x[1]; x2[sD[3]] = x[3];
    } else if (LAT_ABL != 2) lat_inv_pass(x, iD_lo, iD_hi0, iD_hi1);                  // bits 2, 3
    FHES_LAT_STAMP(11);
    if (V2) {
      lat_transpose_e0_l2(x);
      lat_transpose_e1_l3(x);                                                   // -> L_C (no LDS)
    } else if (LAT_ABL != 1) {
      if (!ILS) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x2[sD[e]] = x[e];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = x2[sC[e]];
      __builtin_amdgcn_sched_barrier(0);
    }
    FHES_LAT_STAMP(12);
    if (LAT_ABL != 2) lat_inv_pass(x, iC_lo, iC_hi0, iC_hi1);                  // bits 4, 5
    lat_transpose_e0_l5(x);
    lat_transpose_e1_l4(x);                                                     // -> L_B
    if (ILS) {
      inv_butterfly(x[0], x[1], iB_lo);
      inv_butterfly(x[2], x[3], iB_lo);
      inv_butterfly(x[0], x[2], iB_hi0);
      x1[sB[0]] = x[0]; x1[sB[2]] = x[2];
      __builtin_amdgcn_sched_barrier(0);
      inv_butterfly(x[1], x[3], iB_hi1);
      x1[sB[1]] = x[1]; x1[sB[3]] = x[3];
    } else {
    if (LAT_ABL != 2) lat_inv_pass(x, iB_lo, iB_hi0, iB_hi1);                  // bits 6, 7
    FHES_LAT_STAMP(13);
    if (LAT_ABL != 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x1[sB[e]] = x[e];
    }
    }
    FHES_LAT_STAMP(14);
    FHES_LAT_HALF_BARRIER();  // B5: the half's own
    FHES_LAT_STAMP(15);
    if (LAT_ABL != 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = x1[sA[e]];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (LAT_ABL != 2) lat_inv_pass(x, iA_lo, iA_hi0, iA_hi1);                  // bits 8, 9
    FHES_LAT_STAMP(16);
    // ---- untwist (negated factors: -y), round, nacc += rint(-y); the image goes back ----
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double2 v = cmul_conj(x[e], make_double2(-twi[e].x, -twi[e].y));
      nlo[e] = add64(torus_from_f64_pair(v.x), nlo[e]);
      nhi[e] = add64(torus_from_f64_pair(v.y), nhi[e]);
      img[q + 256 * e] = nlo[e];
      img[q + 256 * e + M] = nhi[e];
    }
    if (RPRE) {
      rot_prepare(a_next);
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(ri0[e]), "+v"(ri1[e]), "+v"(m0v[e]), "+v"(m1v[e]));   // materialised HERE
    }
    FHES_LAT_STAMP(17);
    FHES_LAT_HALF_BARRIER();  // B6: the half's own (a half rotates its own polynomial)
    FHES_LAT_STAMP(18);
  }
#undef FHES_LAT_HALF_BARRIER
#undef FHES_LAT_STAMP
  __syncthreads();

  // ---- sample extract from the image of -acc: a'_0 = a_0, a'_j = -a_{N-j}, b' = b_0 ----
  uint64_t *out = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
  for (int j = tid; j < N; j += kLatThreads) {
    const int jj = (j == 0) ? 0 : N - j;
    const uint64_t v = img_all[jj];
    out[j] = (j == 0) ? (uint64_t)0 - v : v;
  }
  if (tid == 0) out[N] = (uint64_t)0 - img_all[N];
}


// ------------------------------------------------------------------------------------------
// Latency build of the MULTI-BIT blind rotate (round 5; grouping factor G = 2, 3; k = 1, one level, N = 2048): the 512-thread
// shape of blind_rotate_lat_kernel -- layouts L_A .. L_E, permlane transposes, every twiddle in registers -- around the step of
// blind_rotate_mb_kernel:
//     bundle = G_0 + sum_{c >= 1} G_c * W_c        W_c[p] = root[deg_c (1 - 4 b_S)] * root[-4 deg_c b_L]   (monomial factors)
//     acc   <- bundle (x) acc                       digits of acc itself; the accumulator is REPLACED, and lives in registers
// No rotation inside the loop, so there is no image of the accumulator and no barrier at the end of a step: the next step's
// first exchange goes into the slots this thread has just read.  Three workgroup barriers per step (A <-> B of both transforms,
// the hand-over of the spectra).  A multi-bit bootstrap is n / G steps against n CMUX iterations of the classic one: with the
// step at the classic latency build's cycles plus the bundle, ONE bootstrap is shorter than a classic one.
// The bundle of step j + 1 is built while step j runs its inverse transform (b0 / b1 are dead behind the external product):
// batches of one point x CB monomials are requested in one slot and consumed in the next.  In layout E a thread's four points
// differ in p3 and p1 while p2 is lane bit 5, so the "uniform" factor of a monomial takes two values per wave: a vector load of
// two addresses (two lines).
// LDS: X double2[2][M] (A <-> B and, wave-local, C <-> D) | S double2[2 parities][2][M] | degrees u16[(n / G)(2^G - 1)].
// ------------------------------------------------------------------------------------------
constexpr size_t lat_mb_lds_bytes(int n_deg) { return (size_t)2 * 1024 * 16 + (size_t)2 * 2 * 1024 * 16 + (size_t)((n_deg + 7) & ~7) * 2; }

template <int BL, int G, bool ILV>
__global__ __launch_bounds__(kLatThreads, 1) void blind_rotate_mb_lat_kernel(
    NodeDesc d, uint64_t *__restrict__ arena, int ct_stride, const uint64_t *__restrict__ ks_out, int ks_stride, int n_small,
    const double2 *__restrict__ bsk_e, const uint64_t *__restrict__ lut_polys, const double2 *__restrict__ twist_inv /* twist / M */,
    const double2 *__restrict__ tw, const double2 *__restrict__ root /* [2N] */, const double2 *__restrict__ psi) {
  constexpr int LOGN = 11, N = 1 << LOGN, M = N / 2, NC = 1 << G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double2 *x1_all = reinterpret_cast<double2 *>(smem_raw);            // [2][M]
  double2 *s_all = x1_all + 2 * M;                                    // [2 parities][2][M]
  uint16_t *degs = reinterpret_cast<uint16_t *>(s_all + 4 * M);

  const int tid = threadIdx.x;
  const int r = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int w = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int q = tid & 255;
  const int l = tid & 63;
  const int l0 = l & 1, l4 = (l >> 4) & 1, l5 = (l >> 5) & 1, lo4 = l & 15;
  const int vidx = d.v0 + blockIdx.x;
  const int inst = vidx / d.count, node = d.first + vidx % d.count;
  const int n_groups = n_small / G;
  double2 *x1 = x1_all + (size_t)r * M;
  double2 *x2 = x1;
  double2 *s_r = s_all + (size_t)r * M;
  const double2 *s_o = s_all + (size_t)(1 - r) * M;

  const uint64_t *ks = ks_out + (size_t)blockIdx.x * ks_stride;
  for (int idx = tid; idx < n_groups * (NC - 1); idx += kLatThreads) {
    const int j = idx / (NC - 1), c = idx % (NC - 1) + 1;
    uint64_t sum = 0;
#pragma unroll
    for (int t = 0; t < G; ++t)
      if ((c >> (G - 1 - t)) & 1) sum += ks[j * G + t];
    degs[idx] = (uint16_t)((((sum >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
  }

  const int blk16 = w * 16 + ((l >> 1) & 15);
  int sA[4], sB[4], sC[4], sD[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sA[e] = q + 256 * e;
    sB[e] = w * 256 + e * 64 + l4 * 32 + l5 * 16 + lo4;
    sC[e] = fswz(w * 256 + l4 * 128 + l5 * 64 + e * 16 + lo4);
    sD[e] = fswz(blk16 * 16 + e * 4 + l5 * 2 + l0);
  }
  // every twiddle of both transforms (as in blind_rotate_lat_kernel)
  const double2 fA_hi = psi[1], fA_lo0 = psi[2], fA_lo1 = psi[3];
  const double2 fB_hi = psi[4 + w], fB_lo0 = psi[8 + 2 * w], fB_lo1 = psi[8 + 2 * w + 1];
  const int blkC = w * 4 + l4 * 2 + l5;
  const double2 fC_hi = psi[16 + blkC], fC_lo0 = psi[32 + 2 * blkC], fC_lo1 = psi[32 + 2 * blkC + 1];
  const double2 fD_hi = psi[64 + blk16], fD_lo0 = psi[128 + 2 * blk16], fD_lo1 = psi[128 + 2 * blk16 + 1];
  const double2 fE_0 = psi[256 + blk16 * 4 + l5], fE_1 = psi[256 + blk16 * 4 + 2 + l5];
  double2 cd[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double2 t = psi[512 + blk16 * 8 + (e >> 1) * 4 + l5 * 2 + (e & 1)];
    cd[e] = l0 ? make_double2(-t.x, t.y) : t;
  }
  const double sgn = l0 ? -1.0 : 1.0;
  const double2 iE = tw[l0 << 8];
  const int low2 = l5 * 2 + l0;
  const double2 iD_lo = tw[low2 << 7], iD_hi0 = tw[low2 << 6], iD_hi1 = tw[(4 + low2) << 6];
  const double2 iC_lo = tw[lo4 << 5], iC_hi0 = tw[lo4 << 4], iC_hi1 = tw[(16 + lo4) << 4];
  const int low6 = l4 * 32 + l5 * 16 + lo4;
  const double2 iB_lo = tw[low6 << 3], iB_hi0 = tw[low6 << 2], iB_hi1 = tw[(64 + low6) << 2];
  const double2 iA_lo = tw[q << 1], iA_hi0 = tw[q], iA_hi1 = tw[256 + q];

  // Fourier key: point p of layout E sits at e8 = (p3 p2 p1) = e1 * 4 + l5 * 2 + e0, position w * 32 + (l & 31) of a polynomial
  // (blind_rotate_lat_kernel); GGSW c of group j: plain layout [c][row][col][e8][128], interleaved [row][col][e8][c][128]
  const __amdgpu_buffer_rsrc_t rs_g =
      __builtin_amdgcn_make_buffer_rsrc((void *)bsk_e, 0, (unsigned)((size_t)n_groups * NC * 4 * M * 16), 0x00020000);
  const unsigned voff_g = ILV ? (unsigned)((r * 8 + l5 * 2) * NC * 128 + w * 32 + (l & 31)) * 16u
                              : (unsigned)((r * 8 + l5 * 2) * 128 + w * 32 + (l & 31)) * 16u;
  auto goff = [&](unsigned so, int c, int row, int e) __attribute__((always_inline)) {
    const int e8 = (e >> 1) * 4 + (e & 1);
    return ILV ? so + (unsigned)(((row * 16 + e8) * NC + c) * 128 * 16) : so + (unsigned)c * (unsigned)(4 * M * 16) + (unsigned)((row * 16 + e8) * 128 * 16);
  };
  // monomial factors: b = bitrev(p); b_S = (p3, p2, p1) << 6 (p2 = lane bit 5 here), b_L = bitrev of the point with those bits cleared
  const int pz = w * 256 + ((l >> 1) & 15) * 16 + l0;
  const int lexp = (-4 * (int)(__brev((unsigned)pz) >> (32 - 10))) & (2 * N - 1);
  auto mono_uniform = [&](int dg, int e) __attribute__((always_inline)) {   // root[deg (1 - 4 b_S)], b_S = ((e1) + 2 l5 + 4 (e0)) << 6
    const int bs4 = (((e >> 1) & 1) + 2 * l5 + 4 * (e & 1)) << 8;
    return root[(dg - ((dg * bs4) & (2 * N - 1))) & (2 * N - 1)];
  };

  // ---- acc = (0, X^{-b~} LUT) in registers: coefficients p and p + M of polynomial r, p = q + 256 e (layout A) ----
  uint64_t own_lo[4], own_hi[4];
  {
    const uint64_t *lut = lut_polys + (size_t)d.lut_ids[node] * N;
    const int bt = (int)((((ks[n_small] >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
    const int t = (2 * N - bt) & (2 * N - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = q + 256 * e;
      uint64_t v0 = 0, v1 = 0;
      if (r == 1) {
        v0 = (t == 0) ? lut[p] : rot_read<N>(lut, t, p);
        v1 = (t == 0) ? lut[p + M] : rot_read<N>(lut, t, p + M);
      }
      own_lo[e] = v0;
      own_hi[e] = v1;
    }
  }
  __syncthreads();   // the degrees are in LDS

  // ---- bundle pipeline: batches of one point x CB monomials, monomial range by monomial range ----
  constexpr int CB = G == 2 ? NC - 1 : 2, NCB = (NC - 1 + CB - 1) / CB, NBATCH = 4 * NCB;   // (g = 3 with CB = 4 spills 19 registers)
  double2 b0[4], b1[4], tg0[CB], tg1[CB], rw[CB], su[CB];
  int deg[NC];
  auto bundle_degrees = [&](int jn) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 1; c < NC; ++c) deg[c] = __builtin_amdgcn_readfirstlane((int)degs[jn * (NC - 1) + (c - 1)]);
  };
  auto bundle_issue = [&](int jn, int k) __attribute__((always_inline)) {
    const int e = k % 4, cb = 1 + (k / 4) * CB;
    const unsigned so = (unsigned)jn * (unsigned)(NC * 4 * M * 16);
    if (e == 0) {
#pragma unroll
      for (int cc = 0; cc < CB; ++cc)
        if (cb + cc < NC) rw[cc] = root[(lexp * deg[cb + cc]) & (2 * N - 1)];
    }
    if (cb == 1) {
      b0[e] = buf_load_d2(rs_g, voff_g, goff(so, 0, 0, e));
      b1[e] = buf_load_d2(rs_g, voff_g, goff(so, 0, 1, e));
    }
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) {
      const int c = cb + cc;
      if (c >= NC) continue;
      su[cc] = mono_uniform(deg[c], e);
      tg0[cc] = buf_load_d2(rs_g, voff_g, goff(so, c, 0, e));
      tg1[cc] = buf_load_d2(rs_g, voff_g, goff(so, c, 1, e));
    }
  };
  auto bundle_consume = [&](int k) __attribute__((always_inline)) {
    const int e = k % 4, cb = 1 + (k / 4) * CB;
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) {
      if (cb + cc >= NC) continue;
      const double2 wv = cmul(su[cc], rw[cc]);
      b0[e] = cfma(tg0[cc], wv, b0[e]);
      b1[e] = cfma(tg1[cc], wv, b1[e]);
    }
  };
  auto bundle_advance = [&](int jn, int first, int count) __attribute__((always_inline)) {
#pragma unroll
    for (int k = first; k < first + count; ++k) {
      __builtin_amdgcn_sched_barrier(0);
      if (k > 0) bundle_consume(k - 1);
      bundle_issue(jn, k);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the batches of a step are spread over five slots of the inverse transform
  constexpr int Q1 = NBATCH / 5 > 0 ? NBATCH / 5 : 1, Q2 = (2 * NBATCH) / 5, Q3 = (3 * NBATCH) / 5, Q4 = (4 * NBATCH) / 5;
  // (Tried and dropped: touching the lines of group j + 2 two steps ahead with one-dword loads -- a lone bootstrap finds its key in the
  // Infinity Cache at best.  The counter of vector loads retires in order, so every real load of the pipeline then waits for the
  // touches in front of it: 2.57 against 2.11 ms (g = 2), 3.0 against 2.37 ms (g = 3); profiles/r05_probe17_*.)
  bundle_degrees(0);
  bundle_advance(0, 0, NBATCH);

  for (int j = 0; j < n_groups; ++j) {
    const int jn = j + 1 < n_groups ? j + 1 : j;   // the last step rebuilds its own bundle (unused)
    double2 x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = make_double2(digit1_f64<BL>(own_lo[e]), digit1_f64<BL>(own_hi[e]));
    // ---- forward transform ----
    fwd_butterfly(x[0], x[2], fA_hi);
    fwd_butterfly(x[1], x[3], fA_hi);
    fwd_butterfly(x[0], x[1], fA_lo0);
    x1[sA[0]] = x[0]; x1[sA[1]] = x[1];
    __builtin_amdgcn_sched_barrier(0);
    fwd_butterfly(x[2], x[3], fA_lo1);
    x1[sA[2]] = x[2]; x1[sA[3]] = x[3];
    __syncthreads();  // B1
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x1[sB[e]];
    __builtin_amdgcn_sched_barrier(0);
    lat_fwd_pass(x, fB_hi, fB_lo0, fB_lo1);
    lat_transpose_e0_l5(x);
    lat_transpose_e1_l4(x);
    fwd_butterfly(x[0], x[2], fC_hi);
    fwd_butterfly(x[1], x[3], fC_hi);
    fwd_butterfly(x[0], x[1], fC_lo0);
    x2[sC[0]] = x[0]; x2[sC[1]] = x[1];
    __builtin_amdgcn_sched_barrier(0);
    fwd_butterfly(x[2], x[3], fC_lo1);
    x2[sC[2]] = x[2]; x2[sC[3]] = x[3];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x2[sD[e]];
    __builtin_amdgcn_sched_barrier(0);
    lat_fwd_pass(x, fD_hi, fD_lo0, fD_lo1);
    lat_transpose_e0_l5(x);
    fwd_butterfly(x[0], x[1], fE_0);
    fwd_butterfly(x[2], x[3], fE_1);
    lat_fwd_stage0_dpp(x, cd);
    {
      double2 *sw = s_r + (j & 1) * 2 * M + q;
      sw[0] = x[0]; sw[256] = x[1]; sw[512] = x[2]; sw[768] = x[3];
    }
    __builtin_amdgcn_sched_barrier(0);
    bundle_consume(NBATCH - 1);   // last batch of this step's bundle
    __syncthreads();  // B3: the spectra meet
    {
      double2 oth[4];
      const double2 *so_ = s_o + (j & 1) * 2 * M + q;
#pragma unroll
      for (int e = 0; e < 4; ++e) oth[e] = so_[e * 256];
      __builtin_amdgcn_sched_barrier(0);
      if (r == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = ext_mac(x[e], oth[e], b0[e], b1[e]);
        asm volatile("; end of form r = 0");
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = ext_mac(oth[e], x[e], b0[e], b1[e]);
        asm volatile("; end of form r = 1");
      }
    }
    bundle_degrees(jn);
    bundle_advance(jn, 0, Q1);
    // ---- inverse transform ----
    lat_inv_stage0_dpp(x, sgn);
    inv_butterfly(x[0], x[1], iE);
    inv_butterfly(x[2], x[3], iE);
    lat_transpose_e0_l5(x);
    inv_butterfly(x[0], x[1], iD_lo);
    inv_butterfly(x[2], x[3], iD_lo);
    inv_butterfly(x[0], x[2], iD_hi0);
    x2[sD[0]] = x[0]; x2[sD[2]] = x[2];
    __builtin_amdgcn_sched_barrier(0);
    inv_butterfly(x[1], x[3], iD_hi1);
    x2[sD[1]] = x[1]; x2[sD[3]] = x[3];
    bundle_advance(jn, Q1, Q2 - Q1);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x2[sC[e]];
    __builtin_amdgcn_sched_barrier(0);
    lat_inv_pass(x, iC_lo, iC_hi0, iC_hi1);
    bundle_advance(jn, Q2, Q3 - Q2);
    lat_transpose_e0_l5(x);
    lat_transpose_e1_l4(x);
    inv_butterfly(x[0], x[1], iB_lo);
    inv_butterfly(x[2], x[3], iB_lo);
    inv_butterfly(x[0], x[2], iB_hi0);
    x1[sB[0]] = x[0]; x1[sB[2]] = x[2];
    __builtin_amdgcn_sched_barrier(0);
    inv_butterfly(x[1], x[3], iB_hi1);
    x1[sB[1]] = x[1]; x1[sB[3]] = x[3];
    bundle_advance(jn, Q3, Q4 - Q3);
    __syncthreads();  // B5
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x1[sA[e]];
    __builtin_amdgcn_sched_barrier(0);
    lat_inv_pass(x, iA_lo, iA_hi0, iA_hi1);
    bundle_advance(jn, Q4, NBATCH - Q4);
    // ---- untwist, round to the torus: the new accumulator (the next step writes into the slots this thread has just read) ----
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double2 v = cmul_conj(x[e], twist_inv[q + 256 * e]);
      own_lo[e] = torus_from_f64_fast(v.x);
      own_hi[e] = torus_from_f64_fast(v.y);
    }
  }

  // ---- sample extract from registers: a'_0 = a_0, a'_j = -a_{N-j}, b' = b_0 ----
  uint64_t *out = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int p = q + 256 * e;
    if (r == 0) {
      if (p == 0) out[0] = own_lo[e];
      else out[N - p] = (uint64_t)0 - own_lo[e];
      out[N - (p + M)] = (uint64_t)0 - own_hi[e];
    } else if (p == 0) {
      out[N] = own_lo[e];
    }
  }
}


// ------------------------------------------------------------------------------------------
// SPLIT multi-bit blind rotate for a handful of bootstraps (round 5): the bundles of ALL steps first, on the whole device, then
// the sequential part alone.
//
// A step of the multi-bit blind rotate is  acc <- bundle_j (x) acc  with  bundle_j = G_0 + sum_c G_c W_c  -- and the bundle depends
// on the key and on the ciphertext's mask (the monomial degrees), NOT on the accumulator.  One call of the reference's API is a chain of
// levels of 3-4 bootstraps on 3-4 of 256 CUs; the other CUs can build every bundle of every step at once:
//   mb_bundle_kernel        grid (n / G groups x 4 quarter-polynomials), 256 threads: a thread holds the 2^G GGSW entries of ONE point in
//                           all four polynomials (the key is read once per launch, whatever the number of ciphertexts) and loops over the
//                           ciphertexts: W_c from two table entries (4.8), bundle = G_0 + sum_c cfma(G_c, W_c) in ascending c -- the
//                           operations of the fused kernels in their order, the same bits -- written in the engine layout of a GGSW;
//   blind_rotate_mb_seq_kernel   the 512-thread latency shape without any bundle work: a step is a CMUX of the classic latency build
//                           (digits of the accumulator in registers, no rotation, no image, three barriers) whose "key" is the
//                           ciphertext's own bundle stream, requested a whole step ahead.
// The chain shrinks from n CMUX iterations (classic) to n / G steps of about the same cost each: the reason multi-bit bootstrapping
// exists, and one this engine had left on the table by building the bundle inside the sequential loop.
// Scratch: n / G x 64 KB per ciphertext (g = 3: 16.3 MB, g = 2: 24.4 MB): launches of up to kSplitMax bootstraps.
// ------------------------------------------------------------------------------------------
constexpr int kSplitMax = 256;

template <int G, bool ILV>
__global__ __launch_bounds__(256) void mb_bundle_kernel(int count, const uint64_t *__restrict__ ks_out, int ks_stride, int n_small,
                                                        const double2 *__restrict__ bsk_e, const double2 *__restrict__ root /* [2N] */,
                                                        double2 *__restrict__ bundles /* [count][n / G][4][M] */) {
  constexpr int LOGN = 11, N = 1 << LOGN, M = N / 2, NC = 1 << G;
  const int j = blockIdx.x >> 2, chunk = blockIdx.x & 3;
  const int n_groups = n_small / G;
  const int i = chunk * 256 + (int)threadIdx.x;          // element of a polynomial in the engine layout: e8 * 128 + q7
  const int e8 = i >> 7, q7 = i & 127;
  const int p = ((q7 >> 1) << 4) | (e8 << 1) | (q7 & 1);  // its spectrum slot (bsk_to_engine_layout_kernel)
  const int b = (int)(__brev((unsigned)p) >> (32 - 10));
  const int bs = b & (7 << 6), bl = b - bs;
  // the 2^G GGSW entries of this element in the four polynomials (rc = row * 2 + column)
  double2 g[NC][4];
  const double2 *grp = bsk_e + (size_t)j * NC * 4 * M;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int rc = 0; rc < 4; ++rc)
      g[c][rc] = ILV ? grp[(size_t)((rc * 8 + e8) * NC + c) * 128 + q7] : grp[(size_t)(c * 4 + rc) * M + i];
  for (int ct = 0; ct < count; ++ct) {
    const uint64_t *ks = ks_out + (size_t)ct * ks_stride + (size_t)j * G;
    double2 acc[4];
#pragma unroll
    for (int rc = 0; rc < 4; ++rc) acc[rc] = g[0][rc];
#pragma unroll
    for (int c = 1; c < NC; ++c) {
      uint64_t sum = 0;
#pragma unroll
      for (int t = 0; t < G; ++t)
        if ((c >> (G - 1 - t)) & 1) sum += ks[t];
      const int dg = (int)((((sum >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
      const double2 a = root[(dg - ((dg * 4 * bs) & (2 * N - 1))) & (2 * N - 1)];   // root[deg (1 - 4 b_S)]
      const double2 wl = root[(-4 * dg * bl) & (2 * N - 1)];                        // root[-4 deg b_L]
      const double2 wv = cmul(a, wl);
#pragma unroll
      for (int rc = 0; rc < 4; ++rc) acc[rc] = cfma(g[c][rc], wv, acc[rc]);
    }
    double2 *dst = bundles + ((size_t)ct * n_groups + j) * 4 * M + i;
#pragma unroll
    for (int rc = 0; rc < 4; ++rc) dst[(size_t)rc * M] = acc[rc];
  }
}

constexpr size_t kSeqLdsBytes = (size_t)2 * 1024 * 16 + (size_t)2 * 2 * 1024 * 16;   // X + S (double-buffered)

template <int BL>
__global__ __launch_bounds__(kLatThreads, 1) void blind_rotate_mb_seq_kernel(
    NodeDesc d, uint64_t *__restrict__ arena, int ct_stride, const uint64_t *__restrict__ ks_out, int ks_stride, int n_small, int n_groups,
    const double2 *__restrict__ bundles, const uint64_t *__restrict__ lut_polys, const double2 *__restrict__ twist_inv /* twist / M */,
    const double2 *__restrict__ tw, const double2 *__restrict__ psi) {
  constexpr int LOGN = 11, N = 1 << LOGN, M = N / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double2 *x1_all = reinterpret_cast<double2 *>(smem_raw);            // [2][M]
  double2 *s_all = x1_all + 2 * M;                                    // [2 parities][2][M]

  const int tid = threadIdx.x;
  const int r = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int w = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int q = tid & 255;
  const int l = tid & 63;
  const int l0 = l & 1, l4 = (l >> 4) & 1, l5 = (l >> 5) & 1, lo4 = l & 15;
  const int vidx = d.v0 + blockIdx.x;
  const int inst = vidx / d.count, node = d.first + vidx % d.count;
  double2 *x1 = x1_all + (size_t)r * M;
  double2 *x2 = x1;
  double2 *s_r = s_all + (size_t)r * M;
  const double2 *s_o = s_all + (size_t)(1 - r) * M;
  const uint64_t *ks = ks_out + (size_t)blockIdx.x * ks_stride;

  const int blk16 = w * 16 + ((l >> 1) & 15);
  int sA[4], sB[4], sC[4], sD[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sA[e] = q + 256 * e;
    sB[e] = w * 256 + e * 64 + l4 * 32 + l5 * 16 + lo4;
    sC[e] = fswz(w * 256 + l4 * 128 + l5 * 64 + e * 16 + lo4);
    sD[e] = fswz(blk16 * 16 + e * 4 + l5 * 2 + l0);
  }
  const double2 fA_hi = psi[1], fA_lo0 = psi[2], fA_lo1 = psi[3];
  const double2 fB_hi = psi[4 + w], fB_lo0 = psi[8 + 2 * w], fB_lo1 = psi[8 + 2 * w + 1];
  const int blkC = w * 4 + l4 * 2 + l5;
  const double2 fC_hi = psi[16 + blkC], fC_lo0 = psi[32 + 2 * blkC], fC_lo1 = psi[32 + 2 * blkC + 1];
  const double2 fD_hi = psi[64 + blk16], fD_lo0 = psi[128 + 2 * blk16], fD_lo1 = psi[128 + 2 * blk16 + 1];
  const double2 fE_0 = psi[256 + blk16 * 4 + l5], fE_1 = psi[256 + blk16 * 4 + 2 + l5];
  double2 cd[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double2 t = psi[512 + blk16 * 8 + (e >> 1) * 4 + l5 * 2 + (e & 1)];
    cd[e] = l0 ? make_double2(-t.x, t.y) : t;
  }
  const double sgn = l0 ? -1.0 : 1.0;
  const double2 iE = tw[l0 << 8];
  const int low2 = l5 * 2 + l0;
  const double2 iD_lo = tw[low2 << 7], iD_hi0 = tw[low2 << 6], iD_hi1 = tw[(4 + low2) << 6];
  const double2 iC_lo = tw[lo4 << 5], iC_hi0 = tw[lo4 << 4], iC_hi1 = tw[(16 + lo4) << 4];
  const int low6 = l4 * 32 + l5 * 16 + lo4;
  const double2 iB_lo = tw[low6 << 3], iB_hi0 = tw[low6 << 2], iB_hi1 = tw[(64 + low6) << 2];
  const double2 iA_lo = tw[q << 1], iA_hi0 = tw[q], iA_hi1 = tw[256 + q];
  double2 twi[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) twi[e] = twist_inv[q + 256 * e];

  // this ciphertext's bundle stream: [step][row][col][e8][128] -- the engine layout of a GGSW, addressed as the classic latency build does
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(bundles + (size_t)blockIdx.x * n_groups * 4 * M), 0, (unsigned)((size_t)n_groups * 4 * M * 16), 0x00020000);
  const unsigned voff_b = (unsigned)((r * 8 + l5 * 2) * 128 + w * 32 + (l & 31)) * 16u;
  double2 b0[4], b1[4];
  auto load_bundle = [&](int j) __attribute__((always_inline)) {
    const unsigned so = (unsigned)j * (unsigned)(4 * M * 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned eo = (unsigned)(((e >> 1) * 4 + (e & 1)) * 128 * 16);
      b0[e] = buf_load_d2(rs_b, voff_b, so + eo);
      b1[e] = buf_load_d2(rs_b, voff_b, so + (unsigned)(16 * 128 * 16) + eo);
    }
  };

  uint64_t own_lo[4], own_hi[4];
  {
    const uint64_t *lut = lut_polys + (size_t)d.lut_ids[node] * N;
    const int bt = (int)((((ks[n_small] >> (64 - LOGN - 2)) + 1) >> 1) & (uint64_t)(2 * N - 1));
    const int t = (2 * N - bt) & (2 * N - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = q + 256 * e;
      uint64_t v0 = 0, v1 = 0;
      if (r == 1) {
        v0 = (t == 0) ? lut[p] : rot_read<N>(lut, t, p);
        v1 = (t == 0) ? lut[p + M] : rot_read<N>(lut, t, p + M);
      }
      own_lo[e] = v0;
      own_hi[e] = v1;
    }
  }
  load_bundle(0);

  for (int j = 0; j < n_groups; ++j) {
    double2 x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = make_double2(digit1_f64<BL>(own_lo[e]), digit1_f64<BL>(own_hi[e]));
    // ---- forward transform ----
    fwd_butterfly(x[0], x[2], fA_hi);
    fwd_butterfly(x[1], x[3], fA_hi);
    fwd_butterfly(x[0], x[1], fA_lo0);
    x1[sA[0]] = x[0]; x1[sA[1]] = x[1];
    __builtin_amdgcn_sched_barrier(0);
    fwd_butterfly(x[2], x[3], fA_lo1);
    x1[sA[2]] = x[2]; x1[sA[3]] = x[3];
    __syncthreads();  // B1
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x1[sB[e]];
    __builtin_amdgcn_sched_barrier(0);
    lat_fwd_pass(x, fB_hi, fB_lo0, fB_lo1);
    lat_transpose_e0_l5(x);
    lat_transpose_e1_l4(x);
    fwd_butterfly(x[0], x[2], fC_hi);
    fwd_butterfly(x[1], x[3], fC_hi);
    fwd_butterfly(x[0], x[1], fC_lo0);
    x2[sC[0]] = x[0]; x2[sC[1]] = x[1];
    __builtin_amdgcn_sched_barrier(0);
    fwd_butterfly(x[2], x[3], fC_lo1);
    x2[sC[2]] = x[2]; x2[sC[3]] = x[3];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x2[sD[e]];
    __builtin_amdgcn_sched_barrier(0);
    lat_fwd_pass(x, fD_hi, fD_lo0, fD_lo1);
    lat_transpose_e0_l5(x);
    {
      auto dpp1 = [&](double2 &v, const double2 c) __attribute__((always_inline)) {
        const double px = dpp_swap1(v.x), py = dpp_swap1(v.y);
        double re = __builtin_fma(c.x, px, v.x);
        re = __builtin_fma(-c.y, py, re);
        double im = __builtin_fma(c.x, py, v.y);
        im = __builtin_fma(c.y, px, im);
        v = make_double2(re, im);
      };
      double2 *sw = s_r + (j & 1) * 2 * M + q;
      fwd_butterfly(x[0], x[1], fE_0);
      dpp1(x[0], cd[0]); dpp1(x[1], cd[1]);
      sw[0] = x[0]; sw[256] = x[1];
      __builtin_amdgcn_sched_barrier(0);
      fwd_butterfly(x[2], x[3], fE_1);
      dpp1(x[2], cd[2]); dpp1(x[3], cd[3]);
      sw[512] = x[2]; sw[768] = x[3];
    }
    __syncthreads();  // B3: the spectra meet
    {
      double2 oth[4];
      const double2 *so_ = s_o + (j & 1) * 2 * M + q;
#pragma unroll
      for (int e = 0; e < 4; ++e) oth[e] = so_[e * 256];
      __builtin_amdgcn_sched_barrier(0);
      if (r == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = ext_mac(x[e], oth[e], b0[e], b1[e]);
        asm volatile("; end of form r = 0");
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = ext_mac(oth[e], x[e], b0[e], b1[e]);
        asm volatile("; end of form r = 1");
      }
    }
    load_bundle(j + 1 < n_groups ? j + 1 : j);   // a whole step ahead (the last request is a repeat: in range, unused)
    // ---- inverse transform ----
    lat_inv_stage0_dpp(x, sgn);
    inv_butterfly(x[0], x[1], iE);
    inv_butterfly(x[2], x[3], iE);
    lat_transpose_e0_l5(x);
    inv_butterfly(x[0], x[1], iD_lo);
    inv_butterfly(x[2], x[3], iD_lo);
    inv_butterfly(x[0], x[2], iD_hi0);
    x2[sD[0]] = x[0]; x2[sD[2]] = x[2];
    __builtin_amdgcn_sched_barrier(0);
    inv_butterfly(x[1], x[3], iD_hi1);
    x2[sD[1]] = x[1]; x2[sD[3]] = x[3];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x2[sC[e]];
    __builtin_amdgcn_sched_barrier(0);
    lat_inv_pass(x, iC_lo, iC_hi0, iC_hi1);
    lat_transpose_e0_l5(x);
    lat_transpose_e1_l4(x);
    inv_butterfly(x[0], x[1], iB_lo);
    inv_butterfly(x[2], x[3], iB_lo);
    inv_butterfly(x[0], x[2], iB_hi0);
    x1[sB[0]] = x[0]; x1[sB[2]] = x[2];
    __builtin_amdgcn_sched_barrier(0);
    inv_butterfly(x[1], x[3], iB_hi1);
    x1[sB[1]] = x[1]; x1[sB[3]] = x[3];
    __syncthreads();  // B5
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x1[sA[e]];
    __builtin_amdgcn_sched_barrier(0);
    lat_inv_pass(x, iA_lo, iA_hi0, iA_hi1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double2 v = cmul_conj(x[e], twi[e]);
      own_lo[e] = torus_from_f64_fast(v.x);
      own_hi[e] = torus_from_f64_fast(v.y);
    }
  }

  uint64_t *out = arena + (size_t)resolve(d, inst, d.out_idx[node]) * ct_stride;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int p = q + 256 * e;
    if (r == 0) {
      if (p == 0) out[0] = own_lo[e];
      else out[N - p] = (uint64_t)0 - own_lo[e];
      out[N - (p + M)] = (uint64_t)0 - own_hi[e];
    } else if (p == 0) {
      out[N] = own_lo[e];
    }
  }
}

}  // namespace dev
}  // namespace fhes
