// micro-benchmark (round 4, the gate of VERDICT r3 item 1): do ds_write_b128 overlap with f64 arithmetic when the
// stores of a phase are spread between the arithmetic of the SAME wave instead of leaving as one burst?
// Measured at the blind rotate's own ratio per barrier-separated phase: NF x 8 v_fma_f64, NW ds_write_b128,
// NR ds_read_b128 per wave.  Structures (all with the same work per phase and wave):
//   I  = no barriers, stores and loads spread evenly over the FMAs (mode 2 of lds_valu_overlap.hip, generalised)
//   B  = [load burst, wait, all FMAs, store burst, wait, barrier]                 (the shipped kernel's pass)
//   T<k> = [load burst, wait, FMAs with one store behind each of the LAST k FMA8 blocks ..., wait, barrier]
//          (what a hand-scheduled pass can do: a butterfly's results leave while the next butterfly computes;
//           only the last register stage of a pass produces results, so k is bounded by ~1/3 of the FMAs)
//   H<k> = T<k> and the loads of the NEXT phase issued behind the barrier are waited for only after the first
//          k FMA8 blocks (those model work that does not depend on the exchange: twiddle products, digits)
// Launch shapes: "1x512" = one 512-thread workgroup per CU (8 waves in lockstep: the worst case),
//                "2x256" = two 256-thread workgroups per CU, 64 KB LDS each (the shipped kernel's occupancy).
//   hipcc --offload-arch=gfx950 -O3 -o lds_valu_gate lds_valu_gate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


template <int OFF> __device__ __forceinline__ void wr(unsigned addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void rd(u32x4 &d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
}
#define WAIT0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

enum { S_I = 0, S_B = 1, S_T = 2, S_H = 3 };
constexpr int ROW = 4096;  // 14 rows x 4 KB = 56 KB window (512 threads wrap onto 256 slots: different waves, no conflicts)

template <int LO, int HI> __device__ __forceinline__ void emit_wr(unsigned addr, u32x4 v) {
  if constexpr (LO < HI) { wr<LO * ROW>(addr, v); emit_wr<LO + 1, HI>(addr, v); }
}
template <int LO, int HI> __device__ __forceinline__ void emit_rd(u32x4 *r, unsigned addr) {
  if constexpr (LO < HI) { rd<LO * ROW>(r[LO & 3], addr); emit_rd<LO + 1, HI>(r, addr); }
}
constexpr int ceil_div(int a, int b) { return b == 0 ? 0 : (a + b - 1) / b; }

#define FMA8X asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b))

// block F of a phase and everything that follows it
template <int STRUCT, int NW, int NR, int NF, int TAIL, int F>
__device__ __forceinline__ void blocks(double (&x)[8], u32x4 *r, unsigned addr, u32x4 wv, double a, double b) {
  if constexpr (F < NF) {
    if constexpr (STRUCT == S_H && F == TAIL) WAIT0;
    FMA8X;
    if constexpr (STRUCT == S_I) {
      emit_wr<ceil_div(F * NW, NF), ceil_div((F + 1) * NW, NF)>(addr, wv);
      emit_rd<ceil_div(F * NR, NF), ceil_div((F + 1) * NR, NF)>(r, addr);
    } else if constexpr (STRUCT != S_B && F >= NF - TAIL) {
      constexpr int J = F - (NF - TAIL);
      emit_wr<ceil_div(J * NW, TAIL), ceil_div((J + 1) * NW, TAIL)>(addr, wv);
    }
    blocks<STRUCT, NW, NR, NF, TAIL, F + 1>(x, r, addr, wv, a, b);
  }
}

template <int STRUCT, int NW, int NR, int NF, int TAIL, int THREADS>
__global__ __launch_bounds__(THREADS) void k(unsigned long long *stamps, double *sink, int iters, double a, double b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + 1.0 + i;
  const unsigned addr = (threadIdx.x & 255) * 16u;
  u32x4 wv = {threadIdx.x, 1u, 2u, 3u};
  u32x4 r[4] = {};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; it++) {
    if constexpr (STRUCT == S_I) {
      blocks<STRUCT, NW, NR, NF, TAIL, 0>(x, r, addr, wv, a, b);
      WAIT0;
    } else {
      emit_rd<0, NR>(r, addr);
      if constexpr (STRUCT != S_H || NF == 0) WAIT0;
      blocks<STRUCT, NW, NR, NF, TAIL, 0>(x, r, addr, wv, a, b);
      if constexpr (STRUCT == S_B || NF == 0) emit_wr<0, NW>(addr, wv);
      WAIT0;
      __builtin_amdgcn_s_barrier();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = t1 - t0;
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7] + (double)(r[0].x + r[1].y + r[2].z + r[3].w);
}

// K: the shipped kernel's own phase structure per CMUX iteration, with its instruction counts (profiles/EXPERIMENTS.md, census of round 4):
//   B6->B1  20 loads | 88 f64 + 157 other VALU | 8 stores | barrier
//   B1->B3   8 loads | 72 f64 | 8 stores, 8 loads (wave-local exchange) | 72 f64 | 32 f64 + 32 dpp-class | 8 loads (twiddles) | 8 stores | barrier
//   B3->B5   8 loads | 64 f64 | 32 f64 + 32 dpp-class | 72 f64 | 8 stores, 8 loads | 72 f64 | 8 stores, 6 loads | barrier
//   B5->B6   8 loads | 184 f64 + 48 other | 8 stores | barrier
// VARIANT 0: as shipped; 1: the same work with every exchange fused into the phase's ends (one load burst at the start, one store burst at the
// end of each barrier phase: what the phases would cost WITHOUT the extra LDS round trips inside them); 2: as 0 without any LDS operation; 3: as
// 0 without any arithmetic.
#define INT8 asm volatile("v_xor_b32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]) : "v"(yk))
template <int N> __device__ __forceinline__ void fma_blocks(double (&x)[8], double a, double b) {
  if constexpr (N > 0) { FMA8X; fma_blocks<N - 1>(x, a, b); }
}
template <int N> __device__ __forceinline__ void int_blocks(unsigned (&y)[8], unsigned yk) {
  if constexpr (N > 0) { INT8; int_blocks<N - 1>(y, yk); }
}
// eight half-wave / row exchanges of register pairs (gfx950: v_permlane32_swap, v_permlane16_swap; two wait states after a VALU write of an operand)
#define SWAP8 asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n s_nop 1\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]))
template <int N> __device__ __forceinline__ void swap_blocks(unsigned (&y)[8]) {
  if constexpr (N > 0) { SWAP8; swap_blocks<N - 1>(y); }
}
template <int VARIANT>
__global__ __launch_bounds__(VARIANT == 9 ? 512 : (VARIANT >= 10 ? 128 : 256)) void kk(unsigned long long *stamps, double *sink, int iters, double a, double b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  double x[8];
  unsigned y[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { x[i] = threadIdx.x + 1.0 + i; y[i] = threadIdx.x + i; }
  const unsigned yk = threadIdx.x | 1u;
  const unsigned addr = (threadIdx.x & 255) * 16u;
  u32x4 wv = {threadIdx.x, 1u, 2u, 3u};
  u32x4 r[4] = {};
  constexpr bool LDS = VARIANT != 2 && VARIANT != 12, ALU = VARIANT != 3 && VARIANT != 13, FUSED = VARIANT == 1;
  // VARIANT 4: the same totals in four IDENTICAL phases (one load burst, one store burst each); 5: as shipped with a workgroup barrier at the
  // two wave-local exchanges as well (six phases); 6: four identical phases with the loads and stores split into two round trips each
#define K_RD(n) do { if constexpr (LDS) { emit_rd<0, n>(r, addr); } } while (0)
#define K_WR(n) do { if constexpr (LDS) { emit_wr<0, n>(addr, wv); } } while (0)
#define K_WAIT do { if constexpr (LDS) { WAIT0; } } while (0)
#define K_F(n) do { if constexpr (ALU) { fma_blocks<n>(x, a, b); } } while (0)
#define K_I(n) do { if constexpr (ALU) { int_blocks<n>(y, yk); } } while (0)
#define K_S(n) do { if constexpr (VARIANT != 8) { swap_blocks<n>(y); } } while (0)
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; it++) {
    if constexpr (VARIANT >= 10) {
      // round 5, VERDICT r4 item 7: ONE WAVE PER POLYNOMIAL -- a bootstrap on 128 threads, 16 complex points per lane, passes of 4 | 4 | 2 index
      // bits, every transform exchange local to its wave (s_waitcnt, no barrier), no lane-pair stage; two 2-wave barriers per CMUX (spectrum
      // share, image).  Per wave and CMUX: 1232 f64 (2 x 656 - the lane-pair stage's 2 x 16 duplicated products - rounding of the blocks),
      // 408 other VALU (2 x 221 without the DPP moves; addresses of the 16-byte rotated reads as in the shipped kernel), 96 stores, 147 loads
      // (40 rotated / own, 5 x 16 exchange, 15 + 12 table twiddles of passes 2 and 3; those of pass 1 are wave-uniform).
      // 10: bursts as the shipped kernel issues them; 11: the stores of an exchange behind the last blocks of its pass and the loads of the
      // next pass in two halves (hand-scheduled, optimistic); 12: 10 without LDS operations; 13: 10 without arithmetic
      if constexpr (VARIANT != 11) {
        K_RD(10); K_RD(10); K_RD(10); K_RD(10); K_WAIT; K_I(39); K_F(4); K_F(24); K_WR(16); K_WAIT;
        K_RD(16); K_RD(15); K_WAIT; K_F(24); K_WR(16); K_WAIT;
        K_RD(16); K_RD(12); K_WAIT; K_F(12); K_WR(16); K_WAIT; __builtin_amdgcn_s_barrier();
        K_RD(16); K_WAIT; K_F(16); K_F(12); K_WR(16); K_WAIT;
        K_RD(16); K_WAIT; K_F(24); K_WR(16); K_WAIT;
        K_RD(16); K_WAIT; K_F(24); K_F(14); K_I(12); K_WR(16); K_WAIT; __builtin_amdgcn_s_barrier();
      } else {
        K_RD(10); K_RD(10); K_WAIT; K_I(20); K_RD(10); K_RD(10); K_F(2); K_F(12); K_WAIT; K_I(19); K_F(2); K_F(4); K_F(4); K_WR(8); K_F(4); K_WR(8); K_RD(15); K_WAIT;
        K_RD(8); K_WAIT; K_RD(8); K_F(12); K_WAIT; K_F(4); K_F(4); K_WR(8); K_F(4); K_WR(8); K_RD(12); K_WAIT;
        K_RD(8); K_WAIT; K_RD(8); K_F(6); K_WAIT; K_F(3); K_WR(8); K_F(3); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
        K_RD(8); K_WAIT; K_RD(8); K_F(8); K_WAIT; K_F(8); K_F(6); K_WR(8); K_F(6); K_WR(8); K_WAIT;
        K_RD(8); K_WAIT; K_RD(8); K_F(12); K_WAIT; K_F(4); K_F(4); K_WR(8); K_F(4); K_WR(8); K_WAIT;
        K_RD(8); K_WAIT; K_RD(8); K_F(12); K_WAIT; K_F(12); K_F(7); K_I(6); K_WR(8); K_F(7); K_I(6); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      }
    } else if constexpr (VARIANT == 9) {
      // what a LATENCY build could be worth: ONE bootstrap on 512 threads -- every wave does half of a shipped wave's work in the same four
      // phases (two waves per SIMD overlap their LDS and VALU phases); optimistic: a 4-point-per-thread transform needs more exchanges
      K_RD(10); K_WAIT; K_F(6); K_I(10); K_WR(4); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(4); K_WAIT; K_F(5); K_WR(4); K_RD(4); K_WAIT; K_F(5); K_F(2); K_I(2); K_RD(4); K_WAIT; K_WR(4); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(4); K_WAIT; K_F(4); K_F(2); K_I(2); K_F(5); K_WR(4); K_RD(4); K_WAIT; K_F(5); K_WR(4); K_RD(3); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(4); K_WAIT; K_F(12); K_I(3); K_WR(4); K_WAIT; __builtin_amdgcn_s_barrier();
    } else if constexpr (VARIANT == 7 || VARIANT == 8) {
      // a DIFFERENT decomposition of the same transforms: register bits exchanged with lane bits 4 and 5 by v_permlane16/32_swap (16 swaps per
      // bit and 8 points), so the passes are 3 | swap | 2 | LDS | 3 | swap | 2 and back: one LDS exchange per direction instead of two, no
      // wave-local LDS round trip, no DPP lane-pair stage.  32 stores, ~69 loads, the same f64 count, 128 swaps - 64 DPP moves.  8: without the swaps
      K_RD(10); K_RD(10); K_RD(3); K_WAIT; K_F(11); K_I(20); K_S(4); K_F(6); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(8); K_RD(8); K_WAIT; K_F(9); K_S(4); K_F(6); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(8); K_RD(8); K_WAIT; K_F(8); K_F(5); K_S(4); K_F(9); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(8); K_RD(6); K_WAIT; K_F(6); K_S(4); K_F(26); K_I(6); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
    } else if constexpr (VARIANT == 4) {
#pragma unroll
      for (int ph = 0; ph < 4; ph++) {
        K_RD(10); K_RD(9); K_WAIT; K_F(21); K_I(8); K_WR(8); K_WR(4); K_WAIT; __builtin_amdgcn_s_barrier();
      }
    } else if constexpr (VARIANT == 6) {
#pragma unroll
      for (int ph = 0; ph < 4; ph++) {
        K_RD(10); K_WAIT; K_F(10); K_I(4); K_WR(6); K_RD(9); K_WAIT; K_F(11); K_I(4); K_WR(6); K_WAIT; __builtin_amdgcn_s_barrier();
      }
    } else if constexpr (VARIANT == 5) {
      K_RD(10); K_RD(10); K_WAIT; K_F(11); K_I(20); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(8); K_WAIT; K_F(9); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier(); K_RD(8); K_WAIT; K_F(9); K_F(4); K_I(4); K_RD(8); K_WAIT; K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(8); K_WAIT; K_F(8); K_F(4); K_I(4); K_F(9); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier(); K_RD(8); K_WAIT; K_F(9); K_WR(8); K_RD(6); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(8); K_WAIT; K_F(23); K_I(6); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
    } else if constexpr (!FUSED) {
      K_RD(10); K_RD(10); K_WAIT; K_F(11); K_I(20); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(8); K_WAIT; K_F(9); K_WR(8); K_RD(8); K_WAIT; K_F(9); K_F(4); K_I(4); K_RD(8); K_WAIT; K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(8); K_WAIT; K_F(8); K_F(4); K_I(4); K_F(9); K_WR(8); K_RD(8); K_WAIT; K_F(9); K_WR(8); K_RD(6); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(8); K_WAIT; K_F(23); K_I(6); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
    } else {
      K_RD(10); K_RD(10); K_WAIT; K_F(11); K_I(20); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(12); K_RD(12); K_WAIT; K_F(22); K_I(4); K_WR(8); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(11); K_RD(11); K_WAIT; K_F(30); K_I(4); K_WR(8); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
      K_RD(8); K_WAIT; K_F(23); K_I(6); K_WR(8); K_WAIT; __builtin_amdgcn_s_barrier();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7] + (double)(r[0].x + r[1].y + r[2].z + r[3].w) + (double)(y[0] ^ y[1] ^ y[2] ^ y[3] ^ y[4] ^ y[5] ^ y[6] ^ y[7]);
}

template <int VARIANT>
static void run_k(unsigned long long *d_st, double *d_sink, int n_cu, int wg_per_cu) {
  const int iters = 5000;
  auto kern = kk<VARIANT>;
  // the allocation decides how many workgroups a CU holds (64 KB -> 2, 52 KB -> 3, 40 KB -> 4); variants 0 and 5 touch rows 0-9 (40 KB) only
  const int lds_bytes = wg_per_cu <= 2 ? 65536 : (wg_per_cu == 3 ? 49152 : 38912);  // 4 per CU: reads of row 9 run past 38 KB and are dropped by the hardware (timing only)
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(n_cu * wg_per_cu), dim3(VARIANT == 9 ? 512 : 256), lds_bytes, 0, d_st, d_sink, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> st((size_t)n_cu * wg_per_cu * (VARIANT == 9 ? 8 : 4));
  hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc;
  for (size_t i = 0; i < st.size(); i++) cyc.push_back((double)st[i] / iters);
  std::sort(cyc.begin(), cyc.end());
  int resident = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, reinterpret_cast<const void *>(kern), 256, lds_bytes);
  static const char *names[] = {"as shipped", "exchanges fused into the phase ends", "no LDS operations", "no arithmetic",
                                "four identical phases, one round trip each", "as shipped + barriers at the wave-local exchanges (six phases)",
                                "four identical phases, two round trips each",
                                "permlane decomposition: 3 | swap | 2 | LDS | 3 | swap | 2 (32 stores, 69 loads, 128 swaps)", "permlane decomposition without the swaps",
                                "latency build: one bootstrap on 512 threads, half the work per wave (optimistic)"};
  printf("{\"structure\": \"K\", \"variant\": \"%s\", \"workgroups_per_cu\": %d, \"cycles_per_iteration_median_wave\": %.1f, "
         "\"cycles_per_iteration_slowest_wave\": %.1f, \"wall_us_per_iteration\": %.4f, \"cycles_per_cmux_per_cu\": %.0f, "
         "\"implied_clock_ghz\": %.3f, \"resident_workgroups_per_cu_by_occupancy_api\": %d}\n",
         names[VARIANT], wg_per_cu, cyc[cyc.size() / 2], cyc.back(), ms * 1e3 / iters, cyc[cyc.size() / 2] / wg_per_cu,
         cyc[cyc.size() / 2] / (ms * 1e3 / iters) / 1e3, resident);
  fflush(stdout);
}

// one wave per polynomial: 128-thread workgroups (one bootstrap each), `per_cu` of them resident by their LDS allocation
template <int VARIANT>
static void run_w(unsigned long long *d_st, double *d_sink, int n_cu, int per_cu) {
  const int iters = 5000;
  auto kern = kk<VARIANT>;
  const int lds_bytes = per_cu <= 2 ? 65536 : (per_cu == 3 ? 53248 : (per_cu == 4 ? 40960 : 20480));   // timing only: reads past the allocation are dropped
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(n_cu * per_cu), dim3(128), lds_bytes, 0, d_st, d_sink, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> st((size_t)n_cu * per_cu * 2);
  hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc;
  for (size_t i = 0; i < st.size(); i++) cyc.push_back((double)st[i] / iters);
  std::sort(cyc.begin(), cyc.end());
  int resident = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, reinterpret_cast<const void *>(kern), 128, lds_bytes);
  static const char *names[] = {"bursts", "stores behind the last blocks of a pass, loads in halves", "no LDS operations", "no arithmetic"};
  printf("{\"structure\": \"W\", \"variant\": \"one wave per polynomial, %s\", \"bootstraps_per_cu\": %d, \"cycles_per_iteration_median_wave\": %.1f, "
         "\"cycles_per_iteration_slowest_wave\": %.1f, \"wall_us_per_iteration\": %.4f, \"cycles_per_cmux_per_cu\": %.0f, "
         "\"cycles_per_cmux_per_cu_by_wall_clock_at_2p4ghz\": %.0f, \"resident_workgroups_per_cu_by_occupancy_api\": %d}\n",
         names[VARIANT - 10], per_cu, cyc[cyc.size() / 2], cyc.back(), ms * 1e3 / iters, cyc[cyc.size() / 2] / per_cu,
         ms * 1e3 / iters * 2400.0 / per_cu, resident);
  fflush(stdout);
}

template <int STRUCT, int NW, int NR, int NF, int TAIL, int THREADS>
static void run(unsigned long long *d_st, double *d_sink, int n_cu) {
  const int iters = 20000;
  auto kern = k<STRUCT, NW, NR, NF, TAIL, THREADS>;
  const int n_blocks = n_cu * (512 / THREADS);
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(THREADS), 65536, 0, d_st, d_sink, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> st((size_t)n_cu * 8);
  hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc;
  for (size_t i = 0; i < st.size(); i++) cyc.push_back((double)st[i] / iters);
  std::sort(cyc.begin(), cyc.end());
  static const char *names[] = {"I", "B", "T", "H"};
  printf("{\"structure\": \"%s\", \"tail_blocks\": %d, \"shape\": \"%s\", \"writes_b128\": %d, \"reads_b128\": %d, \"fma_f64\": %d, "
         "\"cycles_per_phase_median_wave\": %.1f, \"cycles_per_phase_max_wave\": %.1f, \"wall_ms\": %.3f}\n",
         names[STRUCT], TAIL, THREADS == 512 ? "1x512" : "2x256", NW, NR, NF * 8, cyc[cyc.size() / 2], cyc.back(), ms);
  fflush(stdout);
}

int main() {
  int n_cu = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) == hipSuccess) n_cu = prop.multiProcessorCount;
  unsigned long long *d_st;
  double *d_sink;
  hipMalloc(&d_st, (size_t)(n_cu + 1) * 32 * 8);       // up to four 256-thread workgroups per CU (16 waves)
  hipMalloc(&d_sink, (size_t)(n_cu + 1) * 1024 * 8);
  if (getenv("GATE_WAVE_PER_POLY")) {
    run_k<0>(d_st, d_sink, n_cu, 2);
    for (int per_cu = 2; per_cu <= 4; per_cu++) {
      run_w<10>(d_st, d_sink, n_cu, per_cu);
      run_w<11>(d_st, d_sink, n_cu, per_cu);
    }
    run_w<12>(d_st, d_sink, n_cu, 2);
    run_w<13>(d_st, d_sink, n_cu, 2);
    run_w<12>(d_st, d_sink, n_cu, 4);
    run_w<13>(d_st, d_sink, n_cu, 4);
    run_w<10>(d_st, d_sink, n_cu, 8);
    run_w<11>(d_st, d_sink, n_cu, 8);
    return 0;
  }
  // VALU only / LDS only references
  run<S_B, 0, 0, 20, 0, 512>(d_st, d_sink, n_cu);
  run<S_B, 0, 0, 20, 0, 256>(d_st, d_sink, n_cu);
  run<S_B, 8, 14, 0, 0, 512>(d_st, d_sink, n_cu);
  run<S_B, 8, 14, 0, 0, 256>(d_st, d_sink, n_cu);
  // the gate: interleaved in every wave at the kernel's ratio, against the burst
  run<S_I, 8, 14, 20, 0, 512>(d_st, d_sink, n_cu);
  run<S_B, 8, 14, 20, 0, 512>(d_st, d_sink, n_cu);
  run<S_I, 8, 14, 20, 0, 256>(d_st, d_sink, n_cu);
  run<S_B, 8, 14, 20, 0, 256>(d_st, d_sink, n_cu);
  // what a hand-scheduled pass can do: stores behind the last k FMA8 blocks, barriers kept
  run<S_T, 8, 14, 20, 8, 512>(d_st, d_sink, n_cu);
  run<S_T, 8, 14, 20, 8, 256>(d_st, d_sink, n_cu);
  run<S_T, 8, 14, 20, 4, 256>(d_st, d_sink, n_cu);
  run<S_T, 8, 14, 20, 2, 256>(d_st, d_sink, n_cu);
  run<S_T, 8, 14, 20, 16, 256>(d_st, d_sink, n_cu);
  run<S_H, 8, 14, 20, 8, 256>(d_st, d_sink, n_cu);
  run<S_H, 8, 14, 20, 4, 256>(d_st, d_sink, n_cu);
  // the exchanges alone (8 loads) at the same arithmetic
  run<S_B, 8, 8, 20, 0, 256>(d_st, d_sink, n_cu);
  run<S_T, 8, 8, 20, 8, 256>(d_st, d_sink, n_cu);
  run<S_H, 8, 8, 20, 8, 256>(d_st, d_sink, n_cu);
  // a shorter phase (the lane-pair stage / the multiply-accumulate: ~64 FMAs around an exchange)
  run<S_B, 8, 8, 8, 0, 256>(d_st, d_sink, n_cu);
  run<S_T, 8, 8, 8, 4, 256>(d_st, d_sink, n_cu);
  run<S_T, 8, 8, 8, 8, 256>(d_st, d_sink, n_cu);
  // the kernel's own phase structure (per CMUX iteration: 941 VALU of which 656 f64, 48 stores, 74 loads, 4 barriers)
  if (getenv("GATE_KERNEL_LIKE") == nullptr || atoi(getenv("GATE_KERNEL_LIKE")) != 0) {
    run_k<0>(d_st, d_sink, n_cu, 1);
    run_k<0>(d_st, d_sink, n_cu, 2);
    run_k<7>(d_st, d_sink, n_cu, 1);
    run_k<7>(d_st, d_sink, n_cu, 2);
    run_k<8>(d_st, d_sink, n_cu, 2);
    run_k<9>(d_st, d_sink, n_cu, 1);
    if (getenv("GATE_ONLY_PERMLANE")) return 0;
    run_k<1>(d_st, d_sink, n_cu, 1);
    run_k<1>(d_st, d_sink, n_cu, 2);
    run_k<2>(d_st, d_sink, n_cu, 2);
    run_k<3>(d_st, d_sink, n_cu, 2);
    run_k<4>(d_st, d_sink, n_cu, 1);
    run_k<4>(d_st, d_sink, n_cu, 2);
    run_k<6>(d_st, d_sink, n_cu, 2);
    run_k<5>(d_st, d_sink, n_cu, 2);
    // what more resident workgroups would be worth IF registers and LDS allowed them (they do not: 240 registers, 80 KB per workgroup)
    run_k<0>(d_st, d_sink, n_cu, 3);
    run_k<5>(d_st, d_sink, n_cu, 3);
    run_k<0>(d_st, d_sink, n_cu, 4);
  }
  return 0;
}
