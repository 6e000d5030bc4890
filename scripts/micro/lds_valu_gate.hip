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
  hipMalloc(&d_st, (size_t)(n_cu + 1) * 8 * 8);
  hipMalloc(&d_sink, (size_t)(n_cu + 1) * 512 * 8);
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
  return 0;
}
