// micro-benchmark (round 3): do f64 VALU work and LDS traffic overlap on a gfx950 CU?
// One 512-thread workgroup per CU (two waves per SIMD).  Per loop iteration a wave issues 64 v_fma_f64 on
// registers and / or NW ds_write_b128 + NR ds_read_b128 (conflict-free, lane-consecutive 16-byte slots), with no
// data dependency between the two streams.  Modes:
//   0 = FMA only   1 = LDS only   2 = both, interleaved in every wave
//   3 = specialised: waves 0-3 (one per SIMD) FMA only, waves 4-7 LDS only
//   4 = both, but the LDS operations in one burst at the end of the iteration behind a workgroup barrier
// Output: cycles per iteration (s_memtime, median wave) -- if mode 2 ~ mode 0 + mode 1 the two pipes serialise.
//   hipcc --offload-arch=gfx950 -O3 -o lds_valu_overlap lds_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define FMA8 asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b))
#define WR(off) asm volatile("ds_write_b128 %0, %1 offset:" #off :: "v"(addr), "v"(wv) : "memory")
#define WR64(off) asm volatile("ds_write_b64 %0, %1 offset:" #off "\n ds_write_b64 %0, %2 offset:" #off "+8" :: "v"(addr), "v"(wlo), "v"(whi) : "memory")
#define RD(dst, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")

#define RDALL if (NR >= 1) RD(r0, 0); if (NR >= 2) RD(r1, 8192); if (NR >= 3) RD(r2, 16384); if (NR >= 4) RD(r3, 24576); \
              if (NR >= 5) RD(r0, 32768); if (NR >= 6) RD(r1, 40960); if (NR >= 7) RD(r2, 49152); if (NR >= 8) RD(r3, 57344); \
              if (NR >= 9) RD(r0, 4096); if (NR >= 10) RD(r1, 12288); if (NR >= 11) RD(r2, 20480); if (NR >= 12) RD(r3, 28672); \
              if (NR >= 13) RD(r0, 36864); if (NR >= 14) RD(r1, 45056); \
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WRALL if (NW >= 1) WR(0); if (NW >= 2) WR(8192); if (NW >= 3) WR(16384); if (NW >= 4) WR(24576); \
              if (NW >= 5) WR(32768); if (NW >= 6) WR(40960); if (NW >= 7) WR(49152); if (NW >= 8) WR(57344); \
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

#define RDRAW if (NR >= 1) RD(r0, 0); if (NR >= 2) RD(r1, 8192); if (NR >= 3) RD(r2, 16384); if (NR >= 4) RD(r3, 24576); \
              if (NR >= 5) RD(r0, 32768); if (NR >= 6) RD(r1, 40960); if (NR >= 7) RD(r2, 49152); if (NR >= 8) RD(r3, 57344); \
              if (NR >= 9) RD(r0, 4096); if (NR >= 10) RD(r1, 12288); if (NR >= 11) RD(r2, 20480); if (NR >= 12) RD(r3, 28672); \
              if (NR >= 13) RD(r0, 36864); if (NR >= 14) RD(r1, 45056)
#define WRRAW if (NW >= 1) WR(0); if (NW >= 2) WR(8192); if (NW >= 3) WR(16384); if (NW >= 4) WR(24576); \
              if (NW >= 5) WR(32768); if (NW >= 6) WR(40960); if (NW >= 7) WR(49152); if (NW >= 8) WR(57344)

#define FMAH _Pragma("unroll") for (int f = 0; f < NF / 2; ++f) { FMA8; }
template <int MODE, int NW, int NR, int NF = 8>
__global__ __launch_bounds__(512) void k(unsigned long long *stamps, double *sink, int iters, double a, double b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  double x0 = threadIdx.x + 1.0, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const unsigned addr = threadIdx.x * 16u;  // 512 threads x 16 B = 8 KB per "row"; offsets step 8 KB
  u32x4 wv = {threadIdx.x, 1u, 2u, 3u};
  u32x4 r0 = {}, r1 = {}, r2 = {}, r3 = {};
  const unsigned long long wlo = threadIdx.x, whi = threadIdx.x * 3ull;
  const bool do_fma = MODE == 0 || MODE == 2 || MODE == 4 || MODE == 8 || (MODE == 3 && threadIdx.x < 256);
  const bool do_lds = MODE == 1 || MODE == 2 || MODE == 4 || (MODE == 3 && threadIdx.x >= 256);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int i = 0; i < iters; i++) {
    if (MODE == 4) {
      if (do_fma) { _Pragma("unroll") for (int f = 0; f < NF; ++f) { FMA8; } }
      if (NW >= 1) WR(0);
      if (NW >= 2) WR(8192);
      if (NW >= 3) WR(16384);
      if (NW >= 4) WR(24576);
      if (NW >= 5) WR(32768);
      if (NW >= 6) WR(40960);
      if (NW >= 7) WR(49152);
      if (NW >= 8) WR(57344);
      __syncthreads();
      if (NR >= 1) RD(r0, 0);
      if (NR >= 2) RD(r1, 8192);
      if (NR >= 3) RD(r2, 16384);
      if (NR >= 4) RD(r3, 24576);
      if (NR >= 5) RD(r0, 32768);
      if (NR >= 6) RD(r1, 40960);
      if (NR >= 7) RD(r2, 49152);
      if (NR >= 8) RD(r3, 57344);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
    } else if (MODE == 7) {  // LDS only, every 16-byte store as two ds_write_b64 (same bytes as mode 1 with NW stores, no loads)
      WR64(0); WR64(8192); WR64(16384); WR64(24576); WR64(32768); WR64(40960); WR64(49152); WR64(57344);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE == 8) {
      // ONE wave per SIMD carrying TWO independent streams A and B half a step apart (software pipeline): every store
      // burst gets half an arithmetic block to drain before the barrier that publishes it, every load burst half a
      // block (plus the other stream's store issue) to land before its first use.  Work per iteration and wave: 2 x NF
      // FMA8, 2 x NW stores, 2 x NR loads -- what TWO waves do in mode 4.
      // LDS queue of a wave (in order): ... La | Sb | Lb | Sa | La | ...
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // La landed (Sb may be in flight)
      FMAH;                                                // Va.1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // Sb complete
      __builtin_amdgcn_s_barrier();
      RDRAW;                                               // Lb
      FMAH;                                                // Va.2
      WRRAW;                                               // Sa
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // Lb landed
      FMAH;                                                // Vb.1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // Sa complete
      __builtin_amdgcn_s_barrier();
      RDRAW;                                               // La
      FMAH;                                                // Vb.2
      WRRAW;                                               // Sb
    } else if (MODE == 3) {
      if (threadIdx.x < 256) { FMA8; FMA8; FMA8; FMA8; FMA8; FMA8; FMA8; FMA8; }
      else { WRALL; RDALL; }
    } else if (MODE == 5 || MODE == 6) {
      // two teams (waves 0-3 / 4-7) half a phase apart, every barrier shared: in each interval one team reads and
      // runs the first half of its arithmetic while the other runs its second half and writes.
      // MODE 6: same stream, both teams in phase (R, V, W together): the lockstep reference with the same barriers.
      if (MODE == 6 || threadIdx.x < 256) {
        RDALL; FMAH;
        __builtin_amdgcn_s_barrier();
        FMAH; WRALL;
        __builtin_amdgcn_s_barrier();
      } else {
        FMAH; WRALL;
        __builtin_amdgcn_s_barrier();
        RDALL; FMAH;
        __builtin_amdgcn_s_barrier();
      }
    } else {
      if (do_fma) FMA8;
      if (do_lds && NW >= 1) WR(0);
      if (do_lds && NR >= 1) RD(r0, 0);
      if (do_fma) FMA8;
      if (do_lds && NW >= 2) WR(8192);
      if (do_lds && NR >= 2) RD(r1, 8192);
      if (do_fma) FMA8;
      if (do_lds && NW >= 3) WR(16384);
      if (do_lds && NR >= 3) RD(r2, 16384);
      if (do_fma) FMA8;
      if (do_lds && NW >= 4) WR(24576);
      if (do_lds && NR >= 4) RD(r3, 24576);
      if (do_fma) FMA8;
      if (do_lds && NW >= 5) WR(32768);
      if (do_lds && NR >= 5) RD(r0, 32768);
      if (do_fma) FMA8;
      if (do_lds && NW >= 6) WR(40960);
      if (do_lds && NR >= 6) RD(r1, 40960);
      if (do_fma) FMA8;
      if (do_lds && NW >= 7) WR(49152);
      if (do_lds && NR >= 7) RD(r2, 49152);
      if (do_fma) FMA8;
      if (do_lds && NW >= 8) WR(57344);
      if (do_lds && NR >= 8) RD(r3, 57344);
      if (do_lds) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (double)(r0.x + r1.y + r2.z + r3.w);
}

template <int MODE, int NW, int NR, int NF = 8>
static void run(unsigned long long *d_st, double *d_sink, int n_cu) {
  const int iters = 20000;
  auto kern = k<MODE, NW, NR, NF>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(MODE == 8 ? 256 : 512), 65536, 0, d_st, d_sink, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> st((size_t)n_cu * 8);
  hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> lo, hi;  // waves 0-3 / 4-7
  for (int b = 0; b < n_cu; b++)
    for (int w = 0; w < 8; w++) (w < 4 ? lo : hi).push_back((double)st[(size_t)b * 8 + w] / iters);
  std::sort(lo.begin(), lo.end());
  std::sort(hi.begin(), hi.end());
  printf("{\"mode\": %d, \"writes_b128\": %d, \"reads_b128\": %d, \"fma_f64\": %d, \"cycles_per_iter_waves0_3\": %.1f, "
         "\"cycles_per_iter_waves4_7\": %.1f, \"wall_ms\": %.3f}\n",
         MODE, NW, NR, MODE == 8 ? NF * 16 : (MODE == 0 || MODE == 4 || MODE == 5 || MODE == 6 ? NF * 8 : 64), lo[lo.size() / 2], hi[hi.size() / 2], ms);
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
  run<0, 0, 0>(d_st, d_sink, n_cu);
  run<1, 4, 8>(d_st, d_sink, n_cu);
  run<2, 4, 8>(d_st, d_sink, n_cu);
  run<3, 4, 8>(d_st, d_sink, n_cu);
  run<4, 4, 8>(d_st, d_sink, n_cu);
  run<1, 8, 8>(d_st, d_sink, n_cu);
  run<2, 8, 8>(d_st, d_sink, n_cu);
  run<3, 8, 8>(d_st, d_sink, n_cu);
  run<4, 8, 8>(d_st, d_sink, n_cu);
  run<5, 8, 8>(d_st, d_sink, n_cu);
  run<6, 8, 8>(d_st, d_sink, n_cu);
  run<5, 8, 14>(d_st, d_sink, n_cu);
  run<6, 8, 14>(d_st, d_sink, n_cu);
  // the blind rotate's ratio: ~160 VALU slots, 8 stores and ~14 loads per barrier-separated phase
  run<4, 0, 0, 20>(d_st, d_sink, n_cu);
  run<4, 8, 14, 20>(d_st, d_sink, n_cu);
  run<5, 8, 14, 20>(d_st, d_sink, n_cu);
  run<6, 8, 14, 20>(d_st, d_sink, n_cu);
  run<4, 8, 8, 20>(d_st, d_sink, n_cu);
  run<5, 8, 8, 20>(d_st, d_sink, n_cu);
  run<6, 8, 8, 20>(d_st, d_sink, n_cu);
  // software pipeline of two streams in one wave per SIMD (mode 8) against the burst structure with two waves per SIMD (mode 4)
  run<8, 8, 14, 20>(d_st, d_sink, n_cu);
  run<8, 8, 8, 20>(d_st, d_sink, n_cu);
  run<8, 8, 8, 8>(d_st, d_sink, n_cu);
  run<7, 8, 0>(d_st, d_sink, n_cu);
  run<1, 8, 0>(d_st, d_sink, n_cu);
  run<2, 8, 0>(d_st, d_sink, n_cu);
  run<1, 0, 8>(d_st, d_sink, n_cu);
  run<2, 0, 8>(d_st, d_sink, n_cu);
  return 0;
}
