// micro-benchmark: issue cost of the VALU instructions the blind-rotate kernels are made of, on gfx950.
// For every instruction kind and 1 / 2 / 4 waves per SIMD: cycles per wave64 instruction per SIMD, measured
// IN KERNEL (s_memtime = shader clock) and the clock the chip actually holds under that load
// (delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6).
// Output: one JSON object per line (bench.py reads them: roofline_compute.issue_cycles).
//   hipcc --offload-arch=gfx950 -O3 -o f64_rate f64_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(S) S S S S S S S S
#define REP64(S) REP8(REP8(S))

enum { K_FMA64, K_ADD64, K_MUL64, K_FMA32, K_DPP, K_CVT, K_LSHLADD64, K_SUB64, K_CNDMASK, K_PERMSWAP, K_AND, K_MIX, K_CND_INDEP, K_CND_SGPR, K_XOR, K_ADDC, K_MAD64, K_MULLO, K_MAD24, K_COUNT };
static const char *kNames[K_COUNT] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_fma_f32", "v_mov_b32_dpp", "v_cvt_f64_i32",
                                      "v_lshl_add_u64", "v_sub_co+v_subb (pair)", "v_cndmask_b32", "v_permlane32_swap",
                                      "v_and_b32", "mix 5 fma_f64 : 3 int",
                                      "v_cndmask_b32 vcc (independent)", "v_cndmask_b32 e64 sgpr cond", "v_xor_b32", "v_add_co+v_addc (pair)",
                                      "v_mad_u64_u32", "v_mul_lo_u32", "v_mad_u32_u24"};

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long *stamps, double *sink, int iters, double a, double b) {
  double x0 = threadIdx.x + 1.0, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
  unsigned long long u0 = threadIdx.x, u1 = u0 * 3;
  const int j0 = threadIdx.x * 7 + 1, j1 = threadIdx.x * 5 + 2;
  const unsigned long long smask = 0x5555555555555555ull ^ (unsigned long long)iters;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int i = 0; i < iters; i++) {
    if (KIND == K_FMA64) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));) }
    if (KIND == K_ADD64) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));) }
    if (KIND == K_MUL64) { REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));) }
    if (KIND == K_FMA32) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(1.0000001f), "v"(1e-9f));) }
    if (KIND == K_DPP) { REP8(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
    if (KIND == K_CVT) { REP8(asm volatile("v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %9\n v_cvt_f64_i32 %2, %10\n v_cvt_f64_i32 %3, %11\n v_cvt_f64_i32 %4, %8\n v_cvt_f64_i32 %5, %9\n v_cvt_f64_i32 %6, %10\n v_cvt_f64_i32 %7, %11" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
    if (KIND == K_LSHLADD64) { REP8(asm volatile("v_lshl_add_u64 %0, %1, 3, %0\n v_lshl_add_u64 %1, %0, 3, %1\n v_lshl_add_u64 %0, %1, 3, %0\n v_lshl_add_u64 %1, %0, 3, %1\n v_lshl_add_u64 %0, %1, 3, %0\n v_lshl_add_u64 %1, %0, 3, %1\n v_lshl_add_u64 %0, %1, 3, %0\n v_lshl_add_u64 %1, %0, 3, %1" : "+v"(u0), "+v"(u1));) }
    if (KIND == K_SUB64) { REP8(asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n v_subb_co_u32 %1, vcc, %1, %3, vcc\n v_sub_co_u32 %2, vcc, %2, %0\n v_subb_co_u32 %3, vcc, %3, %1, vcc\n v_sub_co_u32 %0, vcc, %0, %2\n v_subb_co_u32 %1, vcc, %1, %3, vcc\n v_sub_co_u32 %2, vcc, %2, %0\n v_subb_co_u32 %3, vcc, %3, %1, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "vcc");) }
    if (KIND == K_CNDMASK) { REP8(asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc\n v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "vcc");) }
    if (KIND == K_PERMSWAP) { REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
    if (KIND == K_AND) { REP8(asm volatile("v_and_b32 %0, %1, %2\n v_and_b32 %1, %2, %3\n v_and_b32 %2, %3, %0\n v_and_b32 %3, %0, %1\n v_and_b32 %0, %1, %2\n v_and_b32 %1, %2, %3\n v_and_b32 %2, %3, %0\n v_and_b32 %3, %0, %1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
    if (KIND == K_CND_INDEP) { REP8(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %5, %4, vcc\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %5, %4, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(j0), "v"(j1) : "vcc");) }
    if (KIND == K_CND_SGPR) { REP8(asm volatile("v_cndmask_b32 %0, %4, %5, %6\n v_cndmask_b32 %1, %5, %4, %6\n v_cndmask_b32 %2, %4, %5, %6\n v_cndmask_b32 %3, %5, %4, %6\n v_cndmask_b32 %0, %4, %5, %6\n v_cndmask_b32 %1, %5, %4, %6\n v_cndmask_b32 %2, %4, %5, %6\n v_cndmask_b32 %3, %5, %4, %6" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(j0), "v"(j1), "s"(smask));) }
    if (KIND == K_XOR) { REP8(asm volatile("v_xor_b32 %0, %1, %2\n v_xor_b32 %1, %2, %3\n v_xor_b32 %2, %3, %0\n v_xor_b32 %3, %0, %1\n v_xor_b32 %0, %1, %2\n v_xor_b32 %1, %2, %3\n v_xor_b32 %2, %3, %0\n v_xor_b32 %3, %0, %1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
    if (KIND == K_ADDC) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc\n v_add_co_u32 %2, vcc, %2, %0\n v_addc_co_u32 %3, vcc, %3, %1, vcc\n v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc\n v_add_co_u32 %2, vcc, %2, %0\n v_addc_co_u32 %3, vcc, %3, %1, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "vcc");) }
    if (KIND == K_MAD64) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %4, %1\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %2, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %4, %1\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %2, %1" : "+v"(u0), "+v"(u1) : "v"(i0), "v"(i1), "v"(i2), "v"(i3) : "vcc");) }
    if (KIND == K_MULLO) { REP8(asm volatile("v_mul_lo_u32 %0, %1, %4\n v_mul_lo_u32 %1, %2, %4\n v_mul_lo_u32 %2, %3, %4\n v_mul_lo_u32 %3, %0, %4\n v_mul_lo_u32 %0, %1, %4\n v_mul_lo_u32 %1, %2, %4\n v_mul_lo_u32 %2, %3, %4\n v_mul_lo_u32 %3, %0, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(j0));) }
    if (KIND == K_MAD24) { REP8(asm volatile("v_mad_u32_u24 %0, %1, %4, %0\n v_mad_u32_u24 %1, %2, %4, %1\n v_mad_u32_u24 %2, %3, %4, %2\n v_mad_u32_u24 %3, %0, %4, %3\n v_mad_u32_u24 %0, %1, %4, %0\n v_mad_u32_u24 %1, %2, %4, %1\n v_mad_u32_u24 %2, %3, %4, %2\n v_mad_u32_u24 %3, %0, %4, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(j0));) }
    if (KIND == K_MIX) { REP8(asm volatile("v_fma_f64 %0, %0, %12, %13\n v_and_b32 %8, %9, %10\n v_fma_f64 %1, %1, %12, %13\n v_fma_f64 %2, %2, %12, %13\n v_and_b32 %9, %10, %11\n v_fma_f64 %3, %3, %12, %13\n v_fma_f64 %4, %4, %12, %13\n v_and_b32 %10, %11, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(i0), "+v"(i1), "+v"(i2) : "v"(i3), "v"(a), "v"(b));) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1 + i2 + i3 + (double)(u0 + u1);
}

template <int KIND>
static void run(unsigned long long *d_st, double *d_sink, int wps, int n_cu) {
  const int iters = KIND == K_FMA32 || KIND == K_AND || KIND == K_XOR ? 40000 : 20000;
  dim3 grid(n_cu), block(64 * 4 * wps);  // one block per CU
  const int n_waves = n_cu * 4 * wps;
  std::vector<unsigned long long> st(2 * (size_t)n_waves);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d_st, d_sink, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc(n_waves), clk(n_waves);
  for (int w = 0; w < n_waves; w++) {
    cyc[w] = (double)st[2 * w];
    clk[w] = (double)st[2 * w] / (double)st[2 * w + 1] * 100e6;
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  const int per_iter = KIND == K_DPP ? 64 : 64;  // instructions per loop iteration (the s_nops of the DPP body not counted)
  const double instr_per_wave = (double)iters * per_iter;
  const double cyc_per_instr_simd = cyc[n_waves / 2] / (instr_per_wave * wps);
  printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_instr_per_simd\": %.3f, \"in_kernel_clock_ghz\": %.3f, "
         "\"wall_ms\": %.3f, \"cycles_from_wall_at_measured_clock\": %.3f}\n",
         kNames[KIND], wps, cyc_per_instr_simd, clk[n_waves / 2] / 1e9, ms,
         ms * 1e-3 * clk[n_waves / 2] / (instr_per_wave * wps));
  fflush(stdout);
}

int main() {
  int n_cu = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) == hipSuccess) n_cu = prop.multiProcessorCount;
  unsigned long long *d_st;
  double *d_sink;
  hipMalloc(&d_st, (size_t)(n_cu + 1) * 16 * 2 * 8);
  hipMalloc(&d_sink, (size_t)(n_cu + 1) * 1024 * 8);
  for (int wps : {1, 2, 4}) {
    run<K_FMA64>(d_st, d_sink, wps, n_cu);
    run<K_ADD64>(d_st, d_sink, wps, n_cu);
    run<K_MUL64>(d_st, d_sink, wps, n_cu);
    run<K_FMA32>(d_st, d_sink, wps, n_cu);
    run<K_DPP>(d_st, d_sink, wps, n_cu);
    run<K_CVT>(d_st, d_sink, wps, n_cu);
    run<K_LSHLADD64>(d_st, d_sink, wps, n_cu);
    run<K_SUB64>(d_st, d_sink, wps, n_cu);
    run<K_CNDMASK>(d_st, d_sink, wps, n_cu);
    run<K_PERMSWAP>(d_st, d_sink, wps, n_cu);
    run<K_AND>(d_st, d_sink, wps, n_cu);
    run<K_MIX>(d_st, d_sink, wps, n_cu);
    run<K_CND_INDEP>(d_st, d_sink, wps, n_cu);
    run<K_CND_SGPR>(d_st, d_sink, wps, n_cu);
    run<K_XOR>(d_st, d_sink, wps, n_cu);
    run<K_ADDC>(d_st, d_sink, wps, n_cu);
    run<K_MAD64>(d_st, d_sink, wps, n_cu);
    run<K_MULLO>(d_st, d_sink, wps, n_cu);
    run<K_MAD24>(d_st, d_sink, wps, n_cu);
  }
  return 0;
}
