// micro-benchmark: issue rate of f64 VALU ops per SIMD on gfx950 (cycles per wave64 instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ void k(double *out, int iters, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (KIND == 0) { x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
                       x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b); }
      if (KIND == 1) { x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a; }
      if (KIND == 2) { x0 *= a; x1 *= a; x2 *= a; x3 *= a; x4 *= a; x5 *= a; x6 *= a; x7 *= a; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int KIND>
__global__ void kf32(float *out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
  double *d; hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char *names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_fma_f32"};
  for (int kind = 0; kind < 4; kind++)
    for (int waves_per_simd : {1, 2, 4}) {
      dim3 grid(256), block(64 * 4 * waves_per_simd);   // one block per CU
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters, 1.0000001, 1e-9);
        if (kind == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters, 1.0000001, 1e-9);
        if (kind == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, iters, 1.0000001, 1e-9);
        if (kind == 3) hipLaunchKernelGGL(kf32<0>, grid, block, 0, 0, (float *)d, iters, 1.0000001f, 1e-9f);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_wave = (double)iters * 64;
      const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
      printf("%-10s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instruction per SIMD (= %.1f cycles @2.4GHz)  -> %.1f TFLOP/s-equivalent(FMA=2)\n",
             names[kind], waves_per_simd, ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4,
             256.0 * 4 * 64 * 2 / ns_per_instr_per_simd / 1e3);
    }
  return 0;
}
