// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (ceiling for the SYRK kernel)
// and a plain fp64 FMA rate.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  if (a0 < 0.0) {  // "random" full-mantissa operands: realistic switching power
    unsigned long long h1 = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 977 * blockIdx.x), h2 = h1 * 0xBF58476D1CE4E5B9ull;
    a = 1.0 + (double)(h1 >> 11) * (1.0 / 9007199254740992.0);
    b = -1.0 + (double)(h2 >> 11) * (1.0 / 9007199254740992.0) * 1e-3;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_loop(double* out, int iters, double a0, double b0) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  double* d;
  hipMalloc(&d, 4096 * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 4; ++mode) {
    int wg_per_cu = 1 + (mode & 1);
    int blocks = 256 * wg_per_cu;
    double a0 = mode < 2 ? 1.0 : -1.0;
    printf("%s operands\n", mode < 2 ? "trivial" : "random-mantissa");
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, d, iters, a0, 0.5);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)blocks * 4 * iters * 16 * 2048.0;
      printf("mfma_f64 16x16x4: %d WG/CU (%d waves/SIMD) %.3f ms  %.2f TFLOP/s\n", wg_per_cu, wg_per_cu, ms, flop / ms / 1e9);
    }
  }
  for (int rep = 0; rep < 3; ++rep) {
    int blocks = 256 * 8;
    hipEventRecord(e0);
    hipLaunchKernelGGL(fma_loop, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999, 0.001);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 256 * iters * 16 * 2.0;
    printf("v_fma_f64: %.3f ms  %.2f TFLOP/s\n", ms, flop / ms / 1e9);
  }
  return 0;
}
