// Does a high-priority stream's small kernel get CU slots while a big grid of long workgroups runs?
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));
// long-running MFMA workgroup using ~all VGPRs (like the trailing-update GEMM): 256 thr, 2 WG/CU
__global__ __launch_bounds__(256, 2) void hog(double* out, int iters, int stagger) {
  if (stagger && blockIdx.x < 512) { int ns = (blockIdx.x % 16) * stagger; for (int q = 0; q < ns; ++q) __builtin_amdgcn_s_sleep(64); }
  v4d acc[30];
  for (int i = 0; i < 30; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 30; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  double s = 0; for (int i = 0; i < 30; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256, 2) void small(double* out, long long* t) {
  long long t0 = wall_clock64();
  double x = threadIdx.x;
  for (int i = 0; i < 2000; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { t[0] = t0; t[1] = wall_clock64(); }
}
int main() {
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  printf("stream priority range: least=%d greatest=%d\n", lo, hi);
  double* d; hipMalloc(&d, 1 << 28); long long* t; hipMalloc(&t, 64);
  for (int stg = 0; stg <= 8; stg += 8)
  for (int mode = 0; mode < 2; ++mode) {
    printf("stagger=%d ", stg);
    hipStream_t sm, sa;
    if (mode == 0) { hipStreamCreateWithFlags(&sm, hipStreamNonBlocking); hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); }
    if (mode == 1) { hipStreamCreateWithFlags(&sm, hipStreamNonBlocking); hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi); }
    if (mode == 2) { hipStreamCreateWithPriority(&sm, hipStreamNonBlocking, lo); hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // hog: 20000 WGs x ~100us each
    hipLaunchKernelGGL(hog, dim3(20000), dim3(256), 0, sm, d, 130, stg);
    std::vector<float> lat;
    for (int i = 0; i < 40; ++i) {
      hipEventRecord(e0, sa);
      hipLaunchKernelGGL(hog, dim3(1), dim3(256), 0, sa, d + (1 << 24), 10, 0);
      hipEventRecord(e1, sa);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); lat.push_back(ms * 1e3f);
    }
    hipStreamSynchronize(sm);
    std::sort(lat.begin(), lat.end());
    printf("mode %d (%s): small-kernel latency under load: min %.1f us median %.1f us max %.1f us\n", mode,
           mode == 0 ? "both default" : mode == 1 ? "aux high" : "main low + aux high", lat[0], lat[20], lat[39]);
    // unloaded reference
    lat.clear();
    for (int i = 0; i < 20; ++i) { hipEventRecord(e0, sa); hipLaunchKernelGGL(hog, dim3(1), dim3(256), 0, sa, d + (1 << 24), 10, 0); hipEventRecord(e1, sa); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); lat.push_back(ms * 1e3f); }
    std::sort(lat.begin(), lat.end());
    printf("        idle GPU: median %.1f us\n", lat[10]);
  }
  return 0;
}
