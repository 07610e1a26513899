// Ablation harness for the covariance fill kernel (diagnostic only).
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include "../battgp_amd/csrc/bgp_fill.hip"
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
int bgp_fail(bgp_handle*, int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); return code; }
__global__ void heat(double* o, int iters) { double a = threadIdx.x; for (int i = 0; i < iters; ++i) a = __builtin_fma(a, 1.0000001, 1e-9); o[blockIdx.x * 256 + threadIdx.x] = a; }

template <int KID, int ABL>
static float run(FillParams p, const double* x, int64_t n, double* out, int64_t ld) {
  const int nti = (int)((n + FT_ROWS - 1) / FT_ROWS), ntj = (int)((n + FT_COLS - 1) / FT_COLS);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<KID, 4, ABL>), dim3((unsigned)lower_blocks(nti)), dim3(256), 0, 0, p, x, n, x, n, out, ld, 1, 1, n, n, nti, ntj, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  FillParams p{}; p.kid = 0; p.D = 4; p.noise = 2.33e-6; p.s0 = 4.23e-13; p.s1 = 0.0099;
  p.scale[0] = 1; p.scale[1] = 0.7071 / 12.11; p.scale[2] = 0.7071 / 33.75; p.scale[3] = 0.7071 / 45.14;
  for (int64_t pad : {64ll, 128ll, 384ll, 576ll})
  for (int64_t n : {131072ll}) {
    int64_t ld = n + pad; printf("ld = n + %lld\n", (long long)pad);
    double *x, *out;
    hipMalloc(&x, n * 4 * 8);
    if (hipMalloc(&out, ld * n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    std::vector<double> hx(n * 4);
    for (int64_t i = 0; i < n; ++i) { hx[4 * i] = 1200.0 * i / n; hx[4 * i + 1] = -80 + 75.0 * ((i * 7919) % 1000) / 1000; hx[4 * i + 2] = 40 + 55.0 * ((i * 104729) % 1000) / 1000; hx[4 * i + 3] = 10 + 35.0 * ((i * 1299709) % 1000) / 1000; }
    hipMemcpy(x, hx.data(), n * 4 * 8, hipMemcpyHostToDevice);
    const double gb = 4.0 * n * (n + 1) / 1e9;
    float t;
    t = run<0, 0>(p, x, n, out, ld); printf("N=%6lld battgp  production     %8.3f ms  %6.0f GB/s\n", (long long)n, t, gb / t * 1e3);
    t = run<0, 1>(p, x, n, out, ld); printf("N=%6lld battgp  stores only    %8.3f ms  %6.0f GB/s\n", (long long)n, t, gb / t * 1e3);
    t = run<2, 0>(p, x, n, out, ld); printf("N=%6lld matern  production     %8.3f ms  %6.0f GB/s\n", (long long)n, t, gb / t * 1e3);
    // memset reference for the same bytes
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipMemsetAsync(out, 0, (size_t)(gb * 1e9), 0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&t, e0, e1); printf("N=%6lld hipMemset of the same bytes %8.3f ms  %6.0f GB/s\n", (long long)n, t, gb / t * 1e3);
    hipFree(out); hipFree(x);
  }
  return 0;
}
