// sclk(t) after an idle period: one wave samples (wall clock, shader clock) pairs while it spins, so the ratio of
// the increments is the shader clock in units of the 100 MHz wall clock.  hipcc --offload-arch=gfx950 -O2 -o clock_ramp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#include <vector>
__global__ void probe(unsigned long long* out, int nsamp, int spin) {
  for (int s = 0; s < nsamp; ++s) {
    out[2 * s] = wall_clock64();
    out[2 * s + 1] = clock64();
    double a = 1.0 + s;
    for (int i = 0; i < spin; ++i) a = __builtin_fma(a, 1.0000001, 1e-9);
    if (a == 0.123) out[0] = 0;
  }
}
// a chip-wide fp64 VALU load next to the sampler (does the ramp depend on load?)
__global__ void burn(double* out, int iters) {
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = __builtin_fma(a, 1.0000001, 1e-9);
  if (a == 0.123) out[0] = a;
}
int main(int argc, char** argv) {
  const int nsamp = 400, spin = 2000;
  unsigned long long* d;
  double* dd;
  hipMalloc(&d, nsamp * 16);
  hipMalloc(&dd, 8);
  std::vector<unsigned long long> h(2 * nsamp);
  int wc_khz = 0;
  hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", wc_khz);
  for (int with_load = 0; with_load < 2; ++with_load)
    for (double idle : {0.0, 0.005, 0.1, 1.0}) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, nsamp, spin);
      hipDeviceSynchronize();
      usleep((useconds_t)(idle * 1e6));
      hipStream_t s2;
      hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
      if (with_load) hipLaunchKernelGGL(burn, dim3(256 * 8), dim3(256), 0, s2, dd, 3000000);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, nsamp, spin);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d, nsamp * 16, hipMemcpyDeviceToHost);
      printf("idle %.0f ms, %s: t[ms] -> sclk[MHz]:", idle * 1e3, with_load ? "fp64 load on all CUs" : "single wave");
      double t0 = (double)h[0];
      for (int s = 20; s < nsamp; s += 20) {
        const double dt = (double)(h[2 * s] - h[2 * (s - 20)]) / (wc_khz * 1e3);
        const double dc = (double)(h[2 * s + 1] - h[2 * (s - 20) + 1]);
        printf(" %.2f:%.0f", ((double)h[2 * s] - t0) / wc_khz, dc / dt / 1e6);
      }
      printf("\n");
      hipStreamDestroy(s2);
    }
  return 0;
}
