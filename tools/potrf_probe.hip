// Where do the 24 us of the 64 x 64 tile kernel go?  (diagnostic only, not part of the product)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/potrf_probe.hip -o tools/potrf_probe
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include "../battgp_amd/csrc/bgp_linalg.hip"
#include <stdarg.h>
#include <stdio.h>
int bgp_fail(bgp_handle*, int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); return code; }

// PH bit 0: tile load/store, bit 1: Cholesky, bit 2: inverse
template <int PH>
__global__ __launch_bounds__(256, 2) void probe_kernel(double* __restrict__ Ajj, int64_t lda, double* __restrict__ inv,
                                                       int* __restrict__ info) {
  __shared__ double s[64][64];
  __shared__ double colbuf[2][64];
  __shared__ int sfail;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (*info != 0) return;
  if (tid == 0) sfail = 0;
  if (PH & 1)
    for (int idx = tid; idx < 4096; idx += 256) s[idx >> 6][idx & 63] = Ajj[(idx & 63) + (int64_t)(idx >> 6) * lda];
  __syncthreads();
  if (PH & 2) {
    switch (wave) {
      case 0: chol_cols<0>(s, colbuf, &sfail, lane); break;
      case 1: chol_cols<1>(s, colbuf, &sfail, lane); break;
      case 2: chol_cols<2>(s, colbuf, &sfail, lane); break;
      default: chol_cols<3>(s, colbuf, &sfail, lane); break;
    }
  }
  __syncthreads();
  if (PH & 1)
    for (int idx = tid; idx < 4096; idx += 256) {
      const int r = idx & 63, c = idx >> 6;
      if (r >= c) Ajj[r + (int64_t)c * lda] = s[c][r];
    }
  if (PH & 4) {
    switch (wave) {
      case 0: inv_cols<0>(s, inv, lane); break;
      case 1: inv_cols<1>(s, inv, lane); break;
      case 2: inv_cols<2>(s, inv, lane); break;
      default: inv_cols<3>(s, inv, lane); break;
    }
  }
}

template <int PH>
static void run(const char* name, double* A, double* A0, double* inv, int* info) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 400;
  hipMemcpy(A, A0, 64 * 64 * 8, hipMemcpyDeviceToDevice);
  hipLaunchKernelGGL((probe_kernel<PH>), dim3(1), dim3(256), 0, 0, A, 64, inv, info);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe_kernel<PH>), dim3(1), dim3(256), 0, 0, A0 + 4096, 64, inv, info);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %7.2f us per launch\n", name, ms * 1e3 / reps);
}

int main() {
  double *A, *A0, *inv;
  int* info;
  hipMalloc(&A, 4096 * 8);
  hipMalloc(&A0, 2 * 4096 * 8);
  hipMalloc(&inv, 4096 * 8);
  hipMalloc(&info, 4);
  hipMemset(info, 0, 4);
  double h[4096];
  for (int c = 0; c < 64; ++c)
    for (int r = 0; r < 64; ++r) h[r + c * 64] = (r == c) ? 70.0 : 1.0 / (1.0 + (r > c ? r - c : c - r));
  hipMemcpy(A0, h, sizeof(h), hipMemcpyHostToDevice);
  // the timed launches work on a copy that is refactored again and again: make it an identity-like SPD tile
  // whose factor is again SPD-factorable (diagonal 70 -> sqrt ... stays positive for a few hundred rounds? no:
  // use a tile that the kernel maps to itself: the identity)
  for (int i = 0; i < 4096; ++i) h[i] = (i % 65 == 0) ? 1.0 : 0.0;
  hipMemcpy(A0 + 4096, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>("empty kernel (launch + barriers)", A, A0, inv, info);
  run<1>("tile load + store", A, A0, inv, info);
  run<3>("load/store + Cholesky", A, A0, inv, info);
  run<5>("load/store + inverse", A, A0, inv, info);
  run<7>("all (= potrf_tile_kernel)", A, A0, inv, info);
  int hi = 0;
  hipMemcpy(&hi, info, 4, hipMemcpyDeviceToHost);
  printf("info = %d\n", hi);
  return 0;
}
