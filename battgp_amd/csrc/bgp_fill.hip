// Covariance fill kernels (gfx950).
//
// One workgroup = one 128 x 128 tile of the (column-major) covariance matrix.  The 128
// column points are scaled once and staged in LDS; every lane keeps two row points in
// registers and walks 32 columns, so each wave writes one contiguous 1 KiB column segment
// (16 B per lane) per step: the kernel is a pure coalesced HBM write stream with ~30 fp64
// VALU ops per entry hidden behind it.  Replaces the >= 6 N^2 passes + N `torch.minimum`
// launches of the reference (src/gp/wiener_kernel.py:21-22, RBFKernel/ScaleKernel/AddedDiag
// at src/batt_models/cell_gp.py:32-36).
#include "bgp_internal.h"

namespace {

// exp(x) for x <= 0 (finite).  Cody-Waite reduction + degree-13 Taylor/Horner on
// |r| <= ln2/2 (truncation 4e-18), scaled by v_ldexp_f64 (correct gradual underflow).
__device__ __forceinline__ double exp_nonpos(double x) {
  const double L2E = 1.44269504088896338700e+00;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  double n = __builtin_rint(x * L2E);
  double r = __builtin_fma(n, -LN2_HI, x);
  r = __builtin_fma(n, -LN2_LO, r);
  double p = 1.6059043836821613e-10;            // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);   // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);  // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);  // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06); // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);   // 1/8!
  p = __builtin_fma(p, r, 1.984126984126984e-04);  // 1/7!
  p = __builtin_fma(p, r, 1.3888888888888889e-03); // 1/6!
  p = __builtin_fma(p, r, 8.333333333333333e-03);  // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  // n >= -1075 matters only; clamp so the int conversion is safe for huge |x|
  int ni = (int)__builtin_fmax(n, -2000.0);
  return __builtin_ldexp(p, ni);
}

template <int KID, int DD>
__device__ __forceinline__ double kfun(const double (&a)[DD], const double (&b)[DD], int D,
                                       double s0, double s1) {
  if (KID == BGP_KERNEL_BATTGP) {
    // s_w (m^3/3 + |dt| m^2/2) + s_r exp(-sum_d ((a_d-b_d)/(l_d sqrt2))^2),  d = 1..D-1
    double q = 0.0;
#pragma unroll
    for (int d = 1; d < DD; ++d)
      if (d < D) {
        double df = a[d] - b[d];
        q = __builtin_fma(df, df, q);
      }
    double e = exp_nonpos(-q);
    double m = __builtin_fmin(a[0], b[0]);
    double ad = __builtin_fabs(a[0] - b[0]);
    double m2 = m * m;
    double w = (m2 * m) * (1.0 / 3.0) + (ad * m2) * 0.5;
    return __builtin_fma(s0, w, s1 * e);
  } else if (KID == BGP_KERNEL_MATERN32) {
    double q = 0.0;
#pragma unroll
    for (int d = 0; d < DD; ++d)
      if (d < D) {
        double df = a[d] - b[d];
        q = __builtin_fma(df, df, q);
      }
    double r = __builtin_sqrt(q);  // inputs pre-scaled by sqrt(3)/l: r = sqrt(3) * dist
    return s0 * (1.0 + r) * exp_nonpos(-r);
  } else {  // SCALED_RBF / ARD_RBF: inputs pre-scaled by 1/(l sqrt2)
    double q = 0.0;
#pragma unroll
    for (int d = 0; d < DD; ++d)
      if (d < D) {
        double df = a[d] - b[d];
        q = __builtin_fma(df, df, q);
      }
    return s0 * exp_nonpos(-q);
  }
}

template <int KID, int DD>
__device__ __forceinline__ void load_point(const double* __restrict__ x, int64_t idx, int64_t nvalid,
                                           int D, const FillParams& p, double (&a)[DD]) {
#pragma unroll
  for (int d = 0; d < DD; ++d) {
    double v = 0.0;
    if (d < D && idx < nvalid) {
      v = x[idx * D + d];
      if (!(KID == BGP_KERNEL_BATTGP && d == 0)) v *= p.scale[d];
    }
    a[d] = v;
  }
}

// decode a linear lower-triangular tile index t = ti (ti+1)/2 + tj, tj <= ti
__device__ __forceinline__ void tri_decode(int64_t t, int& ti, int& tj) {
  int64_t i = (int64_t)((__builtin_sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  while (i * (i + 1) / 2 > t) --i;
  ti = (int)i;
  tj = (int)(t - i * (i + 1) / 2);
}

template <int KID, int DT>
__global__ __launch_bounds__(256) void fill_kernel(FillParams p, const double* __restrict__ x1,
                                                   int64_t n1, const double* __restrict__ x2,
                                                   int64_t n2, double* __restrict__ out, int64_t ld,
                                                   int lower, int add_diag, int64_t nv1, int64_t nv2,
                                                   int nti, int vec_ok) {
  constexpr int DD = DT ? DT : BGP_MAX_DIM;
  const int D = DT ? DT : p.D;
  __shared__ double sB[128][DD];

  int ti, tj;
  if (lower) {
    tri_decode((int64_t)blockIdx.x, ti, tj);
  } else {
    ti = (int)(blockIdx.x % (unsigned)nti);
    tj = (int)(blockIdx.x / (unsigned)nti);
  }
  const int64_t i0 = (int64_t)ti * 128, j0 = (int64_t)tj * 128;

  for (int idx = threadIdx.x; idx < 128 * DD; idx += 256) {
    int c = idx / DD, d = idx % DD;
    int64_t j = j0 + c;
    double v = 0.0;
    if (j < nv2 && d < D) {
      v = x2[j * D + d];
      if (!(KID == BGP_KERNEL_BATTGP && d == 0)) v *= p.scale[d];
    }
    sB[c][d] = v;
  }

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t i = i0 + 2 * lane;
  double a0[DD], a1[DD];
  load_point<KID, DD>(x1, i, nv1, D, p, a0);
  load_point<KID, DD>(x1, i + 1, nv1, D, p, a1);
  __syncthreads();

  const double s0 = p.s0, s1 = p.s1, noise = p.noise;
  const bool r0_in = i < n1, r1_in = i + 1 < n1;
  const bool r0_val = i < nv1, r1_val = i + 1 < nv1;
#pragma unroll 4
  for (int cc = 0; cc < 32; ++cc) {
    const int c = wave * 32 + cc;
    const int64_t j = j0 + c;
    if (j >= n2) break;
    double b[DD];
#pragma unroll
    for (int d = 0; d < DD; ++d) b[d] = sB[c][d];
    double v0 = kfun<KID, DD>(a0, b, D, s0, s1);
    double v1 = kfun<KID, DD>(a1, b, D, s0, s1);
    const bool cval = j < nv2;
    if (add_diag) {
      // training fill: + noise on the diagonal; padding rows/cols form an identity block
      if (!(cval && r0_val)) v0 = (i == j) ? 1.0 : 0.0;
      else if (i == j) v0 += noise;
      if (!(cval && r1_val)) v1 = (i + 1 == j) ? 1.0 : 0.0;
      else if (i + 1 == j) v1 += noise;
    } else {
      if (!(cval && r0_val)) v0 = 0.0;
      if (!(cval && r1_val)) v1 = 0.0;
    }
    double* dst = out + i + j * ld;
    if (vec_ok && r1_in) {
      *reinterpret_cast<double2*>(dst) = make_double2(v0, v1);
    } else {
      if (r0_in) dst[0] = v0;
      if (r1_in) dst[1] = v1;
    }
  }
}

template <int KID>
int fill_dispatch(hipStream_t st, const FillParams& p, const double* x1, int64_t n1, const double* x2,
                  int64_t n2, double* out, int64_t ld, int lower, int add_diag, int64_t nv1,
                  int64_t nv2) {
  const int nti = (int)((n1 + 127) / 128), ntj = (int)((n2 + 127) / 128);
  int64_t nblocks = lower ? (int64_t)nti * (nti + 1) / 2 : (int64_t)nti * ntj;
  if (nblocks <= 0) return 0;
  if (nblocks > 0x7fffffffLL) return -1;
  const int vec_ok = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  dim3 grid((unsigned)nblocks), block(256);
  if (p.D == 4)
    hipLaunchKernelGGL((fill_kernel<KID, 4>), grid, block, 0, st, p, x1, n1, x2, n2, out, ld, lower,
                       add_diag, nv1, nv2, nti, vec_ok);
  else
    hipLaunchKernelGGL((fill_kernel<KID, 0>), grid, block, 0, st, p, x1, n1, x2, n2, out, ld, lower,
                       add_diag, nv1, nv2, nti, vec_ok);
  return 0;
}

// r[i] = sum_j k(x_i, x_j) v_j + diag_add v_i  (Sigma re-evaluated on the fly; residual check)
template <int KID>
__global__ __launch_bounds__(256) void kmatvec_kernel(FillParams p, const double* __restrict__ x,
                                                      int64_t n, const double* __restrict__ v,
                                                      double diag_add, double* __restrict__ out) {
  constexpr int DD = BGP_MAX_DIM;
  const int D = p.D;
  __shared__ double sB[256][DD];
  __shared__ double sV[256];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double a[DD];
  load_point<KID, DD>(x, i, n, D, p, a);
  double acc = 0.0, comp = 0.0;  // Kahan: this is a check, make it trustworthy
  for (int64_t j0 = 0; j0 < n; j0 += 256) {
    __syncthreads();
    {
      double b[DD];
      load_point<KID, DD>(x, j0 + threadIdx.x, n, D, p, b);
#pragma unroll
      for (int d = 0; d < DD; ++d) sB[threadIdx.x][d] = b[d];
      sV[threadIdx.x] = (j0 + threadIdx.x < n) ? v[j0 + threadIdx.x] : 0.0;
    }
    __syncthreads();
    const int lim = (int)((n - j0 < 256) ? (n - j0) : 256);
    for (int c = 0; c < lim; ++c) {
      double b[DD];
#pragma unroll
      for (int d = 0; d < DD; ++d) b[d] = sB[c][d];
      double kv = kfun<KID, DD>(a, b, D, p.s0, p.s1);
      if (j0 + c == i) kv += diag_add;
      double term = kv * sV[c] - comp;
      double t = acc + term;
      comp = (t - acc) - term;
      acc = t;
    }
  }
  if (i < n) out[i] = acc;
}

// sampled check of (L L^T)_ij against Sigma_ij: one wave per sample
template <int KID>
__global__ __launch_bounds__(64) void llt_sample_kernel(FillParams p, const double* __restrict__ x,
                                                        const double* __restrict__ L, int64_t lda,
                                                        int64_t n, double diag_add, int nsample,
                                                        double* __restrict__ out_err) {
  constexpr int DD = BGP_MAX_DIM;
  const int s = blockIdx.x;
  // deterministic pseudo-random (i >= j) pair, biased towards the far corner too
  uint64_t hsh = 0x9E3779B97F4A7C15ull * (uint64_t)(s + 1);
  hsh ^= hsh >> 29; hsh *= 0xBF58476D1CE4E5B9ull; hsh ^= hsh >> 32;
  int64_t i = (int64_t)(hsh % (uint64_t)n);
  hsh *= 0x94D049BB133111EBull; hsh ^= hsh >> 31;
  int64_t j = (int64_t)(hsh % (uint64_t)(i + 1));
  if (s == 0) { i = n - 1; j = n - 1; }
  if (s == 1) { i = n - 1; j = 0; }
  double acc = 0.0;
  for (int64_t q = threadIdx.x; q <= j; q += 64) acc = __builtin_fma(L[i + q * lda], L[j + q * lda], acc);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (threadIdx.x == 0) {
    double a[DD], b[DD];
    load_point<KID, DD>(x, i, n, p.D, p, a);
    load_point<KID, DD>(x, j, n, p.D, p, b);
    double kij = kfun<KID, DD>(a, b, p.D, p.s0, p.s1) + (i == j ? diag_add : 0.0);
    double kii = kfun<KID, DD>(a, a, p.D, p.s0, p.s1) + diag_add;
    double kjj = kfun<KID, DD>(b, b, p.D, p.s0, p.s1) + diag_add;
    out_err[s] = __builtin_fabs(acc - kij) / __builtin_sqrt(kii * kjj);
  }
}

}  // namespace

#define BGP_KID_SWITCH(kid, CALL)                                  \
  switch (kid) {                                                   \
    case BGP_KERNEL_BATTGP: { constexpr int KID_ = BGP_KERNEL_BATTGP; CALL; } break;       \
    case BGP_KERNEL_SCALED_RBF: { constexpr int KID_ = BGP_KERNEL_SCALED_RBF; CALL; } break; \
    case BGP_KERNEL_MATERN32: { constexpr int KID_ = BGP_KERNEL_MATERN32; CALL; } break;   \
    case BGP_KERNEL_ARD_RBF: { constexpr int KID_ = BGP_KERNEL_ARD_RBF; CALL; } break;     \
    default: return bgp_fail(h, -1, "unknown kernel id %d", kid);  \
  }

int launch_fill(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x1, int64_t n1,
                const double* x2, int64_t n2, double* out, int64_t ld, int lower, int add_diag,
                int64_t nvalid1, int64_t nvalid2) {
  int rc = 0;
  BGP_KID_SWITCH(p.kid, rc = fill_dispatch<KID_>(st, p, x1, n1, x2, n2, out, ld, lower, add_diag,
                                                 nvalid1, nvalid2));
  if (rc) return bgp_fail(h, -1, "fill: grid too large");
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_kmatvec(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x, int64_t n,
                   const double* v, double diag_add, double* out) {
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  BGP_KID_SWITCH(p.kid, hipLaunchKernelGGL((kmatvec_kernel<KID_>), grid, block, 0, st, p, x, n, v,
                                           diag_add, out));
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_llt_sample(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x,
                      const double* L, int64_t lda, int64_t n, double diag_add, int nsample,
                      double* out_err) {
  dim3 grid((unsigned)nsample), block(64);
  BGP_KID_SWITCH(p.kid, hipLaunchKernelGGL((llt_sample_kernel<KID_>), grid, block, 0, st, p, x, L, lda,
                                           n, diag_add, nsample, out_err));
  BGP_HIP(h, hipGetLastError());
  return 0;
}
