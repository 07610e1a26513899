// Dense linear-algebra kernels of the blocked Cholesky path (gfx950, fp64 MFMA).
//
// Everything is column-major.  The one hot kernel is gemm_nt_kernel:
//     C[m,n] -= A[m,k] * B[n,k]^T        (MODE 0; `lower` keeps only tiles touching i >= j)
//     C[m,n]  = A[m,k] * B[n,k]^T        (MODE 1; C may alias A: used as TRSM-by-inverse)
// built on v_mfma_f64_16x16x4_f64.  It serves the SYRK trailing update of the Cholesky, the
// in-panel updates, the TRSM by inverted 64x64 diagonal tiles, the triangular solve of the
// query block (posterior variance) and the full-covariance downdate.
#include "bgp_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 16;  // k-depth of one LDS stage

// Map a linear block id to a tile.  Tiles are grouped in 8x8 super-tiles and super-tiles are
// dealt round-robin to the 8 XCDs (block b runs on XCD b % 8 - observed, speed only), so the
// ~64 workgroups resident on one XCD share 16 operand panels in that XCD's private L2.
__device__ __forceinline__ bool map_tile(int64_t b, int nti, int ntj, int lower, int& ti, int& tj) {
  const int nsi = (nti + 7) >> 3;
  int nsj = (ntj + 7) >> 3;
  if (lower && nsj > nsi) nsj = nsi;  // tiles with tj > ti are never needed
  const int64_t slot = b >> 3;
  const int xcd = (int)(b & 7);
  const int64_t s = (slot >> 6) * 8 + xcd;
  const int w = (int)(slot & 63);
  int si, sj;
  if (lower) {
    // valid super pairs: sj <= si, sj < nsj.  First the triangle si < nsj, then full rows.
    const int64_t ntri = (int64_t)nsj * (nsj + 1) / 2;
    const int64_t total = ntri + (int64_t)(nsi > nsj ? nsi - nsj : 0) * nsj;
    if (s >= total) return false;
    if (s < ntri) {
      int64_t i = (int64_t)((__builtin_sqrt(8.0 * (double)s + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= s) ++i;
      while (i * (i + 1) / 2 > s) --i;
      si = (int)i;
      sj = (int)(s - i * (i + 1) / 2);
    } else {
      const int64_t r = s - ntri;
      si = nsj + (int)(r / nsj);
      sj = (int)(r % nsj);
    }
  } else {
    if (s >= (int64_t)nsi * nsj) return false;
    si = (int)(s % nsi);
    sj = (int)(s / nsi);
  }
  ti = si * 8 + (w & 7);
  tj = sj * 8 + (w >> 3);
  if (ti >= nti || tj >= ntj) return false;
  if (lower && ti < tj) return false;
  return true;
}

__host__ int64_t gemm_grid_blocks(int nti, int ntj, int lower) {
  const int64_t nsi = (nti + 7) >> 3, nsj = (ntj + 7) >> 3;
  int64_t total;
  if (lower) {
    const int64_t nsjc = nsj < nsi ? nsj : nsi;
    total = nsjc * (nsjc + 1) / 2 + (nsi > nsjc ? nsi - nsjc : 0) * nsjc;
  } else {
    total = nsi * nsj;
  }
  return ((total + 7) / 8) * 8 * 64;
}

template <int TM, int TN, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(double* C, int64_t ldc, const double* A,
                                                         int64_t lda, const double* B, int64_t ldb,
                                                         int64_t m, int64_t n, int k, int lower, int nti,
                                                         int ntj) {
  constexpr int LDA_S = TM + 16;  // (ld % 32 == 16) => the two 16-lane groups of a ds_read_b64
  constexpr int LDB_S = TN + 16;  //  half-wave hit disjoint bank halves: conflict-free
  constexpr int MI = TM / 32, MJ = TN / 32;          // 16x16 MFMA tiles per wave along i / j
  constexpr int RPA = TM / 2, RPB = TN / 2;          // double2 per staged column
  constexpr int NLA = TM / 32, NLB = TN / 32;        // double2 loads per thread per stage
  constexpr int CSA = 256 / RPA, CSB = 256 / RPB;    // column stride between a thread's loads
  __shared__ __attribute__((aligned(16))) double sA[2][BK][LDA_S];
  __shared__ __attribute__((aligned(16))) double sB[2][BK][LDB_S];

  int ti, tj;
  if (!map_tile((int64_t)blockIdx.x, nti, ntj, lower, ti, tj)) return;
  const int64_t i0 = (int64_t)ti * TM, j0 = (int64_t)tj * TN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave & 1, wj = wave >> 1;

  // staging coordinates
  const int ra = (tid % RPA) * 2, ca = tid / RPA;
  const int rb = (tid % RPB) * 2, cb = tid / RPB;
  const bool a_in = (i0 + ra) < m;  // m, n even by contract of the launcher
  const bool b_in = (j0 + rb) < n;
  const double* gA = A + (i0 + ra) + (int64_t)ca * lda;
  const double* gB = B + (j0 + rb) + (int64_t)cb * ldb;

  double2 regA[NLA], regB[NLB];
  auto gload = [&](int kt) {
    const int64_t koff = (int64_t)kt * BK;
#pragma unroll
    for (int q = 0; q < NLA; ++q)
      regA[q] = a_in ? *reinterpret_cast<const double2*>(gA + (koff + q * CSA) * lda)
                     : make_double2(0.0, 0.0);
#pragma unroll
    for (int q = 0; q < NLB; ++q)
      regB[q] = b_in ? *reinterpret_cast<const double2*>(gB + (koff + q * CSB) * ldb)
                     : make_double2(0.0, 0.0);
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NLA; ++q)
      *reinterpret_cast<double2*>(&sA[buf][ca + q * CSA][ra]) = regA[q];
#pragma unroll
    for (int q = 0; q < NLB; ++q)
      *reinterpret_cast<double2*>(&sB[buf][cb + q * CSB][rb]) = regB[q];
  };

  v4d acc[MJ][MI];
#pragma unroll
  for (int a = 0; a < MJ; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

  const int nk = k / BK;
  gload(0);
  sstore(0);
  __syncthreads();

  const int l15 = lane & 15, l4 = lane >> 4;
  const int ibase = wi * (TM / 2) + l15, jbase = wj * (TN / 2) + l15;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int pp = kk * 4 + l4;
      double fa[MJ], fb[MI];
#pragma unroll
      for (int a = 0; a < MJ; ++a) fa[a] = sB[buf][pp][jbase + a * 16];
#pragma unroll
      for (int b = 0; b < MI; ++b) fb[b] = sA[buf][pp][ibase + b * 16];
#pragma unroll
      for (int a = 0; a < MJ; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane (l15, l4), reg r of tile (a, b) holds (A B^T)(i, j) with
  //   i = i0 + wi*TM/2 + b*16 + l15,   j = j0 + wj*TN/2 + a*16 + l4 + 4 r
#pragma unroll
  for (int a = 0; a < MJ; ++a) {
#pragma unroll
    for (int b = 0; b < MI; ++b) {
      const int64_t i = i0 + wi * (TM / 2) + b * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t j = j0 + wj * (TN / 2) + a * 16 + l4 + 4 * r;
        if (i < m && j < n) {
          double* cp = C + i + j * ldc;
          if (MODE == 0) *cp = *cp - acc[a][b][r];
          else *cp = acc[a][b][r];
        }
      }
    }
  }
}

// ---- 64x64 diagonal tile: Cholesky + explicit inverse ------------------------------------
// One workgroup.  The tile lives in LDS column-major (s[c][r] = A(r,c)): column walks are
// contiguous and pivot-row reads are broadcasts, so the factorisation is bank-conflict free.
// info (global, 0 = ok) receives col0 + j + 1 for the first non-positive pivot.
__global__ __launch_bounds__(256) void potrf_tile_kernel(double* __restrict__ Ajj, int64_t lda,
                                                         double* __restrict__ inv,
                                                         int* __restrict__ info, int col0) {
  __shared__ double s[64][64];
  __shared__ double sx[64][64];
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  const int row = tid & 63, wv = tid >> 6;
  if (tid == 0) s_fail = 0;
  for (int idx = tid; idx < 4096; idx += 256) {
    const int r = idx & 63, c = idx >> 6;
    s[c][r] = (r >= c) ? Ajj[r + (int64_t)c * lda] : 0.0;
  }
  __syncthreads();

  for (int j = 0; j < 64; ++j) {
    const double ajj = s[j][j];
    if (!(ajj > 0.0) && tid == 0 && s_fail == 0) s_fail = j + 1;  // also catches NaN
    const double d = __builtin_sqrt(ajj);
    const double rd = 1.0 / d;
    __syncthreads();  // everybody has read the pivot
    if (tid > j && tid < 64) s[j][tid] *= rd;
    if (tid == j) s[j][j] = d;
    __syncthreads();
    // rank-1 update of the trailing lower triangle: lane <-> row, wave w takes columns j+1+w (+4..)
    const double lrj = s[j][row];
    for (int c = j + 1 + wv; c < 64; c += 4)
      if (row >= c) s[c][row] = __builtin_fma(-lrj, s[j][c], s[c][row]);
    __syncthreads();
  }

  for (int idx = tid; idx < 4096; idx += 256) {
    const int r = idx & 63, c = idx >> 6;
    if (r >= c) Ajj[r + (int64_t)c * lda] = s[c][r];
  }
  if (tid == 0 && s_fail != 0) atomicCAS(info, 0, col0 + s_fail);

  // X = L^-1, one thread per column c (sx[p][c] = X(p,c)):
  //   x_c = 1/L_cc,  x_i = -(sum_{p=c}^{i-1} L_ip x_p) / L_ii
  if (tid < 64) {
    const int c = tid;
    for (int i = 0; i < c; ++i) sx[i][c] = 0.0;
    sx[c][c] = 1.0 / s[c][c];
    for (int i = c + 1; i < 64; ++i) {
      double acc = 0.0;
      for (int p = c; p < i; ++p) acc = __builtin_fma(s[p][i], sx[p][c], acc);
      sx[i][c] = -acc / s[i][i];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < 4096; idx += 256) {
    const int c = idx & 63, r = idx >> 6;
    inv[r + c * 64] = sx[r][c];
  }
}

// out[0] = sum_i log A_ii, out[1] = sum_i z_i^2   (z strided by ldz).  One workgroup.
__global__ __launch_bounds__(1024) void fit_scalars_kernel(const double* __restrict__ A, int64_t lda,
                                                           const double* __restrict__ z, int64_t ldz,
                                                           int64_t n, double* __restrict__ out) {
  __shared__ double r0[1024], r1[1024];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    a += log(A[i + i * lda]);
    const double zi = z[i * ldz];
    b = __builtin_fma(zi, zi, b);
  }
  r0[threadIdx.x] = a;
  r1[threadIdx.x] = b;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      r0[threadIdx.x] += r0[threadIdx.x + s];
      r1[threadIdx.x] += r1[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = r0[0];
    out[1] = r1[0];
  }
}

// ---- backward solve  alpha = L^-T z, one 64-column block at a time ---------------------------
// part[chunk][c] = sum_{r in chunk} L[r, c] x[r]  for the 64 columns of the panel.
constexpr int GV_ROWS = 4096;  // rows per workgroup
__global__ __launch_bounds__(256) void gemvt_partial_kernel(const double* __restrict__ Lp, int64_t lda,
                                                            const double* __restrict__ x, int64_t rows,
                                                            double* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * GV_ROWS;
  const int64_t r1 = (r0 + GV_ROWS < rows) ? r0 + GV_ROWS : rows;
  for (int cc = 0; cc < 16; ++cc) {
    const int c = wave * 16 + cc;
    const double* col = Lp + (int64_t)c * lda;
    double acc = 0.0;
    for (int64_t r = r0 + lane; r < r1; r += 64) acc = __builtin_fma(col[r], x[r], acc);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) part[(int64_t)blockIdx.x * 64 + c] = acc;
  }
}

// w = z_j - sum_chunks part;  alpha_j = inv^T w
__global__ __launch_bounds__(64) void solve_tile_t_kernel(const double* __restrict__ inv,
                                                          const double* __restrict__ z, int64_t ldz,
                                                          const double* __restrict__ part, int nchunks,
                                                          double* __restrict__ alpha_j) {
  __shared__ double w[64];
  const int c = threadIdx.x;
  double v = z[(int64_t)c * ldz];
  for (int q = 0; q < nchunks; ++q) v -= part[(int64_t)q * 64 + c];
  w[c] = v;
  __syncthreads();
  double acc = 0.0;
  for (int p = c; p < 64; ++p) acc = __builtin_fma(inv[p + c * 64], w[p], acc);
  alpha_j[c] = acc;
}

// ---- row-wise reductions over the query block E[M, n] (column-major, rows contiguous) ---------
// part[chunk][m] = sum_{i in chunk} E[m,i] * (vec ? vec[i] : E[m,i])
constexpr int RD_COLS = 512;
__global__ __launch_bounds__(256) void rowdot_kernel(const double* __restrict__ E, int64_t lde, int64_t M,
                                                     int64_t n, const double* __restrict__ vec,
                                                     double* __restrict__ part) {
  const int64_t mrow = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t c0 = (int64_t)blockIdx.y * RD_COLS;
  const int64_t c1 = (c0 + RD_COLS < n) ? c0 + RD_COLS : n;
  if (mrow >= M) return;
  double acc = 0.0;
  if (vec) {
    for (int64_t i = c0; i < c1; ++i) acc = __builtin_fma(E[mrow + i * lde], vec[i], acc);
  } else {
    for (int64_t i = c0; i < c1; ++i) {
      const double e = E[mrow + i * lde];
      acc = __builtin_fma(e, e, acc);
    }
  }
  part[(int64_t)blockIdx.y * M + mrow] = acc;
}

// mean:  out[m] = sum_chunks part         (kdiag == 0)
// var :  out[m] = max(kdiag(xq_m) - sum_chunks part, min_var)
__global__ __launch_bounds__(256) void rowdot_finish_kernel(const double* __restrict__ part, int nchunks,
                                                            int64_t M, const double* __restrict__ xq,
                                                            FillParams p, int is_var, double min_var,
                                                            double* __restrict__ out) {
  const int64_t mrow = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (mrow >= M) return;
  // pairwise-ish: sum in two interleaved accumulators to keep the dependency chain short
  double a0 = 0.0, a1 = 0.0;
  int q = 0;
  for (; q + 1 < nchunks; q += 2) {
    a0 += part[(int64_t)q * M + mrow];
    a1 += part[(int64_t)(q + 1) * M + mrow];
  }
  if (q < nchunks) a0 += part[(int64_t)q * M + mrow];
  double sum = a0 + a1;
  if (is_var) {
    double kd;
    if (p.kid == BGP_KERNEL_BATTGP) {
      const double t = xq[mrow * p.D];
      kd = p.s0 * ((t * t * t) * (1.0 / 3.0)) + p.s1;  // diag branch of wiener_kernel.py:15-16
    } else {
      kd = p.s0;
    }
    double v = kd - sum;
    if (min_var >= 0.0) v = __builtin_fmax(v, min_var);
    out[mrow] = v;
  } else {
    out[mrow] = sum;
  }
}

__global__ __launch_bounds__(1024) void norm2_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                     int64_t n, double* __restrict__ out) {
  __shared__ double r0[1024];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double d = b ? a[i] - b[i] : a[i];
    acc = __builtin_fma(d, d, acc);
  }
  r0[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) r0[threadIdx.x] += r0[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = r0[0];
}

__global__ __launch_bounds__(256) void copy_strided_kernel(const double* __restrict__ src, int64_t n,
                                                           double* __restrict__ dst, int64_t ld,
                                                           int64_t npad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < npad) dst[i * ld] = (i < n) ? src[i] : 0.0;
}

}  // namespace

int launch_gemm_nt(bgp_handle* h, hipStream_t st, int mode, int tn, double* C, int64_t ldc,
                   const double* A, int64_t lda, const double* B, int64_t ldb, int64_t m, int64_t n,
                   int64_t k, int lower) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  if ((k % BK) != 0) return bgp_fail(h, -1, "gemm_nt: k=%lld not a multiple of %d", (long long)k, BK);
  if ((m & 1) || (n & 1) || (lda & 1) || (ldb & 1))
    return bgp_fail(h, -1, "gemm_nt: m, n, lda, ldb must be even (m=%lld n=%lld lda=%lld ldb=%lld)",
                    (long long)m, (long long)n, (long long)lda, (long long)ldb);
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15))
    return bgp_fail(h, -1, "gemm_nt: operands must be 16-byte aligned");
  const int TM = 128;
  const int nti = (int)((m + TM - 1) / TM), ntj = (int)((n + tn - 1) / tn);
  const int64_t blocks = gemm_grid_blocks(nti, ntj, lower);
  if (blocks > 0x7fffffffLL) return bgp_fail(h, -1, "gemm_nt: grid too large");
  dim3 grid((unsigned)blocks), block(256);
  if (tn == 128 && mode == 0)
    hipLaunchKernelGGL((gemm_nt_kernel<128, 128, 0>), grid, block, 0, st, C, ldc, A, lda, B, ldb, m, n,
                       (int)k, lower, nti, ntj);
  else if (tn == 64 && mode == 0)
    hipLaunchKernelGGL((gemm_nt_kernel<128, 64, 0>), grid, block, 0, st, C, ldc, A, lda, B, ldb, m, n,
                       (int)k, lower, nti, ntj);
  else if (tn == 64 && mode == 1)
    hipLaunchKernelGGL((gemm_nt_kernel<128, 64, 1>), grid, block, 0, st, C, ldc, A, lda, B, ldb, m, n,
                       (int)k, lower, nti, ntj);
  else
    return bgp_fail(h, -1, "gemm_nt: unsupported variant tn=%d mode=%d", tn, mode);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_potrf_tile(bgp_handle* h, hipStream_t st, double* Ajj, int64_t lda, double* inv, int* info,
                      int col0, int /*nvalid*/) {
  hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(256), 0, st, Ajj, lda, inv, info, col0);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_fit_scalars(bgp_handle* h, hipStream_t st, const double* A, int64_t lda, const double* z,
                       int64_t ldz, int64_t n, double* out2) {
  hipLaunchKernelGGL(fit_scalars_kernel, dim3(1), dim3(1024), 0, st, A, lda, z, ldz, n, out2);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_gemvt_partial(bgp_handle* h, hipStream_t st, const double* Lpanel, int64_t lda,
                         const double* x, int64_t rows, double* part, int* nchunks_out) {
  const int nch = (int)((rows + GV_ROWS - 1) / GV_ROWS);
  *nchunks_out = nch;
  if (nch == 0) return 0;
  hipLaunchKernelGGL(gemvt_partial_kernel, dim3(nch), dim3(256), 0, st, Lpanel, lda, x, rows, part);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_solve_tile_t(bgp_handle* h, hipStream_t st, const double* inv, const double* z, int64_t ldz,
                        const double* part, int nchunks, double* alpha_j) {
  hipLaunchKernelGGL(solve_tile_t_kernel, dim3(1), dim3(64), 0, st, inv, z, ldz, part, nchunks, alpha_j);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_rowdot(bgp_handle* h, hipStream_t st, const double* E, int64_t lde, int64_t M, int64_t n,
                  const double* vec, double* part, int* nchunks_out) {
  const int nch = (int)((n + RD_COLS - 1) / RD_COLS);
  *nchunks_out = nch;
  dim3 grid((unsigned)((M + 255) / 256), (unsigned)nch);
  hipLaunchKernelGGL(rowdot_kernel, grid, dim3(256), 0, st, E, lde, M, n, vec, part);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_rowdot_finish(bgp_handle* h, hipStream_t st, const double* part, int nchunks, int64_t M,
                         const double* kdiag_x, const FillParams* p, double min_var, double* out) {
  FillParams pp = *p;
  hipLaunchKernelGGL(rowdot_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, part,
                     nchunks, M, kdiag_x, pp, kdiag_x != nullptr ? 1 : 0, min_var, out);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_norm2(bgp_handle* h, hipStream_t st, const double* a, const double* b, int64_t n, double* out) {
  hipLaunchKernelGGL(norm2_kernel, dim3(1), dim3(1024), 0, st, a, b, n, out);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_copy_strided(bgp_handle* h, hipStream_t st, const double* src, int64_t n, double* dst,
                        int64_t ld_dst, int64_t npad) {
  hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, st, src, n,
                     dst, ld_dst, npad);
  BGP_HIP(h, hipGetLastError());
  return 0;
}
