// Dense linear-algebra kernels of the blocked Cholesky path (gfx950, fp64 MFMA).
//
// Everything is column-major.  The one hot kernel is gemm_nt_kernel:
//     C[m,n] -= A[m,k] * B[n,k]^T        (MODE 0; `lower` keeps only tiles touching i >= j)
//     C[m,n]  = A[m,k] * B[n,k]^T        (MODE 1; C may alias A: used as TRSM-by-inverse)
//     C[m,n] += -(A B^T) by L2 atomics   (MODE 2; no C read: the deep rank-NB trailing updates)
//   (the `C += A B^T` steps of the gradient's in-place M^T M run as MODE 2 / 0 against a negated copy of B)
// built on v_mfma_f64_16x16x4_f64.  It serves the SYRK trailing update of the Cholesky, the
// in-panel updates, the TRSM by inverted 64x64 diagonal tiles, the triangular solve of the
// query block (posterior variance) and the full-covariance downdate.
#include <type_traits>

#include "bgp_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

// register budget of a kernel = 512 / n VGPRs per lane (a host build of these sources for tests defines it away)
#ifndef BGP_WAVES_PER_EU
#define BGP_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#define BGP_GLOBAL_AS __attribute__((address_space(1)))
#endif

namespace {

constexpr int BK = 16;  // k-depth of one LDS stage

// Map a linear block id to a tile.  Tiles are grouped in 8x8 super-tiles and super-tiles are
// dealt round-robin to the 8 XCDs (block b runs on XCD b % 8 - observed, speed only), so the
// ~64 workgroups resident on one XCD share 16 operand panels in that XCD's private L2.
#ifndef BGP_SUPER_LOG_SI
#define BGP_SUPER_LOG_SI 3  // super-tile = 2^SI x 2^(6-SI) tiles (64 per super-tile); 8 x 8 by default
#endif
__device__ __forceinline__ bool map_tile(int64_t b, int nti, int ntj, int lower, int& ti, int& tj) {
  if (lower & 2) {
    // small grids (the latency-bound panel kernels): one block per tile, no padding blocks that
    // would each queue for a CU slot behind the trailing update
    ti = (int)(b % nti);
    tj = (int)(b / nti);
    return tj < ntj && !((lower & 1) && ti < tj);
  }
  constexpr int LSI = BGP_SUPER_LOG_SI, LSJ = 6 - BGP_SUPER_LOG_SI;
  constexpr int SI = 1 << LSI, SJ = 1 << LSJ;
  const int nsi = (nti + SI - 1) >> LSI;
  int nsj = (ntj + SJ - 1) >> LSJ;
  const int64_t slot = b >> 3;
  const int xcd = (int)(b & 7);
  const int64_t s = (slot >> 6) * 8 + xcd;
  const int w = (int)(slot & 63);
  int si, sj;
  if (lower) {
    // a super-tile (si, sj) is needed iff it contains a tile with ti >= tj:  (si+1) SI - 1 >= sj SJ
    // rows of super-tiles: row si has min(nsj, ((si+1) SI - 1) / SJ + 1) valid super-tiles
    auto rowcount = [&](int64_t r) {
      int64_t c = (((r + 1) << LSI) - 1) / SJ + 1;
      return c < nsj ? c : (int64_t)nsj;
    };
    // linear search over super-tile rows would be O(nsi); use the closed form of the prefix for the
    // square case (SI == SJ) and a short loop otherwise
    if (SI == SJ) {
      if (nsj > nsi) nsj = nsi;
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
      int64_t acc = 0, r = 0;
      for (; r < nsi; ++r) {
        const int64_t c = rowcount(r);
        if (s < acc + c) break;
        acc += c;
      }
      if (r >= nsi) return false;
      si = (int)r;
      sj = (int)(s - acc);
    }
  } else {
    if (s >= (int64_t)nsi * nsj) return false;
    si = (int)(s % nsi);
    sj = (int)(s / nsi);
  }
  ti = si * SI + (w & (SI - 1));
  tj = sj * SJ + (w >> LSI);
  if (ti >= nti || tj >= ntj) return false;
  if (lower && ti < tj) return false;
  return true;
}

__host__ int64_t gemm_grid_blocks(int nti, int ntj, int lower) {
  constexpr int LSI = BGP_SUPER_LOG_SI, LSJ = 6 - BGP_SUPER_LOG_SI;
  constexpr int SI = 1 << LSI, SJ = 1 << LSJ;
  const int64_t nsi = (nti + SI - 1) >> LSI, nsj = (ntj + SJ - 1) >> LSJ;
  int64_t total = 0;
  if (lower) {
    for (int64_t r = 0; r < nsi; ++r) {
      int64_t c = (((r + 1) << LSI) - 1) / SJ + 1;
      total += c < nsj ? c : nsj;
    }
  } else {
    total = nsi * nsj;
  }
  return ((total + 7) / 8) * 8 * 64;
}

// ABL != 0 builds ablation variants for tools/gemm_ablate.hip only (bit 0: no operand re-staging,
// bit 1: no barriers, bit 2: fragments read once) - production always instantiates ABL = 0.
template <int TM, int TN, int MODE, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(double* C, int64_t ldc, const double* A,
                                                         int64_t lda, const double* B, int64_t ldb,
                                                         int64_t m, int64_t n, int k, int lower, int nti,
                                                         int ntj, const int* __restrict__ abort_flag, int btri) {
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

  // A failed pivot earlier in the factorisation poisons the rest of the enqueued pipeline:
  // every later kernel sees the flag and returns at once (no host round trip needed).
  if (abort_flag != nullptr && *abort_flag != 0) return;

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

  // Out-of-range rows are clamped to a valid address and zeroed when staged: the loads stay
  // branch-free, so the whole k-step is one basic block the scheduler can interleave.
  const double* gA_safe = a_in ? gA : A + (int64_t)ca * lda;
  const double* gB_safe = b_in ? gB : B + (int64_t)cb * ldb;
  double2 regA[NLA], regB[NLB];
  auto gload = [&](int kt) {
    const int64_t koff = (int64_t)kt * BK;
#pragma unroll
    for (int q = 0; q < NLA; ++q)
      regA[q] = *reinterpret_cast<const double2*>(gA_safe + (koff + q * CSA) * lda);
#pragma unroll
    for (int q = 0; q < NLB; ++q)
      regB[q] = *reinterpret_cast<const double2*>(gB_safe + (koff + q * CSB) * ldb);
  };
  // No zeroing of the clamped out-of-range rows is needed: a garbage row of A (B) only reaches
  // accumulators of C rows >= m (columns >= n), which the epilogue never writes.
  // MODE 0/2 need -A B^T: the f64 MFMA negates its A operand when bit 0 of the last immediate is set.
  constexpr int NEG = (MODE == 0 || MODE == 2) ? 1 : 0;
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NLA; ++q)
      *reinterpret_cast<double2*>(&sA[buf][ca + q * CSA][ra]) = make_double2(regA[q].x, regA[q].y);
#pragma unroll
    for (int q = 0; q < NLB; ++q)
      *reinterpret_cast<double2*>(&sB[buf][cb + q * CSB][rb]) = make_double2(regB[q].x, regB[q].y);
  };

  const int l15 = lane & 15, l4 = lane >> 4;
  const int ibase = wi * (TM / 2) + l15, jbase = wj * (TN / 2) + l15;
  // btri (TRSM by an explicit lower-triangular inverse, B[j, kk] = 0 for kk > j): columns j0..j0+TN-1 of
  // the product only need kk < j0 + TN
  const int nk = (MODE == 1 && btri && j0 + TN < k) ? (int)((j0 + TN) / BK) : k / BK;
  gload(0);

  // MODE 0: the accumulators START as the C tile and the B operand is staged negated, so the
  // MFMA chain computes C - A B^T directly: the C reads overlap the first operand loads (one
  // HBM latency for the whole prologue) and the epilogue is a pure store stream.
  // lane (l15, l4), reg r of tile (a, b) <-> C(i, j),
  //   i = i0 + wi*TM/2 + b*16 + l15,   j = j0 + wj*TN/2 + a*16 + l4 + 4 r
  v4d acc[MJ][MI];
#pragma unroll
  for (int a = 0; a < MJ; ++a) {
#pragma unroll
    for (int b = 0; b < MI; ++b) {
      const int64_t i = i0 + wi * (TM / 2) + b * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t j = j0 + wj * (TN / 2) + a * 16 + l4 + 4 * r;
        acc[a][b][r] = (MODE == 0 && i < m && j < n) ? C[i + j * ldc] : 0.0;
      }
    }
  }
  sstore(0);
  if (nk > 1) gload(1);
  __syncthreads();

  // One barrier per k-step.  Registers always hold tile kt+1 (loaded a full k-step earlier);
  // it is written to the idle LDS buffer DURING step kt - that buffer was last read in step
  // kt-1, behind the previous barrier - and the loads of tile kt+2 are issued right after.
  // The staging instructions are placed behind the first fragment reads and spread one per
  // MFMA (sched_group_barrier), so they issue in the shadow of the 64-cycle MFMAs instead of
  // stalling the matrix pipe at the top of the step.
  auto kstep = [&](int kt, auto do_store, auto do_load) {
    const int buf = (ABL & 1) ? 0 : (kt & 1);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int pp = kk * 4 + l4;
      double fa[MJ], fb[MI];
#pragma unroll
      for (int a = 0; a < MJ; ++a) fa[a] = sB[buf][pp][jbase + a * 16];
#pragma unroll
      for (int b = 0; b < MI; ++b) fb[b] = sA[buf][pp][ibase + b * 16];
      if (kk == 0 && !(ABL & 1)) {
        if (decltype(do_store)::value) sstore(buf ^ 1);
        if (decltype(do_load)::value) gload(kt + 2);
      }
#pragma unroll
      for (int a = 0; a < MJ; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, NEG);
      if (kk == 0 && !(ABL & 1) && !(ABL & 8)) {
        if (decltype(do_store)::value) {
#pragma unroll
          for (int q = 0; q < NLA + NLB; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 1 DS write
          }
        }
        if (decltype(do_load)::value) {
#pragma unroll
          for (int q = 0; q < NLA + NLB; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
          }
        }
      }
    }
    if (!(ABL & 2)) __syncthreads();
  };
  {
    int kt = 0;
    for (; kt + 2 < nk; ++kt) kstep(kt, std::true_type{}, std::true_type{});
    if (kt + 1 < nk) { kstep(kt, std::true_type{}, std::false_type{}); ++kt; }
    if (kt < nk) kstep(kt, std::false_type{}, std::false_type{});
  }

  // epilogue: pure stores
#pragma unroll
  for (int a = 0; a < MJ; ++a) {
#pragma unroll
    for (int b = 0; b < MI; ++b) {
      const int64_t i = i0 + wi * (TM / 2) + b * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t j = j0 + wj * (TN / 2) + a * 16 + l4 + 4 * r;
        if (i < m && j < n) {
          if (MODE == 2) {
            // accumulators started at 0 and hold -A B^T: one fire-and-forget L2 atomic per element
            // (exactly one add per element per launch => deterministic); no C read in the prologue
            unsafeAtomicAdd(C + i + j * ldc, acc[a][b][r]);  // global_atomic_add_f64, no return
          } else {
            C[i + j * ldc] = acc[a][b][r];
          }
        }
      }
    }
  }
}

// ---- 64x64 diagonal tile: Cholesky + explicit inverse ------------------------------------
// Register-resident, 4 waves: lane = row, wave w owns columns 16w..16w+15 of the tile (32 VGPRs of
// data).  Right-looking column Cholesky: the wave that owns column j scales it (v_readlane pivot, rsq +
// Newton) and publishes it in a double-buffered 64-entry LDS column; after ONE barrier per step every
// wave applies it to its own columns > j (uniform-address LDS reads broadcast L(c,j)).  The critical
// path per step is {pivot, LDS write, barrier, LDS read, one FMA}: the 16-column updates of the other
// waves run in its shadow (single-wave version: 36 us per tile, this one ~20 us).  Then X = L^-1 by
// forward substitution in the same lane = row layout, columns dealt round-robin to the waves (column c
// costs 64 - c steps).  Fully unrolled: every register index and readlane lane-select is a constant.
// The arithmetic per element is the same sequence of FMAs as in the single-wave version.
__device__ __forceinline__ double readlane_d(double v, int srclane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(p) to ~1 ulp: v_rsq_f64 seed (~2^-26) + two Newton steps; p must be a normal number
__device__ __forceinline__ double rsqrt_newton(double p) {
  double y = __builtin_amdgcn_rsq(p);
  double hp = 0.5 * p;
  y = y * __builtin_fma(-hp * y, y, 1.5);
  y = y * __builtin_fma(-hp * y, y, 1.5);
  return y;
}

// wave W: columns C0..C0+15 of the tile, s[c][r] = A(r, c) on entry and L(r, c) (zero above the
// diagonal) on exit; every wave executes exactly 64 barriers, in three phases: (A) steps j < C0 update
// all 16 columns, (B) the 16 steps that own the pivot column, (C) steps j >= C0 + 16 only keep the
// barrier count
template <int W>
__device__ __forceinline__ void chol_cols(double (*s)[64], double (*colbuf)[64], int* sfail, int lane) {
  constexpr int NC = 16, C0 = NC * W;
  double a[NC];
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) a[cc] = s[C0 + cc][lane];
  // (A) a rolled loop is fine here: every register index is static
#pragma unroll 1
  for (int j = 0; j < C0; ++j) {
    __syncthreads();
    const double* col = colbuf[j & 1];
    const double lij = col[lane];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) a[cc] = __builtin_fma(-lij, col[C0 + cc], a[cc]);
  }
  // (B)
#pragma unroll
  for (int jl = 0; jl < NC; ++jl) {
    const int j = C0 + jl;
    const double piv = readlane_d(a[jl], j);
    if (!(piv > 0.0) && lane == 0 && *sfail == 0) *sfail = j + 1;  // also catches NaN; first failing j wins
    const double rd = rsqrt_newton(piv);
    double d = piv * rd;
    d = __builtin_fma(0.5 * rd, __builtin_fma(-d, d, piv), d);  // one correction: d = sqrt(piv)
    const double lij = (lane > j) ? a[jl] * rd : 0.0;
    a[jl] = (lane == j) ? d : lij;
    colbuf[j & 1][lane] = lij;
    __syncthreads();
    // my own columns right of j: L(c, j) straight from the owner's registers (no LDS round trip on the
    // critical path of the next pivot)
#pragma unroll
    for (int cc = jl + 1; cc < NC; ++cc) a[cc] = __builtin_fma(-lij, readlane_d(lij, C0 + cc), a[cc]);
  }
  // (C)
#pragma unroll 1
  for (int j = C0 + NC; j < 64; ++j) __syncthreads();
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) s[C0 + cc][lane] = (C0 + cc <= lane) ? a[cc] : 0.0;
}

// The same column Cholesky for the slim tile kernel: the 16 columns stay in the caller's registers, L(c, j) of the own
// columns comes out of the one LDS read of column j by v_readlane (scalar registers) instead of 16 more uniform LDS reads
// (vector registers: 32 VGPRs fewer alive), and `done(j, col, d)` is called by the owner wave the moment column j is
// final (col = L(lane, j) with d on the diagonal).  Same operations on the same values as chol_cols.
template <int W, typename Done>
__device__ __forceinline__ void chol_cols_reg(double (&a)[16], double (*colbuf)[64], int* sfail, int lane, Done done) {
  constexpr int NC = 16, C0 = NC * W;
  // (A) a rolled loop is fine here: every register index is static
#pragma unroll 1
  for (int j = 0; j < C0; ++j) {
    __syncthreads();
    const double* col = colbuf[j & 1];
    const double lij = col[lane];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) a[cc] = __builtin_fma(-lij, readlane_d(lij, C0 + cc), a[cc]);
  }
  // (B)
#pragma unroll
  for (int jl = 0; jl < NC; ++jl) {
    const int j = C0 + jl;
    const double piv = readlane_d(a[jl], j);
    if (!(piv > 0.0) && lane == 0 && *sfail == 0) *sfail = j + 1;  // also catches NaN; first failing j wins
    const double rd = rsqrt_newton(piv);
    double d = piv * rd;
    d = __builtin_fma(0.5 * rd, __builtin_fma(-d, d, piv), d);  // one correction: d = sqrt(piv)
    const double lij = (lane > j) ? a[jl] * rd : 0.0;
    a[jl] = (lane == j) ? d : lij;
    colbuf[j & 1][lane] = lij;
    __syncthreads();
    done(j, a[jl], d);
    // my own columns right of j: L(c, j) straight from the owner's registers (no LDS round trip on the
    // critical path of the next pivot)
#pragma unroll
    for (int cc = jl + 1; cc < NC; ++cc) a[cc] = __builtin_fma(-lij, readlane_d(lij, C0 + cc), a[cc]);
  }
  // (C)
#pragma unroll 1
  for (int j = C0 + NC; j < 64; ++j) __syncthreads();
}

// wave W: columns W, W+4, ..., W+60 of X = L^-1
template <int W>
__device__ __forceinline__ void inv_cols(const double (*s)[64], double* __restrict__ inv, int lane) {
  constexpr int NC = 16;
  const double rd_own = 1.0 / s[lane][lane];
  double x[NC];
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) x[cc] = (lane == W + 4 * cc) ? 1.0 : 0.0;
#pragma unroll
  for (int p = W; p < 64; ++p) {
    const double rdp = readlane_d(rd_own, p);
    const double f = (lane == p) ? rdp : 1.0;
    const double lm = (lane == p) ? 0.0 : s[p][lane];  // L(lane, p); zero above the diagonal
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      if (W + 4 * cc <= p) {
        x[cc] *= f;                                // lane p: X(p,c) = acc_p / L_pp
        const double xpc = readlane_d(x[cc], p);
        x[cc] = __builtin_fma(-lm, xpc, x[cc]);    // rows below: acc_i -= L(i,p) X(p,c)
      }
    }
  }
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) inv[lane + (W + 4 * cc) * 64] = x[cc];
}

// the tile kernel below as a device function over caller-provided LDS (s[64][64], colbuf[2][64], sfail), for
// chain_update_potrf_kernel; the stand-alone kernel keeps its own text so that its measured code does not move
__device__ __forceinline__ void potrf_tile_body(double* __restrict__ Ajj, int64_t lda, double* __restrict__ inv,
                                                int* __restrict__ info, int col0, double (*s)[64], double (*colbuf)[64],
                                                int* sfail) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (*info != 0) return;  // an earlier tile already failed: the whole pipeline is void
  if (tid == 0) *sfail = 0;
  // coalesced tile load through LDS.  The strict upper triangle holds don't-care values: they are
  // updated like everything else but never read as a pivot or broadcast, so they cannot contaminate the factor.
  for (int idx = tid; idx < 4096; idx += 256) {
    const int r = idx & 63, c = idx >> 6;
    s[c][r] = Ajj[r + (int64_t)c * lda];
  }
  __syncthreads();
  switch (wave) {
    case 0: chol_cols<0>(s, colbuf, sfail, lane); break;
    case 1: chol_cols<1>(s, colbuf, sfail, lane); break;
    case 2: chol_cols<2>(s, colbuf, sfail, lane); break;
    default: chol_cols<3>(s, colbuf, sfail, lane); break;
  }
  __syncthreads();
  if (tid == 0 && *sfail != 0) atomicCAS(info, 0, col0 + *sfail);
  for (int idx = tid; idx < 4096; idx += 256) {
    const int r = idx & 63, c = idx >> 6;
    if (r >= c) Ajj[r + (int64_t)c * lda] = s[c][r];
  }
  switch (wave) {
    case 0: inv_cols<0>(s, inv, lane); break;
    case 1: inv_cols<1>(s, inv, lane); break;
    case 2: inv_cols<2>(s, inv, lane); break;
    default: inv_cols<3>(s, inv, lane); break;
  }
}

__global__ __launch_bounds__(256, 2) void potrf_tile_kernel(double* __restrict__ Ajj, int64_t lda,
                                                            double* __restrict__ inv,
                                                            int* __restrict__ info, int col0) {
  __shared__ double s[64][64];      // s[c][r] = A(r, c) on entry, L(r, c) (zero above the diagonal) after
  __shared__ double colbuf[2][64];  // the scaled pivot column of the current step
  __shared__ int sfail;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (*info != 0) return;  // an earlier tile already failed: the whole pipeline is void
  if (tid == 0) sfail = 0;
  // coalesced tile load through LDS.  The strict upper triangle holds don't-care values: they are
  // updated like everything else but never read as a pivot or broadcast, so they cannot contaminate the factor.
  for (int idx = tid; idx < 4096; idx += 256) {
    const int r = idx & 63, c = idx >> 6;
    s[c][r] = Ajj[r + (int64_t)c * lda];
  }
  __syncthreads();
  switch (wave) {
    case 0: chol_cols<0>(s, colbuf, &sfail, lane); break;
    case 1: chol_cols<1>(s, colbuf, &sfail, lane); break;
    case 2: chol_cols<2>(s, colbuf, &sfail, lane); break;
    default: chol_cols<3>(s, colbuf, &sfail, lane); break;
  }
  __syncthreads();
  if (tid == 0 && sfail != 0) atomicCAS(info, 0, col0 + sfail);
  for (int idx = tid; idx < 4096; idx += 256) {
    const int r = idx & 63, c = idx >> 6;
    if (r >= c) Ajj[r + (int64_t)c * lda] = s[c][r];
  }
  switch (wave) {
    case 0: inv_cols<0>(s, inv, lane); break;
    case 1: inv_cols<1>(s, inv, lane); break;
    case 2: inv_cols<2>(s, inv, lane); break;
    default: inv_cols<3>(s, inv, lane); break;
  }
}

#ifdef BGP_EXPERIMENTAL
// ==== EXPERIMENTAL (compiled only with -DBGP_EXPERIMENTAL, battgp_amd/build.py --experimental): the fused chain launch and
// the slim chain kernels below have never been timed on an MI355X; they are not part of the default library ====
// One launch for {rank-64 update of the rest of the panel block by column step j} + {tile Cholesky of step j + 1}:
// the workgroup that updates the next diagonal tile (tile (0, 0) of the launch) goes on to factor and invert it while
// the other workgroups finish their tiles.  The serial chain per 64 columns drops from three dependent launches
// {potrf, TRSM, update} to two {TRSM, update + potrf}, and the tile Cholesky runs in the shadow of the update instead of
// behind it.  The tile product is the latency-lean form a k = 64 step allows: BOTH 64 x 64 operand blocks go to LDS in
// one stage (sixteen 16-byte loads in flight per thread, one barrier), then 16 x 4 MFMAs per wave in the same order over
// k as gemm_nt_kernel<64, 64, 0>.  Same operations on the same values as the separate kernels (the tile goes through
// global memory either way): bit-identical.   C[m, n] -= A[m, 64] B[n, 64]^T, lower: tiles with ti >= tj only.
__global__ __launch_bounds__(256) void chain_update_potrf_kernel(double* C, int64_t ldc, const double* A, int64_t lda,
                                                                 const double* B, int64_t ldb, int64_t m, int64_t n,
                                                                 int lower, int* __restrict__ info,
                                                                 double* __restrict__ inv_next, int col0_next) {
  constexpr int T = 64, LDS_S = T + 16;
  __shared__ __attribute__((aligned(16))) double raw[2 * 64 * LDS_S];  // 81 920 B: sA[64][80], sB[64][80]; then the tile
  __shared__ int sfail;
  double (*sA)[LDS_S] = reinterpret_cast<double (*)[LDS_S]>(raw);
  double (*sB)[LDS_S] = reinterpret_cast<double (*)[LDS_S]>(raw + 64 * LDS_S);
  const int ti = (int)blockIdx.x, tj = (int)blockIdx.y;
  if ((lower & 1) && ti < tj) return;
  if (*info != 0) return;
  const int64_t i0 = (int64_t)ti * T, j0 = (int64_t)tj * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave & 1, wj = wave >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  // staging: thread -> rows 2 (tid % 32).., k columns tid / 32 + 8 q (q = 0..7), for A and for B
  const int rs = (tid & 31) * 2, cs = tid >> 5;
  const double* gA = A + ((i0 + rs) < m ? (i0 + rs) : 0) + (int64_t)cs * lda;  // out-of-range rows: any valid address
  const double* gB = B + ((j0 + rs) < n ? (j0 + rs) : 0) + (int64_t)cs * ldb;  // (they only reach unwritten outputs)
  double2 ra[8], rb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    ra[q] = *reinterpret_cast<const double2*>(gA + (int64_t)(8 * q) * lda);
    rb[q] = *reinterpret_cast<const double2*>(gB + (int64_t)(8 * q) * ldb);
  }
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int64_t i = i0 + wi * 32 + b * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t j = j0 + wj * 32 + a * 16 + l4 + 4 * r;
        acc[a][b][r] = (i < m && j < n) ? C[i + j * ldc] : 0.0;
      }
    }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    *reinterpret_cast<double2*>(&sA[cs + 8 * q][rs]) = ra[q];
    *reinterpret_cast<double2*>(&sB[cs + 8 * q][rs]) = rb[q];
  }
  __syncthreads();
  const int ibase = wi * 32 + l15, jbase = wj * 32 + l15;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int pp = kk * 4 + l4;
    double fa[2], fb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) fa[a] = sB[pp][jbase + a * 16];
#pragma unroll
    for (int b = 0; b < 2; ++b) fb[b] = sA[pp][ibase + b * 16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 1);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int64_t i = i0 + wi * 32 + b * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t j = j0 + wj * 32 + a * 16 + l4 + 4 * r;
        if (i < m && j < n) C[i + j * ldc] = acc[a][b][r];
      }
    }
  if (ti != 0 || tj != 0 || inv_next == nullptr) return;  // (uniform over the workgroup)
  __syncthreads();  // this workgroup's stores of the tile are visible to all its threads; the operand stages are free
  potrf_tile_body(C, ldc, inv_next, info, col0_next, reinterpret_cast<double (*)[64]>(raw), reinterpret_cast<double (*)[64]>(raw + 4096),
                  &sfail);
}

// ---- "slim" chain kernels: made to fit NEXT TO two resident trailing-update workgroups ----------------
// gemm_nt_kernel<128,128,2> holds 224 VGPRs per wave and 73 728 B of LDS per workgroup, two workgroups per CU: what a
// CU has left while `rest(k)` runs is 64 VGPRs per SIMD lane and <= 16 KB of LDS (tools/kernel_resources.py).  The
// chain kernels above (78-117 VGPRs, 33-41 KB) cannot be placed until a trailing-update workgroup retires - a whole
// 128 x 128 x NB tile, 110-250 us - and that wait, not their arithmetic, is what the panel stream spends its time on
// underneath a saturating update.  The variants below do the same arithmetic IN THE SAME ORDER (bit-identical results)
// inside <= 64 VGPRs and <= 12 KB of LDS, so the dispatcher can place them at once beside the running update.
// Stand-alone they are slower (no LDS tile for the Cholesky, 8-deep single-buffered GEMM stages), so the driver only
// uses them for the diagonal-block chain of a panel that is factored underneath a trailing update.

// pins a wave-uniform pointer in scalar registers (the intrinsic also keeps the compiler from folding the lane offset
// into it) and marks it as a global-memory address: accesses become  global_load/store v, v_lane_offset, s[base]  -
// ONE 32-bit offset VGPR for all columns instead of a 64-bit address per column
typedef BGP_GLOBAL_AS double gmem_double;
__device__ __forceinline__ gmem_double* scalar_ptr(const double* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gmem_double*)(((uint64_t)hi << 32) | lo);
}

// wave W of the slim tile kernel: columns 16W..16W+15 of the tile in registers (lane = row).  A column leaves the
// registers the moment it is final: to global memory (the factor), and - columns 16..63, strictly below the diagonal,
// packed: 1128 doubles - to LDS for the inverse; columns 0..15 are re-read from global memory by the inverse
// (written by this workgroup, read behind a barrier: workgroup-scope coherent through the CU's own L1 / L2).
constexpr int SLIM_LQ = 1128;  // sum_{c = 16}^{63} (63 - c)
__device__ __forceinline__ constexpr int slim_lq_off(int c) {  // first entry (row c + 1) of column c >= 16
  return (c - 16) * 47 - ((c - 16) * (c - 17)) / 2;            // sum_{t = 16}^{c - 1} (63 - t)
}

template <int W>
__device__ __forceinline__ void potrf_slim_body(double* __restrict__ Ajj, int64_t lda, double* __restrict__ inv,
                                                double (*colbuf)[64], double* lq, double* srd, int* sfail,
                                                int* __restrict__ info, int col0, int lane) {
  constexpr int NC = 16, C0 = NC * W;
  const unsigned ulane = (unsigned)lane;
  {
    double a[NC];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) a[cc] = scalar_ptr(Ajj + (int64_t)(C0 + cc) * lda)[ulane];  // 512 B per column, coalesced
    chol_cols_reg<W>(a, colbuf, sfail, lane, [&](int j, double col, double d) {
      gmem_double* colp = scalar_ptr(Ajj + (int64_t)j * lda);  // (formed by the whole wave, outside the lane condition)
      if (lane >= j) colp[ulane] = col;
      if (j >= 16 && lane > j) lq[slim_lq_off(j) + lane - j - 1] = col;
      if (lane == j) srd[j] = 1.0 / d;
    });
  }
  __syncthreads();
  if (W == 0 && lane == 0 && *sfail != 0) atomicCAS(info, 0, col0 + *sfail);
  // X = L^-1: the forward substitution of inv_cols (column c = W + 4 t of X belongs to wave W), every step with the
  // same operations in the same order; a column that is not active yet (c > p) still holds e_c, for which a step is
  // exactly the identity (0 * f = 0, x - lm * 0 = x), so the rolled loops need no per-column condition
  double x[NC];
  auto step = [&](int p, double lcol, auto ncols) {
    constexpr int NCOL = decltype(ncols)::value;
    const double f = (lane == p) ? srd[p] : 1.0;
    const double lm = (lane > p) ? lcol : 0.0;  // L(lane, p); zero on and above the diagonal
#pragma unroll
    for (int cc = 0; cc < NCOL; ++cc) {
      x[cc] *= f;
      const double xpc = readlane_d(x[cc], p);
      x[cc] = __builtin_fma(-lm, xpc, x[cc]);
    }
  };
  // steps 0..15 only concern the columns c < 16 of X (four per wave) and take their columns of L from global memory,
  // four loads in flight ahead of the four steps being applied (rolled loops: the unrolled form keeps the operands of
  // many steps alive at once and spills)
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) x[cc] = (lane == W + 4 * cc) ? 1.0 : 0.0;
  {
    double g[4], gn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) g[t] = scalar_ptr(Ajj + (int64_t)t * lda)[ulane];
#pragma unroll 1
    for (int grp = 0; grp < 4; ++grp) {
      const int nxt = grp < 3 ? 4 * grp + 4 : 0;  // (the last group re-reads columns 0..3: harmless, keeps the loop uniform)
#pragma unroll
      for (int t = 0; t < 4; ++t) gn[t] = scalar_ptr(Ajj + (int64_t)(nxt + t) * lda)[ulane];
#pragma unroll
      for (int t = 0; t < 4; ++t) step(4 * grp + t, g[t], std::integral_constant<int, 4>{});
#pragma unroll
      for (int t = 0; t < 4; ++t) g[t] = gn[t];
    }
  }
  // steps 16..63: all sixteen columns, L from LDS
#pragma unroll
  for (int cc = 4; cc < NC; ++cc) x[cc] = (lane == W + 4 * cc) ? 1.0 : 0.0;
#pragma unroll 1
  for (int p = 16; p < 64; ++p) {
    const int off = (p - 16) * 47 - ((p - 16) * (p - 17)) / 2;
    step(p, lane > p ? lq[off + lane - p - 1] : 0.0, std::integral_constant<int, NC>{});
  }
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) scalar_ptr(inv + (W + 4 * cc) * 64)[ulane] = x[cc];
}

constexpr int SLIM_POTRF_LDS = 2 * 64 + SLIM_LQ + 64;  // doubles: colbuf[2][64], lq, srd
// the whole slim tile kernel over caller-provided LDS (SLIM_POTRF_LDS doubles + an int)
__device__ __forceinline__ void potrf_slim_tile(double* __restrict__ Ajj, int64_t lda, double* __restrict__ inv,
                                                int* __restrict__ info, int col0, double* lds, int* sfail) {
  double (*colbuf)[64] = reinterpret_cast<double (*)[64]>(lds);  // the scaled pivot column of the current step
  double* lq = lds + 2 * 64;        // columns 16..63 of L, rows below the diagonal, packed: the inverse reads them
  double* srd = lq + SLIM_LQ;       // 1 / L_pp
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (*info != 0) return;
  if (tid == 0) *sfail = 0;
  __syncthreads();
  switch (wave) {
    case 0: potrf_slim_body<0>(Ajj, lda, inv, colbuf, lq, srd, sfail, info, col0, lane); break;
    case 1: potrf_slim_body<1>(Ajj, lda, inv, colbuf, lq, srd, sfail, info, col0, lane); break;
    case 2: potrf_slim_body<2>(Ajj, lda, inv, colbuf, lq, srd, sfail, info, col0, lane); break;
    default: potrf_slim_body<3>(Ajj, lda, inv, colbuf, lq, srd, sfail, info, col0, lane); break;
  }
}

__global__ __launch_bounds__(256) BGP_WAVES_PER_EU(8)
void potrf_tile_slim_kernel(double* __restrict__ Ajj, int64_t lda, double* __restrict__ inv, int* __restrict__ info,
                            int col0) {
  __shared__ double lds[SLIM_POTRF_LDS];
  __shared__ int sfail;
  potrf_slim_tile(Ajj, lda, inv, info, col0, lds, &sfail);
}

// The k = 64 products of the chain on 64 x 64 tiles:  MODE 0  C -= A B^T (`lower`: tiles with ti >= tj only),
// MODE 1  C = A B^T (C may alias A: TRSM by the inverted tile).  Same fragment layout and the same order of MFMAs
// over k as gemm_nt_kernel<64, 64, MODE> (bit-identical), but 8-deep single-buffered stages: 10 KB of LDS.
constexpr int SLIM_GEMM_LDS = 2 * 8 * (64 + 16);  // doubles: sA[8][80], sB[8][80]
// the tile product over caller-provided LDS (SLIM_GEMM_LDS doubles); false when the block has no tile / the pipeline
// is aborted
template <int MODE>
__device__ __forceinline__ bool chain_gemm_slim_tile(double* lds, double* C, int64_t ldc, const double* A, int64_t lda,
                                                     const double* B, int64_t ldb, int64_t m, int64_t n, int lower,
                                                     const int* __restrict__ abort_flag) {
  constexpr int T = 64, SBK = 8, LDS_S = T + 16, KTOT = 64;
  double (*sA)[LDS_S] = reinterpret_cast<double (*)[LDS_S]>(lds);
  double (*sB)[LDS_S] = reinterpret_cast<double (*)[LDS_S]>(lds + SBK * LDS_S);
  const int ti = (int)blockIdx.x, tj = (int)blockIdx.y;  // 2-D grid: tile indices arrive in scalar registers
  if ((lower & 1) && ti < tj) return false;
  if (abort_flag != nullptr && *abort_flag != 0) return false;
  const int64_t i0 = (int64_t)ti * T, j0 = (int64_t)tj * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 1, wj = wave >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  // staging: one double2 of A and one of B per thread and stage (rows 2 (tid % 32).., k column tid / 32)
  const int rs = (tid & 31) * 2, cs = tid >> 5;
  const double* gA = A + ((i0 + rs) < m ? (i0 + rs) : 0) + (int64_t)cs * lda;  // out-of-range rows: any valid address
  const double* gB = B + ((j0 + rs) < n ? (j0 + rs) : 0) + (int64_t)cs * ldb;  // (they only reach unwritten outputs)
  constexpr int NEG = (MODE == 0) ? 1 : 0;
  // C(i, j) of lane (l15, l4), register r of MFMA tile (a, b):  i = i0 + wi 32 + b 16 + l15,  j = j0 + wj 32 + a 16 + l4 + 4 r
  // = a wave-uniform base (scalar registers) + ONE 32-bit lane offset: sixteen 64-bit addresses would not fit the budget
  const unsigned coff = (unsigned)((wi * 32 + l15) + (int64_t)(wj * 32 + l4) * ldc);
  const bool i_in[2] = {i0 + wi * 32 + l15 < m, i0 + wi * 32 + 16 + l15 < m};
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t j = j0 + wj * 32 + a * 16 + l4 + 4 * r;
        gmem_double* cp = scalar_ptr(C + (i0 + b * 16) + (j0 + a * 16 + 4 * r) * ldc);
        acc[a][b][r] = (MODE == 0 && i_in[b] && j < n) ? cp[coff] : 0.0;
      }
  double2 ra = *reinterpret_cast<const double2*>(gA), rb = *reinterpret_cast<const double2*>(gB);
  const int ibase = wi * 32 + l15, jbase = wj * 32 + l15;
#pragma unroll 1
  for (int kt = 0; kt < KTOT / SBK; ++kt) {
    __syncthreads();  // the fragments of the previous stage have been read
    *reinterpret_cast<double2*>(&sA[cs][rs]) = ra;
    *reinterpret_cast<double2*>(&sB[cs][rs]) = rb;
    if (kt + 1 < KTOT / SBK) {
      ra = *reinterpret_cast<const double2*>(gA + (int64_t)(kt + 1) * SBK * lda);
      rb = *reinterpret_cast<const double2*>(gB + (int64_t)(kt + 1) * SBK * ldb);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < SBK / 4; ++kk) {
      const int pp = kk * 4 + l4;
      double fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = sB[pp][jbase + a * 16];
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = sA[pp][ibase + b * 16];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, NEG);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t j = j0 + wj * 32 + a * 16 + l4 + 4 * r;
        gmem_double* cp = scalar_ptr(C + (i0 + b * 16) + (j0 + a * 16 + 4 * r) * ldc);
        if (i_in[b] && j < n) cp[coff] = acc[a][b][r];
      }
  return true;
}

template <int MODE>
__global__ __launch_bounds__(256) BGP_WAVES_PER_EU(8)
void chain_gemm_slim_kernel(double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb,
                            int64_t m, int64_t n, int lower, const int* __restrict__ abort_flag) {
  __shared__ __attribute__((aligned(16))) double lds[SLIM_GEMM_LDS];
  chain_gemm_slim_tile<MODE>(lds, C, ldc, A, lda, B, ldb, m, n, lower, abort_flag);
}

// slim rank-64 update + slim tile Cholesky of the launch's tile (0, 0) (chain_update_potrf_kernel's counterpart for a
// chain that runs underneath a trailing update: still <= 64 VGPRs and ~10.6 KB of LDS, and one slot acquisition less
// per 64 columns)
__global__ __launch_bounds__(256) BGP_WAVES_PER_EU(8)
void chain_update_potrf_slim_kernel(double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb,
                                    int64_t m, int64_t n, int lower, int* __restrict__ info, double* __restrict__ inv_next,
                                    int col0_next) {
  constexpr int NLDS = SLIM_GEMM_LDS > SLIM_POTRF_LDS ? SLIM_GEMM_LDS : SLIM_POTRF_LDS;
  __shared__ __attribute__((aligned(16))) double lds[NLDS];
  __shared__ int sfail;
  const bool did = chain_gemm_slim_tile<0>(lds, C, ldc, A, lda, B, ldb, m, n, lower, info);
  if (!did || blockIdx.x != 0 || blockIdx.y != 0) return;  // (uniform over the workgroup)
  __syncthreads();  // the tile's stores are visible to the whole workgroup; the operand stages are free
  potrf_slim_tile(C, ldc, inv_next, info, col0_next, lds, &sfail);
}

#endif  // BGP_EXPERIMENTAL

// ---- explicit inverses of ALL diagonal panel blocks of a factor, in one launch ---------------------
// Linv_all[p] (ld = NB) = inv(L_pp) for every outer panel p, from the factor's blocks and the stored 64 x 64
// tile inverses by block forward substitution:  X_jj = inv_j,  X_ij = -inv_i sum_{q=j}^{i-1} L_iq X_qj.
// One workgroup per (panel, block column j): the block columns are independent, so the whole launch is as
// long as ONE chain of <= NB/64 - 1 steps.  With these inverses a later query block goes through the factor
// with 3 launches per outer panel (solve by inverse, copy, deep update) instead of 2 per 64 columns.
// Plain fp64 FMA on 64 x 64 blocks staged in LDS (4 x 4 outputs per thread): latency, not throughput,
// is what matters here.
__global__ __launch_bounds__(256) void trinv_panels_kernel(SlabView L, const double* __restrict__ inv_tiles,
                                                           double* __restrict__ Linv_all, int64_t n, int NB) {
  __shared__ double sA[64][65];  // [r][k]
  __shared__ double sB[64][65];  // [k][c]
  const int nblk_max = NB / 64;
  const int p = (int)blockIdx.x / nblk_max, jb = (int)blockIdx.x % nblk_max;
  const int64_t K0 = (int64_t)p * NB;
  const int64_t nbk = (n - K0 < NB) ? (n - K0) : NB;
  const int nblk = (int)(nbk / 64);
  if (jb >= nblk) return;
  double* X = Linv_all + (int64_t)p * NB * NB;
  const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
  const double* tiles = inv_tiles + (K0 / 64) * 4096;
  for (int idx = tid; idx < 4096; idx += 256) {
    const int r = idx & 63, c = idx >> 6;
    X[(64 * jb + r) + (int64_t)(64 * jb + c) * NB] = tiles[(int64_t)jb * 4096 + idx];
  }
  for (int i = jb + 1; i < nblk; ++i) {
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int q = jb; q < i; ++q) {
      __threadfence_block();
      __syncthreads();  // previous products done with sA / sB; X blocks written by this workgroup visible
      for (int idx = tid; idx < 4096; idx += 256) {
        const int r = idx & 63, c = idx >> 6;
        sA[r][c] = *L.at(K0 + 64 * i + r, K0 + 64 * q + c);        // L_iq (r, k = c)
        sB[r][c] = X[(64 * q + r) + (int64_t)(64 * jb + c) * NB];  // X_qj (k = r, c)
      }
      __syncthreads();
#pragma unroll 4
      for (int k = 0; k < 64; ++k) {
        double av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) av[a] = sA[4 * tr + a][k];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = sB[k][4 * tc + b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_fma(av[a], bv[b], acc[a][b]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) sB[4 * tr + a][4 * tc + b] = acc[a][b];  // S as [k][c]
    for (int idx = tid; idx < 4096; idx += 256) sA[idx & 63][idx >> 6] = tiles[(int64_t)i * 4096 + idx];  // inv_i (r, k)
    __syncthreads();
    double out[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) out[a][b] = 0.0;
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = sA[4 * tr + a][k];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = sB[k][4 * tc + b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) out[a][b] = __builtin_fma(-av[a], bv[b], out[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) X[(64 * i + 4 * tr + a) + (int64_t)(64 * jb + 4 * tc + b) * NB] = out[a][b];
  }
}

// out[0] = sum_i log A_ii, out[1] = sum_i z_i^2   (z strided by ldz).  One workgroup.
__global__ __launch_bounds__(1024) void fit_scalars_kernel(SlabView A, const double* __restrict__ z, int64_t ldz,
                                                           int64_t n, double* __restrict__ out) {
  __shared__ double r0[1024], r1[1024];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    a += log(*A.at(i, i));
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

// ---- triangular solves with ONE right-hand side (z = L^-1 y, alpha = L^-T z) ------------------
// Two-level like the factorisation: per outer panel one single-workgroup kernel solves the
// nbk x nbk diagonal block (64-wide steps using the stored tile inverses) and one streaming
// GEMV kernel applies the panel to the rest of the vector.  HBM-bound: L is read once per solve.
constexpr int TRSV_MAX_NB = 2048;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;
}

// part[chunk * nbk + c] = sum_{r in chunk} P[r, c] x[r];  grid = (row chunks of 1024, nbk / 64)
constexpr int GT_ROWS = 1024;
__global__ __launch_bounds__(256) void gemv_t_partial_kernel(const double* __restrict__ P, int64_t lda,
                                                             const double* __restrict__ x, int64_t rows,
                                                             int nbk, double* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * GT_ROWS;
  double xv[GT_ROWS / 64];
#pragma unroll
  for (int t = 0; t < GT_ROWS / 64; ++t) {
    const int64_t r = r0 + lane + 64 * t;
    xv[t] = (r < rows) ? x[r] : 0.0;
  }
  for (int cc = 0; cc < 16; ++cc) {
    const int c = blockIdx.y * 64 + wave * 16 + cc;
    const double* col = P + (int64_t)c * lda;
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < GT_ROWS / 64; ++t) {
      const int64_t r = r0 + lane + 64 * t;
      const double v = (r < rows) ? col[r] : 0.0;
      acc = __builtin_fma(v, xv[t], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) part[(int64_t)blockIdx.x * nbk + c] = acc;
  }
}

// alpha_K <- L_KK^-T (z_K - sum_chunks part).   One workgroup of 512 threads.
__global__ __launch_bounds__(512) void trsv_block_bwd_kernel(const double* __restrict__ Lkk, int64_t lda,
                                                             const double* __restrict__ invK,
                                                             const double* __restrict__ z,
                                                             const double* __restrict__ part, int nchunks,
                                                             int nbk, double* __restrict__ alpha) {
  __shared__ double sw[TRSV_MAX_NB];
  __shared__ double sv[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < nbk; c += 512) {
    double v = z[c];
    for (int q = 0; q < nchunks; ++q) v -= part[(int64_t)q * nbk + c];
    sw[c] = v;
  }
  __syncthreads();
  for (int j = nbk - 64; j >= 0; j -= 64) {
    // v_c = w_{j+c} - sum_{i >= j+64} L(i, j+c) alpha_i   (wave q: columns 8q..8q+7, lanes over rows)
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int c = wave * 8 + cc;
      const double* col = Lkk + (int64_t)(j + c) * lda;
      double acc = 0.0;
      for (int i = j + 64 + lane; i < nbk; i += 64) acc = __builtin_fma(col[i], sw[i], acc);
      acc = wave_sum(acc);
      if (lane == 0) sv[c] = sw[j + c] - acc;
    }
    __syncthreads();
    // alpha_{j+q} = sum_{p >= q} inv(p, q) v_p   (inv is zero above the diagonal)
    const double* inv_j = invK + (int64_t)(j / 64) * 4096;
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
      const int q = wave * 8 + qq;
      double acc = wave_sum(inv_j[lane + q * 64] * sv[lane]);
      if (lane == 0) sw[j + q] = acc;
    }
    __syncthreads();
  }
  for (int c = tid; c < nbk; c += 512) alpha[c] = sw[c];
}

// alpha_K = inv(L_KK)^T (z_K - sum_chunks part): the diagonal block's share of the backward solve as ONE
// GEMV with the explicit panel inverse (one wave per column, coalesced down the column) instead of a
// single-workgroup substitution over the 64 x 64 tiles (0.5 ms per 1024-wide panel).
__global__ __launch_bounds__(256) void linvT_gemv_kernel(const double* __restrict__ Linv, int64_t ldl,
                                                         const double* __restrict__ z, const double* __restrict__ part,
                                                         int nchunks, int nbk, double* __restrict__ alpha) {
  __shared__ double sw[TRSV_MAX_NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < nbk; c += 256) {
    double v = z[c];
    for (int q = 0; q < nchunks; ++q) v -= part[(int64_t)q * nbk + c];
    sw[c] = v;
  }
  __syncthreads();
  const int c = (int)blockIdx.x * 4 + wave;  // alpha_c = sum_{r >= c} Linv(r, c) w_r
  if (c >= nbk) return;
  const double* col = Linv + (int64_t)c * ldl;
  double acc = 0.0;
  for (int r = c - (c & 63) + lane; r < nbk; r += 64) acc = __builtin_fma(r >= c ? col[r] : 0.0, sw[r], acc);
  acc = wave_sum(acc);
  if (lane == 0) alpha[c] = acc;
}

// ---- row-wise reductions over the query block E[M, n] (column-major, rows contiguous) ---------
// part[chunk][m] = sum_{i in chunk} E[m,i] * (vec ? vec[i] : E[m,i])
constexpr int RD_COLS = BGP_RD_COLS;  // 128: four times the workgroups of a 512-column chunk on a latency-bound pass
__global__ __launch_bounds__(256) void rowdot_kernel(const double* __restrict__ E, int64_t lde, int64_t M,
                                                     int64_t n, const double* __restrict__ vec,
                                                     double* __restrict__ part) {
  const int64_t mrow = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t c0 = (int64_t)blockIdx.y * RD_COLS;
  const int64_t c1 = (c0 + RD_COLS < n) ? c0 + RD_COLS : n;
  if (mrow >= M) return;
  // 8 independent partial sums: the loads of 8 columns are in flight together (one dependent chain of
  // 512 global loads made this kernel 90 us at N = 2048)
  double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int64_t i = c0;
  for (; i + 8 <= c1; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double e = E[mrow + (i + u) * lde];
      a8[u] = __builtin_fma(e, vec ? vec[i + u] : e, a8[u]);
    }
  }
  for (; i < c1; ++i) {
    const double e = E[mrow + i * lde];
    a8[0] = __builtin_fma(e, vec ? vec[i] : e, a8[0]);
  }
  part[(int64_t)blockIdx.y * M + mrow] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
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

// dst = scale * op(mask(src)) for a block src[rows, cols] (column-major, lds_):  TRANS: dst[c + r ldd] (the transposed
// block [cols, rows]), else dst[r + c ldd];  tri: entries above the diagonal of src (r < c) are taken as ZERO - a
// triangular block whose upper half holds whatever the factorisation left there.  64 x 64 tiles through LDS: both the
// reads and the writes run down columns (coalesced).  HBM-bound helper of the in-place inverse (bgp_lml_grad): it moves
// a row block of the slab-stored triangle into the [k][row] operand form of gemm_nt_kernel and back.
template <bool TRANS>
__global__ __launch_bounds__(256) void block_copy_kernel(const double* __restrict__ src, int64_t lds_, int64_t rows, int64_t cols,
                                                         double* __restrict__ dst, int64_t ldd, double scale, int tri) {
  __shared__ double t[64][65];
  const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int cc = ty; cc < 64; cc += 4) {
    const int64_t r = r0 + tx, c = c0 + cc;
    double v = 0.0;
    if (r < rows && c < cols && !(tri && r < c)) v = scale * src[r + c * lds_];
    if (TRANS) t[cc][tx] = v;
    else if (r < rows && c < cols) dst[r + c * ldd] = v;
  }
  if (TRANS) {
    __syncthreads();
    for (int rr = ty; rr < 64; rr += 4) {
      const int64_t r = r0 + rr, c = c0 + tx;  // dst row index = c (contiguous), dst column = r
      if (r < rows && c < cols) dst[c + r * ldd] = t[tx][rr];
    }
  }
}

// augmented block below the matrix: row 0 = y^T (zero in the padding), other rows zero
__global__ __launch_bounds__(256) void aug_rows_kernel(const double* __restrict__ y, int64_t n,
                                                       double* __restrict__ Aaug, int64_t lda, int64_t ncols,
                                                       int naug) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = idx / naug;
  const int r = (int)(idx % naug);
  if (j < ncols) Aaug[r + j * lda] = (r == 0 && j < n) ? y[j] : 0.0;
}

__global__ __launch_bounds__(256) void gather_row_kernel(const double* __restrict__ row, int64_t ld, int64_t n,
                                                         double* __restrict__ dst) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n) dst[j] = row[j * ld];
}

__global__ __launch_bounds__(256) void copy_strided_kernel(const double* __restrict__ src, int64_t n,
                                                           double* __restrict__ dst, int64_t ld,
                                                           int64_t npad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < npad) dst[i * ld] = (i < n) ? src[i] : 0.0;
}

// ---- panel factorisation through a workspace copy of the diagonal block -----------------------
// D [2 nbk, nbk] (ld ldd): rows 0..nbk-1 = the diagonal block A_kk, rows nbk..2nbk-1 = identity.
// The identity rides through the 64-wide chain like any rows below the diagonal and comes out as
// L_kk^-T, from which diag_out writes Linv = L_kk^-1 (lower, row index contiguous) for the one deep
// TRSM-by-inverse GEMM of the tall part of the panel.
__global__ __launch_bounds__(256) void diag_in_kernel(const double* __restrict__ Akk, int64_t lda,
                                                      double* __restrict__ D, int64_t ldd, int nbk) {
  const int c = blockIdx.x;
  for (int r = threadIdx.x; r < 2 * nbk; r += 256)
    D[r + (int64_t)c * ldd] = (r < nbk) ? Akk[r + (int64_t)c * lda] : ((r - nbk == c) ? 1.0 : 0.0);
}

// Akk (lower triangle) <- L_kk;  Linv[j + kk ldl] = (L^-T)[kk, j] = D[nbk + kk, j]  (LDS-tiled transpose).
// T = 64 (33 KB of LDS) or, next to a running trailing update, T = 32 (8.4 KB: fits what two of its workgroups leave).
template <int T>
__global__ __launch_bounds__(256) void diag_out_kernel(const double* __restrict__ D, int64_t ldd,
                                                       double* __restrict__ Akk, int64_t lda,
                                                       double* __restrict__ Linv, int64_t ldl, int nbk) {
  __shared__ double t[T][T + 1];
  constexpr int STEP = 256 / T;
  const int bi = blockIdx.x * T, bj = blockIdx.y * T;  // tile (rows bi.., cols bj..) of D / Akk
  const int tx = threadIdx.x % T, ty = threadIdx.x / T;
  if (bi + T > bj) {
    for (int cc = ty; cc < T; cc += STEP) {
      const int r = bi + tx, c = bj + cc;
      if (r >= c) Akk[r + (int64_t)c * lda] = D[r + (int64_t)c * ldd];
    }
  }
  // R tile (rows kk = bi.., cols j = bj..) -> Linv tile (rows j = bj.., cols kk = bi..)
  for (int cc = ty; cc < T; cc += STEP) t[cc][tx] = D[(nbk + bi + tx) + (int64_t)(bj + cc) * ldd];
  __syncthreads();
  for (int cc = ty; cc < T; cc += STEP) Linv[(bj + tx) + (int64_t)(bi + cc) * ldl] = t[tx][cc];
}

// dst[r + c ldd] = src[r + c lds] for r < rows (even), c < ncols: copy-back of the solved panel
__global__ __launch_bounds__(256) void copy_panel_kernel(const double* __restrict__ src, int64_t lds_,
                                                         double* __restrict__ dst, int64_t ldd, int64_t rows) {
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (r >= rows) return;
  const int64_t c = blockIdx.y;
  *reinterpret_cast<double2*>(dst + r + c * ldd) = *reinterpret_cast<const double2*>(src + r + c * lds_);
}

// *flag |= 1 if column 0 of the row-major X[n, D] is not ascending (NaN counts as a descent)
__global__ __launch_bounds__(256) void check_sorted_kernel(const double* __restrict__ x, int64_t n, int D, int* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i + 1 < n && !(x[i * D] <= x[(i + 1) * D])) atomicOr(flag, 1);
}

// failing-minor flag of the sharded factorisation: dst (an 8-byte slot behind a packed panel) <- *info
__global__ void flag_store_kernel(const int* __restrict__ info, int* __restrict__ dst) {
  dst[0] = *info;
  dst[1] = 0;
}

// a rank that receives a panel whose flag is set poisons its own pipeline too
__global__ void flag_merge_kernel(int* __restrict__ info, const int* __restrict__ src) {
  if (*info == 0 && *src != 0) *info = *src;
}

}  // namespace

int launch_check_sorted(bgp_handle* h, hipStream_t st, const double* x, int64_t n, int D, int* flag) {
  BGP_HIP(h, hipMemsetAsync(flag, 0, sizeof(int), st));
  if (n > 1) {
    hipLaunchKernelGGL(check_sorted_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, D, flag);
    BGP_HIP(h, hipGetLastError());
  }
  return 0;
}

int launch_flag_store(bgp_handle* h, hipStream_t st, const int* info, double* slot) {
  hipLaunchKernelGGL(flag_store_kernel, dim3(1), dim3(1), 0, st, info, reinterpret_cast<int*>(slot));
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_flag_merge(bgp_handle* h, hipStream_t st, int* info, const double* slot) {
  hipLaunchKernelGGL(flag_merge_kernel, dim3(1), dim3(1), 0, st, info, reinterpret_cast<const int*>(slot));
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_diag_in(bgp_handle* h, hipStream_t st, const double* Akk, int64_t lda, double* D, int64_t ldd, int nbk) {
  hipLaunchKernelGGL(diag_in_kernel, dim3((unsigned)nbk), dim3(256), 0, st, Akk, lda, D, ldd, nbk);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_diag_out(bgp_handle* h, hipStream_t st, const double* D, int64_t ldd, double* Akk, int64_t lda,
                    double* Linv, int64_t ldl, int nbk, int slim) {
#ifdef BGP_EXPERIMENTAL
  if (slim)
    hipLaunchKernelGGL((diag_out_kernel<32>), dim3((unsigned)(nbk / 32), (unsigned)(nbk / 32)), dim3(256), 0, st, D, ldd,
                       Akk, lda, Linv, ldl, nbk);
  else
#else
  if (slim) return bgp_fail(h, -1, "diag_out: the slim kernels are not in this library (built without BGP_EXPERIMENTAL)");
#endif
    hipLaunchKernelGGL((diag_out_kernel<64>), dim3((unsigned)(nbk / 64), (unsigned)(nbk / 64)), dim3(256), 0, st, D, ldd,
                       Akk, lda, Linv, ldl, nbk);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_copy_panel(bgp_handle* h, hipStream_t st, const double* src, int64_t lds_, double* dst, int64_t ldd,
                      int64_t rows, int ncols) {
  if (rows <= 0 || ncols <= 0) return 0;
  if ((rows & 1) || (lds_ & 1) || (ldd & 1)) return bgp_fail(h, -1, "copy_panel: rows and strides must be even");
  hipLaunchKernelGGL(copy_panel_kernel, dim3((unsigned)((rows / 2 + 255) / 256), (unsigned)ncols), dim3(256), 0, st, src,
                     lds_, dst, ldd, rows);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_gemm_nt(bgp_handle* h, hipStream_t st, int mode, int tn, double* C, int64_t ldc,
                   const double* A, int64_t lda, const double* B, int64_t ldb, int64_t m, int64_t n,
                   int64_t k, int lower, const int* abort_flag, int btri) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  if ((k % BK) != 0) return bgp_fail(h, -1, "gemm_nt: k=%lld not a multiple of %d", (long long)k, BK);
  if ((m & 1) || (n & 1) || (lda & 1) || (ldb & 1))
    return bgp_fail(h, -1, "gemm_nt: m, n, lda, ldb must be even (m=%lld n=%lld lda=%lld ldb=%lld)",
                    (long long)m, (long long)n, (long long)lda, (long long)ldb);
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15))
    return bgp_fail(h, -1, "gemm_nt: operands must be 16-byte aligned");
  // the 64-deep steps of the panel chain on a few thousand rows: 64 x 64 tiles put four times as many CUs
  // on the (latency-bound) launch as long as everything still fits one round of workgroup slots
  const bool t64 = k == 64 && (mode == 0 || mode == 1) && ((m + 63) / 64) * ((n + 63) / 64) <= 1536;
  const int TM = t64 ? 64 : 128;
  if (t64) tn = 64;
  const int nti = (int)((m + TM - 1) / TM), ntj = (int)((n + tn - 1) / tn);
  // up to one super-tile round per XCD of tiles: direct mapping (bit 1 of `lower`), exact grid
  const bool small = (int64_t)nti * ntj <= 512;
  if (small) lower |= 2;
  const int64_t blocks = small ? (int64_t)nti * ntj : gemm_grid_blocks(nti, ntj, lower);
  if (blocks > 0x7fffffffLL) return bgp_fail(h, -1, "gemm_nt: grid too large");
  dim3 grid((unsigned)blocks), block(256);
  if (t64 && mode == 0)
    hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 0>), grid, block, 0, st, C, ldc, A, lda, B, ldb, m, n,
                       (int)k, lower, nti, ntj, abort_flag, btri);
  else if (t64 && mode == 1)
    hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 1>), grid, block, 0, st, C, ldc, A, lda, B, ldb, m, n,
                       (int)k, lower, nti, ntj, abort_flag, btri);
  else if (tn == 128 && mode == 0)
    hipLaunchKernelGGL((gemm_nt_kernel<128, 128, 0>), grid, block, 0, st, C, ldc, A, lda, B, ldb, m, n,
                       (int)k, lower, nti, ntj, abort_flag, btri);
  else if (tn == 128 && mode == 2)
    hipLaunchKernelGGL((gemm_nt_kernel<128, 128, 2>), grid, block, 0, st, C, ldc, A, lda, B, ldb, m, n,
                       (int)k, lower, nti, ntj, abort_flag, btri);
  else if (tn == 64 && mode == 0)
    hipLaunchKernelGGL((gemm_nt_kernel<128, 64, 0>), grid, block, 0, st, C, ldc, A, lda, B, ldb, m, n,
                       (int)k, lower, nti, ntj, abort_flag, btri);
  else if (tn == 64 && mode == 1)
    hipLaunchKernelGGL((gemm_nt_kernel<128, 64, 1>), grid, block, 0, st, C, ldc, A, lda, B, ldb, m, n,
                       (int)k, lower, nti, ntj, abort_flag, btri);
  else
    return bgp_fail(h, -1, "gemm_nt: unsupported variant tn=%d mode=%d", tn, mode);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_potrf_tile(bgp_handle* h, hipStream_t st, double* Ajj, int64_t lda, double* inv, int* info,
                      int col0, int slim) {
#ifdef BGP_EXPERIMENTAL
  if (slim)
    hipLaunchKernelGGL(potrf_tile_slim_kernel, dim3(1), dim3(256), 0, st, Ajj, lda, inv, info, col0);
  else
#else
  if (slim) return bgp_fail(h, -1, "potrf_tile: the slim kernels are not in this library (built without BGP_EXPERIMENTAL)");
#endif
    hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(256), 0, st, Ajj, lda, inv, info, col0);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

#ifdef BGP_EXPERIMENTAL
// rank-64 update of the rest of the panel block + tile Cholesky (and inverse) of its first diagonal tile, one launch
int launch_chain_update_potrf(bgp_handle* h, hipStream_t st, double* C, int64_t ldc, const double* A, int64_t lda,
                              const double* B, int64_t ldb, int64_t m, int64_t n, int lower, int* info, double* inv_next,
                              int col0_next, int slim) {
  if (m <= 0 || n <= 0) return 0;
  if ((m & 1) || (n & 1) || (lda & 1) || (ldb & 1) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15))
    return bgp_fail(h, -1, "chain_update_potrf: m, n, lda, ldb must be even and the operands 16-byte aligned");
  const int64_t nti = (m + 63) / 64, ntj = (n + 63) / 64;
  if (nti > 0x7fffffffLL || ntj > 65535) return bgp_fail(h, -1, "chain_update_potrf: grid too large");
  if (slim)
    hipLaunchKernelGGL(chain_update_potrf_slim_kernel, dim3((unsigned)nti, (unsigned)ntj), dim3(256), 0, st, C, ldc, A, lda, B,
                       ldb, m, n, lower, info, inv_next, col0_next);
  else
    hipLaunchKernelGGL(chain_update_potrf_kernel, dim3((unsigned)nti, (unsigned)ntj), dim3(256), 0, st, C, ldc, A, lda, B, ldb, m,
                       n, lower, info, inv_next, col0_next);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

// the k = 64 products of the panel chain with the slim kernels (<= 64 VGPRs, 10 KB LDS): mode 0  C -= A B^T
// (lower: tiles touching i >= j only), mode 1  C = A B^T (C may alias A)
int launch_chain_gemm_slim(bgp_handle* h, hipStream_t st, int mode, double* C, int64_t ldc, const double* A, int64_t lda,
                           const double* B, int64_t ldb, int64_t m, int64_t n, int lower, const int* abort_flag) {
  if (m <= 0 || n <= 0) return 0;
  if ((m & 1) || (n & 1) || (lda & 1) || (ldb & 1) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15))
    return bgp_fail(h, -1, "chain_gemm_slim: m, n, lda, ldb must be even and the operands 16-byte aligned");
  const int64_t nti = (m + 63) / 64, ntj = (n + 63) / 64;
  if (nti > 0x7fffffffLL || ntj > 65535) return bgp_fail(h, -1, "chain_gemm_slim: grid too large");
  const dim3 grid((unsigned)nti, (unsigned)ntj);
  if (mode == 0)
    hipLaunchKernelGGL((chain_gemm_slim_kernel<0>), grid, dim3(256), 0, st, C, ldc, A, lda, B, ldb, m, n, lower, abort_flag);
  else if (mode == 1)
    hipLaunchKernelGGL((chain_gemm_slim_kernel<1>), grid, dim3(256), 0, st, C, ldc, A, lda, B, ldb, m, n, lower, abort_flag);
  else
    return bgp_fail(h, -1, "chain_gemm_slim: unsupported mode %d", mode);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

#else   // the default library has neither kernel family (bgp_set_options rejects the bits that would lead here)
int launch_chain_update_potrf(bgp_handle* h, hipStream_t, double*, int64_t, const double*, int64_t, const double*, int64_t, int64_t, int64_t,
                              int, int*, double*, int, int) {
  return bgp_fail(h, -1, "chain_update_potrf: not in this library (built without BGP_EXPERIMENTAL)");
}
int launch_chain_gemm_slim(bgp_handle* h, hipStream_t, int, double*, int64_t, const double*, int64_t, const double*, int64_t, int64_t, int64_t,
                           int, const int*) {
  return bgp_fail(h, -1, "chain_gemm_slim: not in this library (built without BGP_EXPERIMENTAL)");
}
#endif  // BGP_EXPERIMENTAL

int launch_trinv_panels(bgp_handle* h, hipStream_t st, const SlabView& L, const double* inv_tiles, double* Linv_all,
                        int64_t n, int NB) {
  const int64_t npanels = (n + NB - 1) / NB;
  BGP_HIP(h, hipMemsetAsync(Linv_all, 0, (size_t)npanels * NB * NB * sizeof(double), st));
  hipLaunchKernelGGL(trinv_panels_kernel, dim3((unsigned)(npanels * (NB / 64))), dim3(256), 0, st, L, inv_tiles, Linv_all,
                     n, NB);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_fit_scalars(bgp_handle* h, hipStream_t st, const SlabView& A, const double* z,
                       int64_t ldz, int64_t n, double* out2) {
  hipLaunchKernelGGL(fit_scalars_kernel, dim3(1), dim3(1024), 0, st, A, z, ldz, n, out2);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_gemv_t_partial(bgp_handle* h, hipStream_t st, const double* P, int64_t lda, const double* x,
                          int64_t rows, int nbk, double* part, int* nchunks_out) {
  const int nch = (int)((rows + GT_ROWS - 1) / GT_ROWS);
  *nchunks_out = nch;
  if (nch == 0) return 0;
  hipLaunchKernelGGL(gemv_t_partial_kernel, dim3((unsigned)nch, (unsigned)(nbk / 64)), dim3(256), 0, st, P, lda,
                     x, rows, nbk, part);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_trsv_block_bwd(bgp_handle* h, hipStream_t st, const double* Lkk, int64_t lda, const double* invK,
                          const double* z, const double* part, int nchunks, int nbk, double* alpha) {
  if (nbk > TRSV_MAX_NB) return bgp_fail(h, -1, "nb_outer=%d exceeds the TRSV limit %d", nbk, TRSV_MAX_NB);
  hipLaunchKernelGGL(trsv_block_bwd_kernel, dim3(1), dim3(512), 0, st, Lkk, lda, invK, z, part, nchunks, nbk,
                     alpha);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_linvT_gemv(bgp_handle* h, hipStream_t st, const double* Linv, int64_t ldl, const double* z, const double* part,
                      int nchunks, int nbk, double* alpha) {
  if (nbk > TRSV_MAX_NB) return bgp_fail(h, -1, "nb_outer=%d exceeds the TRSV limit %d", nbk, TRSV_MAX_NB);
  hipLaunchKernelGGL(linvT_gemv_kernel, dim3((unsigned)((nbk + 3) / 4)), dim3(256), 0, st, Linv, ldl, z, part, nchunks, nbk,
                     alpha);
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

int launch_aug_rows(bgp_handle* h, hipStream_t st, const double* y, int64_t n, double* Aaug, int64_t lda,
                    int64_t ncols, int naug) {
  const int64_t total = ncols * naug;
  hipLaunchKernelGGL(aug_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, n, Aaug, lda, ncols,
                     naug);
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int launch_gather_row(bgp_handle* h, hipStream_t st, const double* row, int64_t ld, int64_t n, double* dst) {
  hipLaunchKernelGGL(gather_row_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, row, ld, n, dst);
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

int launch_block_copy(bgp_handle* h, hipStream_t st, const double* src, int64_t lds_, int64_t rows, int64_t cols, double* dst,
                      int64_t ldd, int trans, double scale, int tri) {
  if (rows <= 0 || cols <= 0) return 0;
  const int64_t gx = (rows + 63) / 64, gy = (cols + 63) / 64;
  if (gx > 0x7fffffffLL || gy > 65535) return bgp_fail(h, -1, "block_copy: grid too large (%lld x %lld tiles)", (long long)gx, (long long)gy);
  const dim3 grid((unsigned)gx, (unsigned)gy);
  if (trans)
    hipLaunchKernelGGL((block_copy_kernel<true>), grid, dim3(256), 0, st, src, lds_, rows, cols, dst, ldd, scale, tri);
  else
    hipLaunchKernelGGL((block_copy_kernel<false>), grid, dim3(256), 0, st, src, lds_, rows, cols, dst, ldd, scale, tri);
  BGP_HIP(h, hipGetLastError());
  return 0;
}
