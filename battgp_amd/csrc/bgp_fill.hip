// Covariance fill kernels (gfx950).
//
// One workgroup = one 512 x 32 tile of the (column-major) covariance matrix.  The 32 column
// points are scaled once and staged in LDS; every lane keeps two row points in registers and
// the whole workgroup walks the 32 columns together, writing one contiguous 4 KiB column
// segment (16 B per lane) per step: the kernel is a pure coalesced HBM write stream with ~30
// fp64 VALU ops per entry hidden behind it.  Replaces the >= 6 N^2 passes + N `torch.minimum`
// launches of the reference (src/gp/wiener_kernel.py:21-22, RBFKernel/ScaleKernel/AddedDiag
// at src/batt_models/cell_gp.py:32-36).
#include <stdlib.h>

#include <type_traits>

#include "bgp_internal.h"

namespace {

// 2^(j/64), j = 0..63, correctly rounded (generated with 60-digit decimal arithmetic)
__device__ const double EXP2_TBL[64] = {
    1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
    1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418,
    1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
    1.189207115002721, 1.202156731452703, 1.215247359980469, 1.22848053610687,
    1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783,
    1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.339667524053303,
    1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
    1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
    1.5422108254079407, 1.559004400237837, 1.5759808451078865, 1.593142151342267,
    1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364,
    1.681792830507429, 1.7001063537185235, 1.718619298122478, 1.7373338352737062,
    1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
    1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
    1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.978456026387951,
};

// exp(x) for x <= 0:  x = k ln2/64 + r,  |r| <= ln2/128,  exp(x) = 2^(k>>6) T[k&63] e^r
// with e^r by a degree-5 Taylor polynomial (truncation r^6/720 <= 3.5e-17) and the scaling by
// v_ldexp_f64 (correct gradual underflow).  k comes out of the rounding itself: adding 1.5 * 2^52 to x * 64/ln2
// rounds to nearest-even in the fp64 adder and leaves k as a two's-complement integer in the low mantissa word -
// no v_rndne / v_cvt pair.  One LDS table read per call; `tbl` may be pre-multiplied by an output scale.
// FINITE = false: NaN stays NaN (it has to poison Sigma, not vanish in a clamp) and -inf gives 0;
// FINITE = true (the fill's interior tiles, whose points were checked when they were staged): one VALU op less.
// BOUNDED: the caller guarantees -1.6e7 < x (the matrix-pipe tiles: |x| <= 256): no clamp at all.
template <bool FINITE, bool BOUNDED = false>
__device__ __forceinline__ double exp_nonpos_t(double x, const double* __restrict__ tbl) {
  const double INV = 92.33248261689366;      // 64 / ln2
  const double L_HI = 0.01083042469326756;  // ln2/64, low 21 bits zero: k * L_HI is exact
  const double L_LO = 2.9815858269852933e-12;
  const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
  const double LIM = -16777216.0;           // below: the result underflows to 0 anyway; keeps k inside 32 bits
  if (!BOUNDED) x = FINITE ? __builtin_fmax(x, LIM) : ((x < LIM) ? LIM : x);
  const double z = __builtin_fma(x, INV, MAGIC);
  const int k = __double2loint(z);
  const double kd = z - MAGIC;
  double r = __builtin_fma(kd, -L_HI, x);
  r = __builtin_fma(kd, -L_LO, r);
  const double t = tbl[k & 63];
  double p = 8.3333333333333332e-03;               // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02);  // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01);  // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(t * p, k >> 6);
}

__device__ __forceinline__ double exp_nonpos(double x, const double* __restrict__ tbl) {
  return exp_nonpos_t<false>(x, tbl);
}

// The interior tiles' exponential: finite x <= 0, a 256-entry table 2^(i/256) (times the output scale) in LDS, so
// |r| <= ln2/512 and a degree-4 polynomial suffices (truncation r^5/120 <= 3.8e-17): one FMA less per entry than
// the 64-entry / degree-5 version above.
template <bool BOUNDED = false>
__device__ __forceinline__ double exp_interior(double x, const double* __restrict__ tbl256) {
  const double INV = 4.0 * 92.33248261689366;       // 256 / ln2
  const double L_HI = 0.25 * 0.01083042469326756;  // ln2/256 (the low bits stay zero under the power-of-two scaling)
  const double L_LO = 0.25 * 2.9815858269852933e-12;
  const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
  if (!BOUNDED) x = __builtin_fmax(x, -4194304.0);  // keeps k = x * 256/ln2 inside 32 bits; the result is 0 there anyway
  const double z = __builtin_fma(x, INV, MAGIC);
  const int k = __double2loint(z);
  const double kd = z - MAGIC;
  double r = __builtin_fma(kd, -L_HI, x);
  r = __builtin_fma(kd, -L_LO, r);
  const double t = tbl256[k & 255];
  double p = 4.1666666666666664e-02;               // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01);  // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(t * p, k >> 8);
}

// sqrt(q), q >= 0, to ~1.5 ulp: v_rsq_f64 seed (2^-26) times q, then ONE residual correction with the seed's own
// half-reciprocal, r += (q - r^2) y / 2: the error is ~1.5 (2^-26)^2.  5 VALU ops (+ the quarter-rate rsq); the
// coupled Newton step that brings it to 0.5 ulp costs 3 more and 2.4 % of the Matern fill rate; the library
// sqrt costs ~2x and made the fill VALU-bound.  Elementwise parity of the kernel matrix stays < 2e-13.
__device__ __forceinline__ double sqrt_nonneg(double q) {
  const double y = __builtin_amdgcn_rsq(q);
  double r = q * y;
  const double d = __builtin_fma(-r, r, q);
  r = __builtin_fma(d, 0.5 * y, r);
  return q > 0.0 ? r : 0.0;
}

template <int KID, int DD>
__device__ __forceinline__ double kfun(const double (&a)[DD], const double (&b)[DD], int D,
                                       double s0, double s1, const double* __restrict__ tbl) {
  if (KID == BGP_KERNEL_BATTGP) {
    // s_w (m^3/3 + |dt| m^2/2) + s_r exp(-sum_d ((a_d-b_d)/(l_d sqrt2))^2),  d = 1..D-1
    double q = 0.0;
#pragma unroll
    for (int d = 1; d < DD; ++d)
      if (d < D) {
        double df = a[d] - b[d];
        q = __builtin_fma(df, df, q);
      }
    double e = exp_nonpos(-q, tbl);
    // min^3/3 + |a - b| min^2/2  =  min^2 (max/2 - min/6): no cancellation (max/2 - min/6 >= min/3)
    double m = __builtin_fmin(a[0], b[0]);
    double mx = __builtin_fmax(a[0], b[0]);
    double u = __builtin_fma(m, -1.0 / 6.0, 0.5 * mx);
    double w = (m * m) * u;
    return __builtin_fma(s0, w, s1 * e);
  } else if (KID == BGP_KERNEL_MATERN32) {
    double q = 0.0;
#pragma unroll
    for (int d = 0; d < DD; ++d)
      if (d < D) {
        double df = a[d] - b[d];
        q = __builtin_fma(df, df, q);
      }
    double r = sqrt_nonneg(q);  // inputs pre-scaled by sqrt(3)/l: r = sqrt(3) * dist
    return s0 * (1.0 + r) * exp_nonpos(-r, tbl);
  } else {  // SCALED_RBF / ARD_RBF: inputs pre-scaled by 1/(l sqrt2)
    double q = 0.0;
#pragma unroll
    for (int d = 0; d < DD; ++d)
      if (d < D) {
        double df = a[d] - b[d];
        q = __builtin_fma(df, df, q);
      }
    return s0 * exp_nonpos(-q, tbl);
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

// Tile = FT_ROWS x FT_COLS = 512 rows x 32 columns.  All 256 threads of the workgroup work on
// the SAME column at a time (2 rows each, 16 B per lane): one 4 KiB contiguous store per step,
// and a workgroup touches only 32 columns - at N = 131 072 a column is 1 MiB, so the old
// 128-column tile walked 64 different 2 MiB pages per workgroup and the fill ran TLB-bound
// (3.9 TB/s instead of 5.1 TB/s at N = 40 000).  Tiles are enumerated column-tile-major so that
// the workgroups resident at any time write the same few columns (same pages, same DRAM rows).
constexpr int FT_ROWS = 512, FT_COLS = 32, FT_RATIO = FT_ROWS / FT_COLS;

// lower trapezoid: column tile tj needs row tiles ti >= tj / FT_RATIO.  Column tiles are grouped
// by g = tj / FT_RATIO (each group: FT_RATIO column tiles x (nti - g) row tiles).
__device__ __forceinline__ bool lower_decode(int64_t t, int nti, int ntj, int& ti, int& tj) {
  // prefix(g) = FT_RATIO * (g nti - g (g-1) / 2)
  const double b = 2.0 * nti + 1.0;
  int64_t g = (int64_t)((b - __builtin_sqrt(b * b - 8.0 * (double)t / FT_RATIO)) * 0.5);
  if (g < 0) g = 0;
  auto prefix = [&](int64_t q) { return (int64_t)FT_RATIO * (q * nti - q * (q - 1) / 2); };
  while (g > 0 && prefix(g) > t) --g;
  while (prefix(g + 1) <= t) ++g;
  if (g >= nti) return false;
  const int64_t r = t - prefix(g);
  const int64_t rows = nti - g;
  tj = (int)(g * FT_RATIO + r / rows);
  ti = (int)(g + r % rows);
  return tj < ntj;
}

__host__ int64_t lower_blocks(int nti) {
  return (int64_t)FT_RATIO * ((int64_t)nti * nti - (int64_t)nti * (nti - 1) / 2);
}

// blocks of a lower trapezoid that is only ntj column tiles wide (a column slab): the enumeration is
// column-group-major, so the needed tiles are a prefix - do not launch the empty remainder
__host__ int64_t lower_blocks(int nti, int ntj) {
  int64_t G = (ntj + FT_RATIO - 1) / FT_RATIO;
  if (G > nti) G = nti;
  return (int64_t)FT_RATIO * (G * nti - G * (G - 1) / 2);
}

// The interior tiles' arithmetic (finite inputs guaranteed; `tbl` is pre-multiplied by the output scale, so the
// exponential arrives scaled).  K0 with SORTED = true: the tile lies strictly below the diagonal of a matrix whose
// time column is ascending, so min(t_i, t_j) = t_j and the integrated-Wiener term
//   s_w (min^3/3 + |t_i - t_j| min^2/2) = (s_w t_j^2 / 2) t_i - s_w t_j^3 / 6 = A_j t_i + B_j
// is ONE fma with two per-column constants staged next to the column point (w[0] = A_j, w[1] = B_j).
// T256: the 256-entry table / degree-4 exponential (exp_interior, one FMA less per entry; NOT yet timed on the GPU -
// the default is the 64-entry / degree-5 form whose rates are on record)
template <int KID, int DD, bool SORTED, bool T256 = false>
__device__ __forceinline__ double kfun_interior(const double (&a)[DD], const double (&b)[DD], const double (&w)[2], int D,
                                                double s0, const double* __restrict__ tbl) {
  if (KID == BGP_KERNEL_BATTGP) {
    double q = 0.0;
#pragma unroll
    for (int d = 1; d < DD; ++d)
      if (d < D) {
        const double df = a[d] - b[d];
        q = __builtin_fma(df, df, q);
      }
    const double e = T256 ? exp_interior(-q, tbl) : exp_nonpos_t<true>(-q, tbl);  // s_r e^-q
    if (SORTED) return __builtin_fma(w[0], a[0], w[1]) + e;
    const double m = __builtin_fmin(a[0], b[0]);
    const double mx = __builtin_fmax(a[0], b[0]);
    const double u = __builtin_fma(m, -1.0 / 6.0, 0.5 * mx);
    return __builtin_fma((s0 * m) * m, u, e);
  } else if (KID == BGP_KERNEL_MATERN32) {
    // the tiny seed keeps rsq finite at coincident points (r = 1e-150 there): no select on the critical path
    double q = 1e-300;
#pragma unroll
    for (int d = 0; d < DD; ++d)
      if (d < D) {
        const double df = a[d] - b[d];
        q = __builtin_fma(df, df, q);
      }
    const double y = __builtin_amdgcn_rsq(q);
    double r = q * y;
    r = __builtin_fma(__builtin_fma(-r, r, q), 0.5 * y, r);
    const double se = T256 ? exp_interior(-r, tbl) : exp_nonpos_t<true>(-r, tbl);  // s e^-r
    return __builtin_fma(r, se, se);
  } else {
    double q = 0.0;
#pragma unroll
    for (int d = 0; d < DD; ++d)
      if (d < D) {
        const double df = a[d] - b[d];
        q = __builtin_fma(df, df, q);
      }
    return T256 ? exp_interior(-q, tbl) : exp_nonpos_t<true>(-q, tbl);
  }
}

__device__ __forceinline__ bool not_finite(double v) { return !(__builtin_fabs(v) <= 1.7976931348623157e308); }

typedef double fill_v4d __attribute__((ext_vector_type(4)));

// Interior tile with the squared distances on the MATRIX pipe (ABL bit 3, BGP_FILL_MFMA=1; written without a GPU at
// hand: NOT timed, off by default).  The fill is bound by its VALU instruction count at the clock the SMU grants it
// (see below); 6 of the ~29 instructions per entry form q_ij = sum_d (a_id - b_jd)^2.  Expanded around a common shift c
// (the mid-range of the tile's 32 column points, so that |a - c|, |b - c| stay small and the cancellation harmless),
//     q_ij = |b'_j|^2 + sum_d (-2 b'_jd) a'_id + sum_d 1 * a'_id^2,        a' = a - c,  b' = b - c,
// is two v_mfma_f64_16x16x4_f64 per 16 x 16 entries - first operand (-2 b'_j | 1), second operand (a'_i | a'_i^2), the
// accumulator starting as |b'_j|^2 - on a pipe the kernel does not otherwise use: 64 cycles per 256 entries against
// the 24 VALU cycles per entry the subtractions and FMAs cost.  A wave owns 128 rows x 32 columns of the tile; lane
// (l15, l4) supplies ONE coordinate (dimension l4) of its points and receives, per MFMA, the entries
// (row(b, l15), column a 16 + l4 + 4 r), r = 0..3; rows are dealt so that the blocks 2c and 2c + 1 of a lane are
// adjacent rows: stores stay 16 B per lane, 256 B contiguous per column.
// Rounding: |error(q)| <~ 4 eps (|a'|^2 + |b'|^2); tiles with |a'|^2 or |b'|^2 above FILL_MFMA_LIMIT take the VALU path
// (return false), so the entry's relative error stays below ~1e-13 (d g / g = dq / (2 (1 + r)) for Matern-3/2, dq for
// the squared-exponential kernels) - inside the 2e-13 elementwise parity bound, an order of magnitude above the VALU path.
constexpr double FILL_MFMA_LIMIT = 64.0;

template <int KID, bool T256>
__device__ __forceinline__ bool fill_interior_mfma(const FillParams& p, const double* __restrict__ x1, int64_t i0, int64_t j0,
                                                   double* __restrict__ out, int64_t ld, const double (*sB)[4],
                                                   const double (*sW)[2], double* __restrict__ sShift,
                                                   double* __restrict__ sNb, const double* __restrict__ sTs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  constexpr int D0 = (KID == BGP_KERNEL_BATTGP) ? 1 : 0;  // K0: the time column is not part of the RBF distance
  if (tid < 4) {  // mid-range of the 32 column points, per dimension (scaled coordinates)
    double lo = sB[0][tid], hi = lo;
    for (int c = 1; c < FT_COLS; ++c) {
      lo = __builtin_fmin(lo, sB[c][tid]);
      hi = __builtin_fmax(hi, sB[c][tid]);
    }
    sShift[tid] = 0.5 * (lo + hi);
  }
  __syncthreads();
  int big = 0;
  if (tid < FT_COLS) {
    double nb = 0.0;
#pragma unroll
    for (int d = D0; d < 4; ++d) {
      const double b = sB[tid][d] - sShift[d];
      nb = __builtin_fma(b, b, nb);
    }
    sNb[tid] = nb;
    big |= !(nb <= FILL_MFMA_LIMIT);
  }
  // this lane's coordinate (dimension l4) of its 8 row points: block b, lane l15 -> row (b >> 1) 32 + 2 l15 + (b & 1)
  const bool used = l4 >= D0;
  const double shift = sShift[l4], scale = p.scale[l4];
  const int64_t irow = i0 + wave * 128 + 2 * l15;
  double ya[8], yq[8], tr[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int64_t i = irow + (b >> 1) * 32 + (b & 1);
    const double v = x1[i * 4 + l4];
    const double a = used ? __builtin_fma(v, scale, -shift) : 0.0;
    ya[b] = a;
    yq[b] = a * a;
    if (KID == BGP_KERNEL_BATTGP) tr[b] = x1[i * 4];  // t_i of the sorted-time Wiener term
  }
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    // |a'|^2 of row (b, l15): the four lanes l15, l15 + 16, + 32, + 48 hold its components.  Lanes 0..15 end up with the
    // full sum, the others with partial sums (<= the full one: no false alarms)
    double n2 = yq[b];
    n2 += __shfl_down(n2, 32);
    n2 += __shfl_down(n2, 16);
    big |= !(n2 <= FILL_MFMA_LIMIT);
  }
  if (__syncthreads_or(big)) return false;

  double xa[2];
  fill_v4d nbv[2];
  double w0[2][4], w1[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    xa[a] = used ? -2.0 * (sB[a * 16 + l15][l4] - shift) : 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = a * 16 + l4 + 4 * r;
      nbv[a][r] = sNb[c];
      if (KID == BGP_KERNEL_BATTGP) {
        w0[a][r] = sW[c][0];
        w1[a][r] = sW[c][1];
      }
    }
  }
  typedef double v2d __attribute__((ext_vector_type(2)));
  // the guard above bounds every argument of the exponential: q <= (|a'| + |b'|)^2 <= 4 FILL_MFMA_LIMIT = 256
  auto entry = [&](double q, int a, int r, int b) -> double {
    if (KID == BGP_KERNEL_BATTGP) {
      const double e = T256 ? exp_interior<true>(-q, sTs) : exp_nonpos_t<true, true>(-q, sTs);  // s_r e^-q
      return __builtin_fma(w0[a][r], tr[b], w1[a][r]) + e;
    } else if (KID == BGP_KERNEL_MATERN32) {
      q = __builtin_fmax(q, 1e-300);  // cancellation may leave a tiny negative number at coincident points
      const double y = __builtin_amdgcn_rsq(q);
      double rr = q * y;
      rr = __builtin_fma(__builtin_fma(-rr, rr, q), 0.5 * y, rr);
      const double se = T256 ? exp_interior<true>(-rr, sTs) : exp_nonpos_t<true, true>(-rr, sTs);
      return __builtin_fma(rr, se, se);
    } else {
      return T256 ? exp_interior<true>(-q, sTs) : exp_nonpos_t<true, true>(-q, sTs);
    }
  };
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      fill_v4d q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[a], ya[2 * c], nbv[a], 0, 0, 0);
      fill_v4d q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[a], ya[2 * c + 1], nbv[a], 0, 0, 0);
      q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, yq[2 * c], q0, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, yq[2 * c + 1], q1, 0, 0, 0);
      double* dst = out + (irow + c * 32) + (j0 + a * 16 + l4) * ld;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v2d vv = {entry(q0[r], a, r, 2 * c), entry(q1[r], a, r, 2 * c + 1)};
        __builtin_nontemporal_store(vv, reinterpret_cast<v2d*>(dst + (int64_t)(4 * r) * ld));
      }
    }
  }
  return true;
}

// ABL != 0: ablation variants for tools/fill_ablate.hip only (1 = no kernel math, 2 = no stores)
// UNR: unroll of the interior column loop.  Interior tiles store non-temporally: the matrix is written once and
// next read by another kernel after > 100 GB of other traffic (+5-6 % on the store stream, tools/fill_gap_probe.py).
//
// What bounds this kernel (tools/fill_gap_probe.py with FILL_PROBE_CLOCK=1): while it runs the SMU holds the shader
// clock at 1.0-1.4 GHz (2.39 GHz idle, >= 2.2 GHz under hipMemset or a pure fp64-FMA load), so its ~30 VALU
// instructions per 8 bytes, not the store stream, set the pace - hence the lean interior arithmetic above.
template <int KID, int DT, int ABL = 0, int UNR = 8>
__global__ __launch_bounds__(256) void fill_kernel(FillParams p, const double* __restrict__ x1,
                                                   int64_t n1, const double* __restrict__ x2,
                                                   int64_t n2, double* __restrict__ out, int64_t ld,
                                                   int lower, int add_diag, int64_t nv1, int64_t nv2,
                                                   int nti, int ntj, int vec_ok) {
#include "bgp_fill_tile.inc"
}

#ifdef BGP_EXPERIMENTAL
// the matrix-pipe variant (fill_interior_mfma): two workgroups per CU by contract, so that the register budget is 256
// and the MFMA results land in ordinary VGPRs (with the default budget of 512 the compiler parks them in AGPRs and pays
// two v_accvgpr_read per entry - the instructions the variant exists to save)
template <int KID, int ABL, int UNR>
__global__ __launch_bounds__(256, 2) void fill_mfma_kernel(FillParams p, const double* __restrict__ x1,
                                                           int64_t n1, const double* __restrict__ x2,
                                                           int64_t n2, double* __restrict__ out, int64_t ld,
                                                           int lower, int add_diag, int64_t nv1, int64_t nv2,
                                                           int nti, int ntj, int vec_ok) {
  static_assert((ABL & 8) != 0, "the matrix-pipe variant");
  constexpr int DT = 4;
#include "bgp_fill_tile.inc"
}
#endif  // BGP_EXPERIMENTAL

template <int KID>
int fill_dispatch(hipStream_t st, const FillParams& p, const double* x1, int64_t n1, const double* x2,
                  int64_t n2, double* out, int64_t ld, int lower, int add_diag, int64_t nv1,
                  int64_t nv2) {
  const int nti = (int)((n1 + FT_ROWS - 1) / FT_ROWS), ntj = (int)((n2 + FT_COLS - 1) / FT_COLS);
  int64_t nblocks = lower ? lower_blocks(nti, ntj) : (int64_t)nti * ntj;
  if (nblocks <= 0) return 0;
  if (nblocks > 0x7fffffffLL) return -1;
  const int vec_ok = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  dim3 grid((unsigned)nblocks), block(256);
#define FILL_UNR(U)                                                                                          \
  hipLaunchKernelGGL((fill_kernel<KID, 4, 0, U>), grid, block, 0, st, p, x1, n1, x2, n2, out, ld, lower, add_diag, \
                     nv1, nv2, nti, ntj, vec_ok)
  // columns per thread and pass, measured at N = 131 072 (steady GB/s for 2 / 4 / 8 / 16): K0 5810 / 5750 / 5640 / 5780,
  // Matern 5440 / 5570 / 5440 / 5470 - the winners are compiled in, the sweep's other instantiations and its knob are gone
  constexpr int UNR_DEFAULT = (KID == BGP_KERNEL_BATTGP) ? 2 : 4;
#ifdef BGP_EXPERIMENTAL  // the two interior variants and their environment knobs: experimental library only (A/B pending)
  static const bool t256 = getenv("BGP_FILL_TABLE") && atoi(getenv("BGP_FILL_TABLE")) == 256;
  static const bool fmfma = getenv("BGP_FILL_MFMA") && atoi(getenv("BGP_FILL_MFMA")) == 1;
  if (p.D == 4 && fmfma && t256)
    hipLaunchKernelGGL((fill_mfma_kernel<KID, 12, UNR_DEFAULT>), grid, block, 0, st, p, x1, n1, x2, n2, out, ld, lower, add_diag,
                       nv1, nv2, nti, ntj, vec_ok);
  else if (p.D == 4 && fmfma)
    hipLaunchKernelGGL((fill_mfma_kernel<KID, 8, UNR_DEFAULT>), grid, block, 0, st, p, x1, n1, x2, n2, out, ld, lower, add_diag,
                       nv1, nv2, nti, ntj, vec_ok);
  else if (p.D == 4 && t256)
    hipLaunchKernelGGL((fill_kernel<KID, 4, 4, UNR_DEFAULT>), grid, block, 0, st, p, x1, n1, x2, n2, out, ld, lower, add_diag,
                       nv1, nv2, nti, ntj, vec_ok);
  else
#endif
  if (p.D == 4) FILL_UNR(UNR_DEFAULT);
  else
    hipLaunchKernelGGL((fill_kernel<KID, 0>), grid, block, 0, st, p, x1, n1, x2, n2, out, ld, lower,
                       add_diag, nv1, nv2, nti, ntj, vec_ok);
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
  __shared__ double sT[64];
  if (threadIdx.x < 64) sT[threadIdx.x] = EXP2_TBL[threadIdx.x];
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
      double kv = kfun<KID, DD>(a, b, D, p.s0, p.s1, sT);
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
                                                        SlabView L, int64_t n, double diag_add, int nsample,
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
  for (int64_t q = threadIdx.x; q <= j; q += 64) acc = __builtin_fma(*L.at(i, q), *L.at(j, q), acc);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (threadIdx.x == 0) {
    double a[DD], b[DD];
    load_point<KID, DD>(x, i, n, p.D, p, a);
    load_point<KID, DD>(x, j, n, p.D, p, b);
    double kij = kfun<KID, DD>(a, b, p.D, p.s0, p.s1, EXP2_TBL) + (i == j ? diag_add : 0.0);
    double kii = kfun<KID, DD>(a, a, p.D, p.s0, p.s1, EXP2_TBL) + diag_add;
    double kjj = kfun<KID, DD>(b, b, p.D, p.s0, p.s1, EXP2_TBL) + diag_add;
    out_err[s] = __builtin_fabs(acc - kij) / __builtin_sqrt(kii * kjj);
  }
}


// ---- LML gradient reduction ------------------------------------------------------------------
//   d lml / d theta = 1/2 sum_jk W_jk dSigma_jk/dtheta,   W = alpha alpha^T - P,  P = Sigma^-1 (LOWER triangle stored)
// One pass over a lower trapezoid of P that starts ON the diagonal - rows [r0, r0 + nrows) x columns [r0, r0 + ncols)
// of the matrix: a column slab of the single-GPU layout or one column panel of the sharded one, element (i, j) at
// P[(i - r0) + (j - r0) ldp] - in 512 x 32 tiles, rows contiguous like the fill; kernel derivatives re-evaluated on the
// fly, NACC partial sums per workgroup (deterministic two-stage reduction):
//   acc[0] = sum W_jj (j < N)                              -> d/d noise
//   acc[1] = sum' W_jk * wiener(t_j, t_k)      (K0 only)   -> d/d s_wiener
//   acc[2] = sum' W_jk * g_jk,  g = exp part ((1+r) e^-r for Matern)      -> d/d outputscale
//   acc[3+d] = sum' W_jk * h_jk * u_d^2,  h = exp part (e^-r for Matern), u_d = scaled difference
// where sum' counts each off-diagonal pair twice (symmetry) and the diagonal once.
constexpr int GR_NACC = 3 + BGP_MAX_DIM;

template <int KID>
__global__ __launch_bounds__(256) void grad_reduce_kernel(FillParams p, const double* __restrict__ x, int64_t n, int64_t r0,
                                                          const double* __restrict__ P, int64_t ldp,
                                                          const double* __restrict__ alpha, int nti, int ntj,
                                                          int64_t iend, int64_t jend, double* __restrict__ part) {
  // iend / jend: one past the last row / column of the described block that lies inside the matrix
  // (min(n, r0 + nrows), min(n, r0 + ncols)): a block that is not a whole number of tiles is not read beyond its edge
  constexpr int DD = BGP_MAX_DIM;
  const int D = p.D;
  __shared__ double sB[FT_COLS][DD];
  __shared__ double sAl[FT_COLS];
  __shared__ double sT[64];
  __shared__ double sred[4][GR_NACC];
  if (threadIdx.x < 64) sT[threadIdx.x] = EXP2_TBL[threadIdx.x];
  double acc[GR_NACC];
#pragma unroll
  for (int q = 0; q < GR_NACC; ++q) acc[q] = 0.0;

  int ti, tj;
  const bool valid_tile = lower_decode((int64_t)blockIdx.x, nti, ntj, ti, tj);
  if (valid_tile) {
    const int64_t i0 = r0 + (int64_t)ti * FT_ROWS, j0 = r0 + (int64_t)tj * FT_COLS;
    for (int idx = threadIdx.x; idx < FT_COLS * DD; idx += 256) {
      int c = idx / DD, d = idx % DD;
      int64_t j = j0 + c;
      double v = 0.0;
      if (j < n && d < D) {
        v = x[j * D + d];
        if (!(KID == BGP_KERNEL_BATTGP && d == 0)) v *= p.scale[d];
      }
      sB[c][d] = v;
    }
    if (threadIdx.x < FT_COLS) sAl[threadIdx.x] = (j0 + threadIdx.x < n) ? alpha[j0 + threadIdx.x] : 0.0;
    const int64_t i = i0 + 2 * (int64_t)threadIdx.x;
    double a[2][DD];
    load_point<KID, DD>(x, i, n, D, p, a[0]);
    load_point<KID, DD>(x, i + 1, n, D, p, a[1]);
    const double al[2] = {i < n ? alpha[i] : 0.0, i + 1 < n ? alpha[i + 1] : 0.0};
    __syncthreads();
    for (int c = 0; c < FT_COLS; ++c) {
      const int64_t j = j0 + c;
      if (j >= jend) break;
      double b[DD];
#pragma unroll
      for (int d = 0; d < DD; ++d) b[d] = sB[c][d];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int64_t ii = i + r;
        if (ii < j || ii >= iend) continue;  // strict upper triangle of the block; rows past the block / the matrix (padding)
        const double wgt = (ii == j) ? 1.0 : 2.0;
        const double W = wgt * (al[r] * sAl[c] - P[(ii - r0) + (j - r0) * ldp]);
        if (ii == j) acc[0] += W;
        double q = 0.0, u2[DD];
#pragma unroll
        for (int d = 0; d < DD; ++d) {
          u2[d] = 0.0;
          if (d < D && !(KID == BGP_KERNEL_BATTGP && d == 0)) {
            const double df = a[r][d] - b[d];
            u2[d] = df * df;
            q += u2[d];
          }
        }
        double g, hh;
        if (KID == BGP_KERNEL_MATERN32) {
          const double rr = sqrt_nonneg(q);
          hh = exp_nonpos(-rr, sT);
          g = (1.0 + rr) * hh;
        } else {
          hh = exp_nonpos(-q, sT);
          g = hh;
        }
        if (KID == BGP_KERNEL_BATTGP) {
          const double m = __builtin_fmin(a[r][0], b[0]);
          const double ad = __builtin_fabs(a[r][0] - b[0]);
          const double m2 = m * m;
          acc[1] = __builtin_fma(W, (m2 * m) * (1.0 / 3.0) + (ad * m2) * 0.5, acc[1]);
        }
        acc[2] = __builtin_fma(W, g, acc[2]);
        const double Wh = W * hh;
#pragma unroll
        for (int d = 0; d < DD; ++d) acc[3 + d] = __builtin_fma(Wh, u2[d], acc[3 + d]);
      }
    }
  }
  // workgroup reduction (fixed order)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < GR_NACC; ++q) {
    double v = acc[q];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) sred[wave][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < GR_NACC)
    part[(int64_t)blockIdx.x * GR_NACC + threadIdx.x] =
        (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
}

// out[q] (+)= sum_b part[b * GR_NACC + q]   (one workgroup, fixed order => run-to-run identical)
__global__ __launch_bounds__(1024) void grad_finish_kernel(const double* __restrict__ part, int64_t nblocks,
                                                           double* __restrict__ out, int accumulate) {
  __shared__ double red[1024];
  for (int q = 0; q < GR_NACC; ++q) {
    double v = 0.0;
    for (int64_t b = threadIdx.x; b < nblocks; b += 1024) v += part[b * GR_NACC + q];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[q] = accumulate ? out[q] + red[0] : red[0];
    __syncthreads();
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
                      const SlabView& L, int64_t n, double diag_add, int nsample,
                      double* out_err) {
  dim3 grid((unsigned)nsample), block(64);
  BGP_KID_SWITCH(p.kid, hipLaunchKernelGGL((llt_sample_kernel<KID_>), grid, block, 0, st, p, x, L,
                                           n, diag_add, nsample, out_err));
  BGP_HIP(h, hipGetLastError());
  return 0;
}

int grad_nacc() { return GR_NACC; }

int64_t grad_blocks(int64_t nrows, int64_t ncols) {
  return lower_blocks((int)((nrows + FT_ROWS - 1) / FT_ROWS), (int)((ncols + FT_COLS - 1) / FT_COLS));
}

// one lower trapezoid of P (see grad_reduce_kernel); out[0 .. GR_NACC) receives (accumulate: is increased by) its sums
int launch_grad_reduce(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x, int64_t n, int64_t r0,
                       int64_t nrows, int64_t ncols, const double* P, int64_t ldp, const double* alpha, double* part,
                       double* out, int accumulate) {
  const int nti = (int)((nrows + FT_ROWS - 1) / FT_ROWS), ntj = (int)((ncols + FT_COLS - 1) / FT_COLS);
  const int64_t nb = lower_blocks(nti, ntj);
  if (nb <= 0) return 0;
  if (nb > 0x7fffffffLL) return bgp_fail(h, -1, "grad_reduce: grid too large");
  const int64_t iend = std::min(n, r0 + nrows), jend = std::min(n, r0 + ncols);
  BGP_KID_SWITCH(p.kid, hipLaunchKernelGGL((grad_reduce_kernel<KID_>), dim3((unsigned)nb), dim3(256), 0, st, p, x, n, r0, P,
                                           ldp, alpha, nti, ntj, iend, jend, part));
  BGP_HIP(h, hipGetLastError());
  hipLaunchKernelGGL(grad_finish_kernel, dim3(1), dim3(1024), 0, st, part, nb, out, accumulate);
  BGP_HIP(h, hipGetLastError());
  return 0;
}
