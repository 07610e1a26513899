// Internal declarations shared by the HIP translation units of libbattgp.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/battgp.h"

#define BGP_IB 64  // inner (diagonal tile) block of the Cholesky; also the K-granule of the path
#define BGP_AUG 64 // rows of the augmented block below the matrix (row 0 of it carries y^T)

struct FillParams {
  int kid;         // BGP_KERNEL_*
  int D;           // columns of X
  double noise;    // added on the diagonal of the training fill (noise + jitter)
  double s0;       // s_wiener (K0) or outputscale s (K1..K3)
  double s1;       // s_rbf (K0)
  double scale[BGP_MAX_DIM];  // per-column input scale: 1/(l*sqrt 2) (RBF) or sqrt(3)/l (Matern); K0 col 0 unused
};

struct bgp_handle {
  int device = 0;
  hipStream_t s_main = nullptr, s_aux = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
  std::vector<hipEvent_t> ev_pool;  // timing pairs around trailing updates
  std::vector<hipEvent_t> ev_sync;  // cross-stream dependencies of the look-ahead schedule
  // kernel
  bool kernel_set = false;
  int kernel_id = 0, nhyp = 0;
  double hyp[BGP_MAX_HYP] = {0};
  // options
  int nb_outer = 512;
  int max_tries = 3;
  double jitter0 = 1e-8;
  int lookahead = 1;
  // problem
  int64_t N = 0, Npad = 0, lda = 0;
  int64_t aug_cap = BGP_AUG;   // rows allocated below the matrix: 64 (y block) + room for riding query rows
  int64_t aug_used = BGP_AUG;  // rows of it that took part in the last factorisation
  int D = 0;
  bool fitted = false;
  bool alpha_ready = false;
  double jitter_used = 0.0, lml = 0.0;
  // device buffers
  double* dX = nullptr;      // [N, D] row-major
  double* dy = nullptr;      // [N]
  double* dA = nullptr;      // [lda, Npad] column-major, lower triangle = Sigma then L
  double* dInv = nullptr;    // [Npad/64][64*64] inverses of the diagonal tiles of L
  double* dz = nullptr;      // [Npad] z = L^-1 y (zero in the padding)
  double* dalpha = nullptr;  // [Npad]
  double* dB = nullptr;      // gradient workspace: U = L^-T (upper), [lda, Npad]; allocated on first bgp_lml_grad
  double* dS = nullptr;      // gradient workspace: S = -Sigma^-1 (lower), [lda, Npad]
  double* dE = nullptr;      // [lde, Npad] cross-covariance row block (queries x train)
  int64_t E_rows_cap = 0;
  double* dXq = nullptr;     // [M, D]
  int64_t Xq_cap = 0;
  double* dpart = nullptr;   // partial sums workspace
  int64_t part_cap = 0;
  double* dout = nullptr;    // small result vectors (mean, var) [2 * out_cap]
  int64_t out_cap = 0;
  double* dscal = nullptr;   // 16 scalars
  int* dinfo = nullptr;
  // pinned host
  double* hscal = nullptr;
  int* hinfo = nullptr;
  int64_t bytes = 0;
  double times[BGP_T_COUNT] = {0};
  std::string err;
};

// ---- error helpers -------------------------------------------------------------------------
int bgp_fail(bgp_handle* h, int code, const char* fmt, ...);
#define BGP_HIP(h, call)                                                                     \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return bgp_fail((h), -2, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                      __LINE__);                                                             \
  } while (0)

// ---- kernel launchers (bgp_kernels.hip) ----------------------------------------------------
int launch_fill(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x1, int64_t n1,
                const double* x2, int64_t n2, double* out, int64_t ld, int lower, int add_diag,
                int64_t nvalid1, int64_t nvalid2);
int launch_gemm_nt(bgp_handle* h, hipStream_t st, int mode, int tn, double* C, int64_t ldc,
                   const double* A, int64_t lda, const double* B, int64_t ldb, int64_t m, int64_t n,
                   int64_t k, int lower, const int* abort_flag = nullptr);
int launch_potrf_tile(bgp_handle* h, hipStream_t st, double* Ajj, int64_t lda, double* inv,
                      int* info, int col0, int nvalid);
int launch_fit_scalars(bgp_handle* h, hipStream_t st, const double* A, int64_t lda, const double* z,
                       int64_t ldz, int64_t n, double* out2);
int launch_aug_rows(bgp_handle* h, hipStream_t st, const double* y, int64_t n, double* Aaug, int64_t lda,
                    int64_t ncols, int naug);  // Aaug[r + j*lda] = (r == 0 && j < n) ? y[j] : 0
int launch_gather_row(bgp_handle* h, hipStream_t st, const double* row, int64_t ld, int64_t n, double* dst);
int launch_gemv_t_partial(bgp_handle* h, hipStream_t st, const double* P, int64_t lda, const double* x,
                          int64_t rows, int nbk, double* part, int* nchunks_out);
int launch_trsv_block_bwd(bgp_handle* h, hipStream_t st, const double* Lkk, int64_t lda, const double* invK,
                          const double* z, const double* part, int nchunks, int nbk, double* alpha);
int launch_rowdot(bgp_handle* h, hipStream_t st, const double* E, int64_t lde, int64_t M, int64_t n,
                  const double* vec /*null => E*E*/, double* part, int* nchunks_out);
int launch_rowdot_finish(bgp_handle* h, hipStream_t st, const double* part, int nchunks, int64_t M,
                         const double* kdiag_x /*null => mean*/, const FillParams* p, double min_var,
                         double* out);
int launch_kmatvec(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x, int64_t n,
                   const double* v, double diag_add, double* out);
int launch_llt_sample(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x,
                      const double* L, int64_t lda, int64_t n, double diag_add, int nsample,
                      double* out_max);
int launch_norm2(bgp_handle* h, hipStream_t st, const double* a, const double* b /*null*/, int64_t n,
                 double* out);  // out[0] = sum (a-b)^2 or sum a^2
int launch_copy_strided(bgp_handle* h, hipStream_t st, const double* src, int64_t n, double* dst,
                        int64_t ld_dst, int64_t npad);  // dst[i*ld_dst] = src[i] (i<n) else 0
int grad_nacc();
int64_t grad_blocks(int64_t n);
int launch_grad_reduce(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x, int64_t n,
                       const double* S, int64_t lds_, const double* alpha, double* part, double* out);
int launch_set_identity(bgp_handle* h, hipStream_t st, double* B, int64_t ld, int64_t n);
