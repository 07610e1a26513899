// Internal declarations shared by the HIP translation units of libbattgp.so (gfx950 only).
#pragma once

// -DBGP_EXPERIMENTAL (battgp_amd/build.py --experimental -> libbattgp_exp.so): also compiles the optional kernel families
// that no MI355X has timed yet - slim chain kernels (look-ahead bit 5), split panels (bit 6), fused update + tile Cholesky
// (bit 7), the table-256 and matrix-pipe interiors of the fill (BGP_FILL_TABLE / BGP_FILL_MFMA).  The default library
// holds none of them: bgp_set_options rejects the three bits and the two environment knobs are not read.
#ifdef BGP_EXPERIMENTAL
constexpr bool BGP_EXP = true;
#else
constexpr bool BGP_EXP = false;
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/battgp.h"

#define BGP_IB 64  // inner (diagonal tile) block of the Cholesky; also the K-granule of the path
#define BGP_RD_COLS 128  // columns per workgroup of the row-dot reductions (partial sums: ceil(n / 128) per row)
#define BGP_MAX_WBUF 5  // solved-panel workspaces (look-ahead depth <= 4)
#define BGP_AUG 64 // rows of the augmented block below the matrix (row 0 of it carries y^T)

struct FillParams {
  int kid;         // BGP_KERNEL_*
  int D;           // columns of X
  double noise;    // added on the diagonal of the training fill (noise + jitter)
  double s0;       // s_wiener (K0) or outputscale s (K1..K3)
  double s1;       // s_rbf (K0)
  double scale[BGP_MAX_DIM];  // per-column input scale: 1/(l*sqrt 2) (RBF) or sqrt(3)/l (Matern); K0 col 0 unused
  int t_sorted;    // K0 training fill: column 0 of X is ascending, so below the diagonal min(t_i, t_j) = t_j
};

// Storage of the in-place covariance / factor.  The columns are cut into slabs of width W; slab g
// keeps only rows [g W, L) - nothing above its own diagonal block - as one column-major block
// with leading dimension L - g W, slabs back to back.  W >= Npad is the plain full-square layout
// (one slab, ld = L).  With W << N the footprint drops from 8 N^2 to ~4 N (N + W) bytes, which is
// what lets N = 262 144 live on ONE 288 GB MI355X.  Every launch of the path works on columns of
// one slab (panels never straddle: W is a multiple of the outer panel width), so the kernels keep
// their plain (pointer, ld) interface.
struct SlabView {
  double* base;
  int64_t L;  // rows of slab 0 (= Npad + rows riding below the matrix)
  int64_t W;  // slab width in columns
  __host__ __device__ int64_t slab(int64_t c) const { return c / W; }
  __host__ __device__ int64_t ld(int64_t c) const { return L - slab(c) * W; }
  __host__ __device__ int64_t offset(int64_t r, int64_t c) const {
    const int64_t g = c / W, c0 = g * W;
    return W * (g * L - W * (g * (g - 1) / 2)) + (r - c0) + (c - c0) * (L - c0);
  }
  __host__ __device__ double* at(int64_t r, int64_t c) const { return base + offset(r, c); }
  // first column after c's slab, clipped to n
  __host__ __device__ int64_t slab_end(int64_t c, int64_t n) const {
    const int64_t e = (slab(c) + 1) * W;
    return (e < n && e > 0) ? e : n;
  }
  // doubles needed for ncols columns
  static int64_t total(int64_t L, int64_t W, int64_t ncols) {
    int64_t t = 0;
    for (int64_t c0 = 0; c0 < ncols; c0 += W) t += (L - c0) * ((ncols - c0 < W) ? (ncols - c0) : W);
    return t;
  }
};
#define BGP_W_FULL ((int64_t)1 << 40)  // "one slab": any W >= Npad

struct bgp_handle {
  int device = 0;
  hipStream_t s_main = nullptr, s_aux = nullptr, s_copy = nullptr, s_bulk = nullptr;  // s_bulk: tall parts of a split panel (lookahead bit 6)
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
  std::vector<hipEvent_t> ev_phase; // one begin / end pair per phase-timer slot (PhaseTimer)
  uint32_t phase_pending = 0, phase_acc = 0;  // slots whose end event is recorded but not read yet / that accumulate
  std::vector<hipEvent_t> ev_pool;  // timing pairs around trailing updates
  std::vector<hipEvent_t> ev_sync;  // cross-stream dependencies of the look-ahead schedule
  // kernel
  bool kernel_set = false;
  int kernel_id = 0, nhyp = 0;
  double hyp[BGP_MAX_HYP] = {0};
  // options
  int nb_outer = 512;
  bool nb_auto = true;   // no explicit bgp_set_options(nb_outer): 1024 for Npad >= 32768 (deeper trailing updates), else 512
  int max_tries = 3;
  double jitter0 = 1e-8;
  int lookahead = 1;
  int panel_mode = -1;   // 1: chain on the diagonal block + one deep TRSM-by-inverse GEMM; 0: 64-wide chain over all rows;
                         // -1 (default): 0 below Npad = 16384, 1 from there on (measured crossover)
  int64_t ld_pad = 0;    // bgp_debug_set_ld_pad: unused rows appended to every column (index arithmetic at large strides on small problems)
  int64_t slab_req = 0;  // bgp_set_layout: 0 auto (full square if it fits, else slabs), -1 full square, > 0 width
  // problem
  int64_t N = 0, Npad = 0, lda = 0;
  int64_t slabW = BGP_W_FULL;  // slab width of dA (BGP_W_FULL: full-square layout, ld = lda)
  int64_t A_doubles = 0;       // allocated size of dA
  SlabView view() const { return SlabView{dA, lda, slabW}; }
  int64_t aug_cap = BGP_AUG;   // rows allocated below the matrix: 64 (y block) + room for riding query rows
  int64_t aug_used = BGP_AUG;  // rows of it that took part in the last factorisation
  int D = 0;
  bool fitted = false;
  bool has_data = false;     // X, y were uploaded through THIS life of the handle (a revived pooled handle starts without)
  bool t_sorted = false;     // column 0 of the resident X is ascending (checked on the device at upload)
  bool alpha_ready = false;
  bool factor_consumed = false;  // bgp_lml_grad has overwritten L with Sigma^-1 in place: the next call that needs the factor
                                 // re-runs the fit on the resident data first (ensure_factor)
  bool keep_factor = false;      // bgp_set_keep_factor: bgp_lml_grad saves the factor storage first (when memory allows) and
  bool keep_valid = false;       // ensure_factor copies it back instead of re-running the fit; dA_keep holds the current factor
  double jitter_used = 0.0, lml = 0.0;
  // device buffers
  double* dX = nullptr;      // [N, D] row-major
  double* dy = nullptr;      // [N]
  double* dA = nullptr;      // column slabs (SlabView) of [lda, Npad] column-major, lower triangle = Sigma then L
  double* dA_keep = nullptr; // copy of dA taken by bgp_lml_grad under keep_factor (A_keep_doubles doubles, 0 = none)
  int64_t A_keep_doubles = 0;
  int64_t keep_failed_doubles = 0;  // size (in doubles) for which the keep buffer's allocation last failed: not retried until
                                    // the factor's size changes or bgp_set_keep_factor is called again
  double* dInv = nullptr;    // [Npad/64][64*64] inverses of the diagonal tiles of L
  double* dz = nullptr;      // [Npad] z = L^-1 y (zero in the padding)
  double* dalpha = nullptr;  // [Npad]
  double* dLinvAll = nullptr;  // inv(L_pp) of every outer panel [npanels][nbL * nbL]: later query blocks (predict after fit)
  int64_t LinvAll_cap = 0;
  int64_t LinvAll_nb = 0;      // panel width they were built for; 0 = not valid for the current factor
  // panel workspaces (panel_mode 1): diagonal block + riding identity, L_kk^-1, two solved-panel buffers
  double* dD = nullptr;      // [2 nbw, nbw]
  double* dLinv = nullptr;   // [nbw, nbw]
  double* dW[BGP_MAX_WBUF] = {nullptr};  // [ldw, nbw] each; look-ahead depth + 1 of them in use
  int64_t nbw = 0, ldw = 0;
  int nwbuf = 0;
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
                   int64_t k, int lower, const int* abort_flag = nullptr, int btri = 0);
int launch_trinv_panels(bgp_handle* h, hipStream_t st, const SlabView& L, const double* inv_tiles, double* Linv_all,
                        int64_t n, int NB);
int launch_diag_in(bgp_handle* h, hipStream_t st, const double* Akk, int64_t lda, double* D, int64_t ldd, int nbk);
int launch_diag_out(bgp_handle* h, hipStream_t st, const double* D, int64_t ldd, double* Akk, int64_t lda,
                    double* Linv, int64_t ldl, int nbk, int slim = 0);
int launch_copy_panel(bgp_handle* h, hipStream_t st, const double* src, int64_t lds_, double* dst, int64_t ldd,
                      int64_t rows, int ncols);
// slim != 0 (here and in launch_diag_out): the variants that fit next to two resident trailing-update workgroups
// (<= 64 VGPRs, <= 12 KB LDS; bgp_linalg.hip "slim chain kernels"); bit-identical results
int launch_potrf_tile(bgp_handle* h, hipStream_t st, double* Ajj, int64_t lda, double* inv,
                      int* info, int col0, int slim = 0);
// the next two exist as kernels only in the experimental library (-DBGP_EXPERIMENTAL); the default library's versions fail
int launch_chain_update_potrf(bgp_handle* h, hipStream_t st, double* C, int64_t ldc, const double* A, int64_t lda,
                              const double* B, int64_t ldb, int64_t m, int64_t n, int lower, int* info, double* inv_next,
                              int col0_next, int slim = 0);
int launch_chain_gemm_slim(bgp_handle* h, hipStream_t st, int mode, double* C, int64_t ldc, const double* A, int64_t lda,
                           const double* B, int64_t ldb, int64_t m, int64_t n, int lower, const int* abort_flag);
int launch_fit_scalars(bgp_handle* h, hipStream_t st, const SlabView& A, const double* z,
                       int64_t ldz, int64_t n, double* out2);
int launch_aug_rows(bgp_handle* h, hipStream_t st, const double* y, int64_t n, double* Aaug, int64_t lda,
                    int64_t ncols, int naug);  // Aaug[r + j*lda] = (r == 0 && j < n) ? y[j] : 0
int launch_gather_row(bgp_handle* h, hipStream_t st, const double* row, int64_t ld, int64_t n, double* dst);
int launch_gemv_t_partial(bgp_handle* h, hipStream_t st, const double* P, int64_t lda, const double* x,
                          int64_t rows, int nbk, double* part, int* nchunks_out);
int launch_trsv_block_bwd(bgp_handle* h, hipStream_t st, const double* Lkk, int64_t lda, const double* invK,
                          const double* z, const double* part, int nchunks, int nbk, double* alpha);
int launch_linvT_gemv(bgp_handle* h, hipStream_t st, const double* Linv, int64_t ldl, const double* z, const double* part,
                      int nchunks, int nbk, double* alpha);
int launch_rowdot(bgp_handle* h, hipStream_t st, const double* E, int64_t lde, int64_t M, int64_t n,
                  const double* vec /*null => E*E*/, double* part, int* nchunks_out);
int launch_rowdot_finish(bgp_handle* h, hipStream_t st, const double* part, int nchunks, int64_t M,
                         const double* kdiag_x /*null => mean*/, const FillParams* p, double min_var,
                         double* out);
int launch_kmatvec(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x, int64_t n,
                   const double* v, double diag_add, double* out);
int launch_llt_sample(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x,
                      const SlabView& L, int64_t n, double diag_add, int nsample,
                      double* out_max);
int launch_norm2(bgp_handle* h, hipStream_t st, const double* a, const double* b /*null*/, int64_t n,
                 double* out);  // out[0] = sum (a-b)^2 or sum a^2
int launch_copy_strided(bgp_handle* h, hipStream_t st, const double* src, int64_t n, double* dst,
                        int64_t ld_dst, int64_t npad);  // dst[i*ld_dst] = src[i] (i<n) else 0
int grad_nacc();
int64_t grad_blocks(int64_t nrows, int64_t ncols);
// one lower trapezoid of P = Sigma^-1 that starts on the diagonal at r0 (a column slab / a sharded column panel)
int launch_grad_reduce(bgp_handle* h, hipStream_t st, const FillParams& p, const double* x, int64_t n, int64_t r0,
                       int64_t nrows, int64_t ncols, const double* P, int64_t ldp, const double* alpha, double* part,
                       double* out, int accumulate);
int launch_flag_store(bgp_handle* h, hipStream_t st, const int* info, double* slot);
int launch_flag_merge(bgp_handle* h, hipStream_t st, int* info, const double* slot);
int launch_check_sorted(bgp_handle* h, hipStream_t st, const double* x, int64_t n, int D, int* flag);  // *flag = 1 if a descent is found
// dst = scale * op(src[rows, cols]); trans: dst[c + r ldd]; tri: src entries with r < c count as zero
int launch_block_copy(bgp_handle* h, hipStream_t st, const double* src, int64_t lds_, int64_t rows, int64_t cols, double* dst,
                      int64_t ldd, int trans, double scale, int tri);
