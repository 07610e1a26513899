/*
 * battgp.h - C-ABI of the MI355X-native exact-GP engine (libbattgp.so).
 *
 * This is the drop-in boundary for BattGP's `full_gp` hot path.  The reference is pure
 * Python over gpytorch/torch (no FFI of its own), so every entry point below names the
 * reference call site whose device work it replaces (paths relative to the reference
 * root).  Plain pointers and sizes only; no torch types.  The reference-side binding (a
 * ctypes stub) is shown in INTEGRATION.md and implemented in battgp_amd/_lib.py.
 *
 * Conventions
 *   - all floating point data is IEEE fp64;
 *   - X / Xq are row-major [N, D] / [M, D] exactly as IBatteryCellGP defines them
 *     (src/batt_models/batt_cell_gp_protocol.py:18-30: time[days], I[A], SOC[%], T[degC]);
 *   - *_host pointers are borrowed for the duration of the call, *_dev pointers are HIP
 *     device pointers on the handle's device (e.g. torch.Tensor.data_ptr()) that must
 *     already be complete on entry (the engine runs on its own streams);
 *   - return value: 0 ok; > 0 = matrix not positive definite after the jitter ladder
 *     (value = 1-based index of the failing leading minor); < 0 = argument / HIP /
 *     allocation error, text available from bgp_last_error();
 *   - a handle is single-threaded; distinct handles (also on distinct GPUs) are
 *     independent and may be driven from different host threads concurrently
 *     (src/batt_models/battgp.py:191-216 trains cells from a thread pool).
 *
 * Hyper-parameter vector layout (hyp):
 *   BGP_KERNEL_BATTGP      [noise, s_wiener, s_rbf, l_1 .. l_{D-1}]   (src/batt_models/cell_gp.py:27-36,
 *                           WienerKernel on column 0: src/gp/wiener_kernel.py:10-32)
 *   BGP_KERNEL_SCALED_RBF  [noise, s, l]                               (src/gp/standard_models.py:24-28)
 *   BGP_KERNEL_MATERN32    [noise, s, l_1 .. l_D]                      (BASELINE config 3; no reference call site)
 *   BGP_KERNEL_ARD_RBF     [noise, s, l_1 .. l_D]                      (tests/gp/test_spatiotemporal_gp.py:65-81)
 */
#ifndef BATTGP_H
#define BATTGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bgp_handle bgp_handle;

enum {
  BGP_KERNEL_BATTGP = 0,
  BGP_KERNEL_SCALED_RBF = 1,
  BGP_KERNEL_MATERN32 = 2,
  BGP_KERNEL_ARD_RBF = 3
};

#define BGP_MAX_DIM 8
#define BGP_MAX_HYP (BGP_MAX_DIM + 3)

/* indices into the bgp_phase_times() array (milliseconds, last call of each phase) */
enum {
  BGP_T_H2D = 0,      /* host->device upload of X, y                    */
  BGP_T_FILL = 1,     /* covariance fill  K + (noise+jitter) I           */
  BGP_T_POTRF = 2,    /* blocked Cholesky (all attempts of the ladder)   */
  BGP_T_SOLVE = 3,    /* z gather + log det + z^T z; lazy alpha = L^-T z */
  BGP_T_CROSS = 4,    /* cross-covariance fill (+ mean for mean-only predictions) */
  BGP_T_VAR = 5,      /* V^T = K_*X L^-T pass (if not fused), mean = V^T z, variance */
  BGP_T_D2H = 6,      /* device->host of results                         */
  BGP_T_TRAIL = 7,    /* sum of the outer trailing-update (MFMA SYRK) launches of the last potrf */
  BGP_T_TRAIL_FLOP = 8, /* algorithmic flop of those launches (2*m*n*k per full tile pair, lower half) */
  BGP_T_FILL_BYTES = 9, /* algorithmic bytes written by the last training fill */
  BGP_T_TRAIL_LAUNCHES = 10, /* number of those trailing-update launches */
  BGP_T_TRAIL_UNION = 11, /* time during which at least one of them was running (they overlap on two streams) */
  BGP_T_GRAD = 12,    /* bgp_lml_grad: Sigma^-1 in place over the factor + the fused reduction pass (+ the copy that saves
                         the factor first, under bgp_set_keep_factor) */
  BGP_T_RESTORE = 13, /* the most recent bringing-back of the factor after a gradient consumed it (copy under
                         bgp_set_keep_factor, else a re-run of the fit on the resident data); 0 = none since the last fit.
                         The fit's own FILL / POTRF / SOLVE / TRAIL* slots are NOT touched by that re-run, whether
                         it succeeds or fails */
  BGP_T_COUNT = 14
};

/* Library version (major*10000 + minor*100 + patch). */
int bgp_version(void);

/* Create / destroy an engine bound to HIP device `device`.  The handle owns every device
 * buffer (K/L in place, X, y, alpha, workspaces).  bgp_destroy is what
 * `del cellmodel.model; gc.collect(); torch.cuda.empty_cache()` does in the reference
 * (src/batt_models/battgp_full.py:102-120) - with the allocator behaviour of torch: the reference
 * builds one model per cell and deletes it after the prediction (battgp_full.py:41-60), so a destroyed
 * handle is PARKED with its streams, events and buffers (at most BGP_POOL = 16 per device holding at most
 * BGP_POOL_BYTES = 40 GiB of buffers between them - a handle that would exceed that parks without its
 * buffers; environment variables; BGP_POOL=0 = free at once) and the next bgp_create on
 * that device revives it as a logically new handle; a problem of the same size then finds its buffers in place (measured per cell: -17 ms, and -147 ms of hipMalloc + first touch
 * at N = 40 000).  Parked memory is released when an allocation fails, when the automatic layout needs
 * it, and by bgp_trim(device) (device < 0: all devices) - the counterpart of torch.cuda.empty_cache(). */
int bgp_create(bgp_handle** out, int device);
void bgp_destroy(bgp_handle* h);
int bgp_trim(int device);

/* Text of the last error on this handle ("" if none).  With h == NULL: last create error. */
const char* bgp_last_error(const bgp_handle* h);

/* Select the covariance function and its hyper-parameters.  Replaces the gpytorch module
 * construction + property setters of src/batt_models/cell_gp.py:27-36,64-194 and
 * src/batt_models/battcellgp_full.py:71-84.  Values are taken verbatim as fp64. */
int bgp_set_kernel(bgp_handle* h, int kernel_id, const double* hyp, int nhyp);

/* Tuning / numerical options.  Any argument < 0 (or NaN) keeps the current value.
 *   nb_outer         outer panel width of the blocked Cholesky (multiple of 64 in [64, 2048]; default: 512 below
 *                    N = 32 768, 1024 from there on).  Changing it invalidates the stored factor: fit again before
 *                    the next predict / gradient call
 *   max_tries        rungs of the jitter ladder after the plain attempt (default 3)
 *   jitter0          first rung (default 1e-8: linear_operator psd_safe_cholesky, fp64)
 *   lookahead        bits 0-2: depth d of the look-ahead (0 = off, default 1, <= 4): the panel stream factors
 *                    panel k+1 and updates it left-looking from the last d panels while the main stream
 *                    applies panel k to the panels from k+d+1 on (d > 1 measured slower: the panel stream is
 *                    the critical path); +8: the panel stream's updates are ordered before the main
 *                    stream's (per-launch timings do not overlap, ~1 % slower); +16: trailing updates
 *                    without the atomic-accumulate epilogue (ablation); +32: the diagonal-block chain of a panel that
 *                    is factored underneath a trailing update uses kernels sized to fit NEXT TO the update's two
 *                    workgroups per CU (<= 64 VGPRs, <= 12 KB LDS) instead of queueing for a CU slot; +64: split
 *                    panels - only the next diagonal block's rows of a panel's solve and look-ahead update stay on
 *                    the panel stream, the tall rest runs on a fourth stream (depth 1, panel scheme 1); +128: the
 *                    rank-64 update of a 64-column chain step and the tile Cholesky of the next step share one launch
 *                    (two dependent launches per 64 columns instead of three).
 *                    Bit-identical results for all.  +32 / +64 / +128 exist in the EXPERIMENTAL library only
 *                    (libbattgp_exp.so, built with -DBGP_EXPERIMENTAL: written while no GPU was available to the build,
 *                    bit-identity shown on the CPU build of the kernel sources, never timed); the default library
 *                    returns -1 for a word that has one of them set and applies nothing of the call. */
int bgp_set_options(bgp_handle* h, int nb_outer, int max_tries, double jitter0, int lookahead);

/* Panel scheme of the blocked Cholesky (speed only; both are exact-Cholesky algebra).
 *   1   the 64-wide critical chain runs on the nb_outer x nb_outer diagonal block only (with an identity
 *       block riding along that yields L_kk^-1); the rows below are solved by ONE deep MFMA GEMM with that
 *       inverse
 *   0   the 64-wide chain {tile Cholesky, TRSM by tile inverse, rank-64 update} over all rows
 *  -1   (default) automatic: 0 below N = 16 384, 1 from there on (measured crossover) */
int bgp_set_panel_scheme(bgp_handle* h, int scheme);

/* HBM layout of the in-place covariance / Cholesky factor ("N_max per GPU", BASELINE.json metric).
 *   slab_width = -1  full square [lda, Npad] column-major: 8 N^2 bytes (N_max ~ 196 000 on 288 GB)
 *   slab_width >  0  column slabs of that width (multiple of nb_outer), each keeping only the rows
 *                    from its own diagonal block down: ~4 N (N + slab_width) bytes - N = 262 144
 *                    (BASELINE config 4's size) fits on ONE MI355X, N_max = 274 432 measured; results are
 *                    bit-identical to the full square
 *   slab_width =  0  (default) full square when it fits in free HBM, else the widest slab that does
 * Changing the layout drops the resident problem.  bgp_get_layout reports the width in use by the
 * resident problem (0 = full square) and the bytes of the factor buffer.  No reference call site
 * (gpytorch keeps the dense square, README's "A1000 ceiling" N = 40 000). */
int bgp_set_layout(bgp_handle* h, int64_t slab_width);
int bgp_get_layout(const bgp_handle* h, int64_t* slab_width_out, int64_t* factor_bytes_out);

/* FIT: upload (or adopt) X[N,D], y[N]; fill Sigma = K(X,X) + noise I in HBM; jittered
 * blocked Cholesky in place; z = L^-1 y; alpha = Sigma^-1 y; log-marginal likelihood
 *   lml = -1/2 y^T alpha - sum_i log L_ii - N/2 log 2 pi.
 * Replaces: DefaultPredictionStrategy / mean_cache + ExactMarginalLogLikelihood as used at
 * src/batt_models/battcellgp_full.py:171-173 and src/gp/training.py:27-30,39-41; jitter
 * ladder = psd_safe_cholesky (evidence: gp_runner.py:14,28).  *jitter_out = jitter added
 * (0 when the plain attempt succeeded). */
int bgp_fit(bgp_handle* h, const double* X_host, const double* y_host, int64_t N, int D,
            double* lml_out, double* jitter_out);
int bgp_fit_dev(bgp_handle* h, const double* X_dev, const double* y_dev, int64_t N, int D,
                double* lml_out, double* jitter_out);

/* Re-fit with new hyper-parameters on the resident X, y (no re-upload): one LML evaluation
 * of the optimiser loops in src/gp/training.py:39-41,126-145. */
int bgp_refit(bgp_handle* h, const double* hyp, int nhyp, double* lml_out, double* jitter_out);

/* FIT + first PREDICT in one pass: the cross-covariance rows K(Xq, X) are appended below the
 * matrix (next to the augmented y row) and ride through the factorisation, so V^T = K_*X L^-T comes
 * out of the Cholesky itself: mean = V^T z, var = k_** - rowsumsq(V^T); no separate triangular
 * solve of the query block.  This is the reference's actual flow: the model is built lazily and the
 * first predict() triggers the factorisation (src/batt_models/battcellgp_full.py:171-173,
 * battgp_full.py:100,109).  The handle stays fitted: later bgp_predict() calls with other queries
 * use the stored factor.  var_out may be NULL. */
int bgp_fit_predict(bgp_handle* h, const double* X_host, const double* y_host, int64_t N, int D,
                    const double* Xq_host, int64_t M, double* lml_out, double* jitter_out, double* mean_out,
                    double* var_out, double min_var);
int bgp_fit_predict_dev(bgp_handle* h, const double* X_dev, const double* y_dev, int64_t N, int D,
                        const double* Xq_dev, int64_t M, double* lml_out, double* jitter_out, double* mean_dev,
                        double* var_dev, double min_var);

/* Gradient of the LML of the last fit w.r.t. the hyper-parameter vector (same layout as hyp):
 *   d lml / d theta_i = 1/2 tr((alpha alpha^T - Sigma^-1) dSigma/dtheta_i).
 * Sigma^-1 is formed explicitly on the GPU, IN PLACE over the Cholesky factor - M = L^-1 by a right-looking blocked
 * inversion, then Sigma^-1 = M^T M by a blocked product, 2/3 N^3 flop that run as rank-nb updates on the same MFMA
 * kernel as the factorisation's trailing update - followed by one fused pass over its lower triangle that
 * re-evaluates the kernel derivatives.  No second N^2 buffer, in either layout (full square / column slabs):
 * training reaches the same N as inference.  The factor is consumed; the next call that needs it (bgp_predict*,
 * bgp_residuals, bgp_get_factor_*, another bgp_lml_grad at the same point) first re-runs the fit on the resident data
 * - bit-identical factor, LML and alpha - so the handle stays usable like before; the optimiser loops re-fit with new
 * hyper-parameters before every gradient anyway.  Replaces the autograd backward of src/gp/training.py:41
 * (loss.backward()) up to the factor -1/N and the raw-parameter chain rule, which stay on the host. */
int bgp_lml_grad(bgp_handle* h, double* grad_out, int ngrad);

/* Opt-in: bgp_lml_grad first saves the factor storage into a second buffer of the same size (8 N^2 B in the full-square
 * layout, ~4 N (N + W) B in slabs; one device-to-device copy, ~2 x that / HBM bandwidth), and the next call that needs the
 * factor copies it back instead of re-running the N^3/3 fit: a prediction at the optimum after
 * train_hyperparameters (src/batt_models/battcellgp_full.py:127-166 followed by :168-195) costs no second
 * factorisation.  "When memory allows": if the second buffer can not be allocated the gradient proceeds without it and
 * the factor comes back by the re-run, exactly as with the switch off (default); the failed allocation is not attempted
 * again for a factor of that size until this function is called again.  on = 0 frees the buffer. */
int bgp_set_keep_factor(bgp_handle* h, int on);

/* PREDICT: posterior of the latent f at Xq[M,D] (no noise added):
 *   mean = K_*X alpha;  var = diag(K_**) - colsumsq(L^-1 K_X*), floored at min_var
 *   (pass min_var < 0 for the unclamped diagonal of src/gp/standard_models.py:48;
 *    1e-10 mirrors MultivariateNormal.variance at battcellgp_full.py:180).
 * var_out may be NULL (the `no_cov` path of battcellgp_full.py:177-178: mean only, no TRSM).
 * Replaces ExactGP.__call__ (eval) at battcellgp_full.py:171-180, standard_models.py:40-48. */
int bgp_predict(bgp_handle* h, const double* Xq_host, int64_t M, double* mean_out,
                double* var_out, double min_var);
int bgp_predict_dev(bgp_handle* h, const double* Xq_dev, int64_t M, double* mean_dev,
                    double* var_dev, double min_var);

/* Full posterior covariance [M, M] (row-major, symmetric) for
 * ScaledRBFModel.predict(full_cov=True), src/gp/standard_models.py:45-46. */
int bgp_predict_cov(bgp_handle* h, const double* Xq_host, int64_t M, double* mean_out,
                    double* cov_out);

/* Dense prior covariance K(X1, X2) -> out[n1, n2] row-major on the host, computed by the
 * same fill kernel as the fit (no noise).  X2_host == NULL means X2 = X1.  Replaces
 * kernel(x1, x2).to_dense() (WienerKernel.forward, src/gp/wiener_kernel.py:10-32). */
int bgp_kernel_matrix(bgp_handle* h, const double* X1_host, int64_t n1, const double* X2_host,
                      int64_t n2, int D, double* out_host);

/* Copy alpha = Sigma^-1 y (length N) of the last fit to the host. */
int bgp_get_alpha(bgp_handle* h, double* alpha_host);

/* On-device residual checks of the last fit (for sizes the CPU oracle cannot reach):
 *   out[0] = || Sigma alpha - y ||_2 / || y ||_2        (Sigma re-evaluated tile by tile)
 *   out[1] = max over `nsample` sampled entries of |(L L^T)_ij - Sigma_ij| / Sigma_ii-scale */
int bgp_residuals(bgp_handle* h, int nsample, double* out2);

/* Rows of the Cholesky factor of the last fit, for checks that must not trust the engine's own covariance
 * function: out_host[r * N + q] = L[rows[r], q] for q <= rows[r] (zero beyond), r < nrows, rows[r] < N.
 * The test side forms (L L^T)_ij = <row i, row j> and compares it with Sigma_ij evaluated by the CPU oracle.
 * Works for both layouts (full square / column slabs). */
int bgp_get_factor_rows(bgp_handle* h, const int64_t* rows, int nrows, double* out_host);

/* diag_host[i] = L_ii, i < N, of the last fit (log det Sigma = 2 sum log L_ii: lets a caller re-derive the LML
 * on the host from alpha and this vector). */
int bgp_get_factor_diag(bgp_handle* h, double* diag_host);

/* Per-phase timings of the most recent calls, see BGP_T_* (n <= BGP_T_COUNT values). */
int bgp_phase_times(const bgp_handle* h, double* out, int n);

/* Bytes of device memory currently owned by the handle. */
int64_t bgp_device_bytes(const bgp_handle* h);

/* ---- building blocks, exported for tests, the bench roofline and the sharded driver ---- */

/* In-place lower Cholesky of a device matrix A[n, n] (column-major, leading dimension lda,
 * lower triangle referenced).  info_out: 0 ok, k > 0 = leading minor k not positive.
 * n and lda must be multiples of 64 is NOT required: any n >= 1, lda >= n. */
int bgp_potrf_dev(bgp_handle* h, double* A_dev, int64_t n, int64_t lda, int* info_out);

/* C[m, n] -= A[m, k] * B[n, k]^T on device, all column-major; lower != 0 restricts the update
 * to tiles touching i >= j (SYRK-style trailing update).  k must be a multiple of 16. */
int bgp_gemm_nt_sub_dev(bgp_handle* h, double* C_dev, int64_t ldc, const double* A_dev,
                        int64_t lda, const double* B_dev, int64_t ldb, int64_t m, int64_t n,
                        int64_t k, int lower);

/* Fill out_dev[i + j*ld] = k(x1_i, x2_j) (+ diag_add on i == j when x2_dev == x1_dev) for the
 * handle's kernel; lower != 0 writes only tiles on or below the diagonal. */
int bgp_fill_dev(bgp_handle* h, const double* x1_dev, int64_t n1, const double* x2_dev,
                 int64_t n2, int D, double* out_dev, int64_t ld, int lower, double diag_add);


/* ---- building blocks of the sharded (column-panel block-cyclic) Cholesky: one process per GPU,
 * panels broadcast over RCCL by the host driver (battgp_amd/sharded.py).  A rank stores whole
 * column panels: all rows from the panel's diagonal down, plus the augmented block.  These calls
 * are ASYNCHRONOUS on the handle's stream unless stated; bgp_sync() drains it. ---- */

/* out[(i-row0) + (j-col0)*ld] = Sigma_ij = K(x_i, x_j) + (noise + extra_diag) [i == j] for
 * i in [row0, row0+nrows), j in [col0, col0+ncols); indices >= N are padding (identity).
 * The diagonal term is applied only when row0 == col0 (a panel that starts on the diagonal). */
int bgp_fill_block_dev(bgp_handle* h, const double* X_dev, int64_t N, int D, int64_t row0, int64_t col0,
                       int64_t nrows, int64_t ncols, double* out_dev, int64_t ld, double extra_diag);

/* Cross-covariance rows that RIDE through a sharded factorisation below the augmented block of a panel (the sharded form
 * of bgp_fit_predict's riding rows): out[i + (j-col0)*ld] = k(xq_i, x_j) for i < M and j < N, zero for the padding rows
 * i in [M, nrows) and padding columns j >= N;  j in [col0, col0+ncols).  No noise term.  After the panel's factorisation
 * and the updates by the earlier panels these rows hold (K_*X L^-T)[:, panel] - src/batt_models/battcellgp_full.py:171-173
 * evaluated without a separate triangular-solve pass. */
int bgp_cross_block_dev(bgp_handle* h, const double* Xq_dev, int64_t M, int64_t nrows, const double* X_dev, int64_t N, int D,
                        int64_t col0, int64_t ncols, double* out_dev, int64_t ld);

/* Augmented block under columns [col0, col0+ncols): aug[r + (j-col0)*ld] = (r == 0 && j < N) ? y[j] : 0,
 * r < 64.  After the factorisation row 0 holds z^T = (L^-1 y)^T for those columns. */
int bgp_aug_rows_dev(bgp_handle* h, const double* y_dev, int64_t N, int64_t col0, int64_t ncols,
                     double* aug_dev, int64_t ld);

/* Factor one column panel in place: panel_dev points at the panel's diagonal element, `nrows` rows
 * from there down (including augmented rows), `nbk` columns (multiple of 64, <= nb_outer).
 * inv_dev receives the nbk/64 inverted 64x64 diagonal tiles.  Synchronous; *info_out = 0 or the
 * 1-based index (within the panel) of the first non-positive pivot. */
int bgp_factor_panel_dev(bgp_handle* h, double* panel_dev, int64_t ld, int64_t nrows, int nbk,
                         double* inv_dev, int* info_out);

/* bgp_factor_panel_dev with the single-GPU engine's panel scheme 1 and the packing for the broadcast fused in:
 * the 64-wide chain runs on the nbk x nbk diagonal block only (riding identity -> L_kk^-1), the rows below
 * are solved by ONE MFMA GEMM straight into the packed buffer pack_dev [nrows, nbk] (ld = nrows), and
 * copied back into the panel.  nbk <= nb_outer. */
int bgp_factor_pack_panel_dev(bgp_handle* h, double* panel_dev, int64_t ld, int64_t nrows, int nbk, double* inv_dev,
                              double* pack_dev, int* info_out);

/* The asynchronous form used by the sharded driver: nothing is read back, the call only enqueues.  The handle's
 * device-side failure flag is NOT reset (bgp_flag_reset_dev does that once per factorisation attempt): a failed pivot -
 * reported as gofs + the 1-based index within the panel - turns every later kernel of this handle's pipeline into a
 * no-op.  After the panel the flag is copied into the 8-byte slot flag_slot_dev (the element behind the packed panel:
 * it travels with the broadcast), as an int in the low word. */
int bgp_factor_pack_panel_async_dev(bgp_handle* h, double* panel_dev, int64_t ld, int64_t nrows, int nbk, double* inv_dev,
                                    double* pack_dev, int64_t gofs, double* flag_slot_dev);

/* Failure flag of the handle's pipeline (sharded driver): reset it (asynchronous); merge a received panel's flag slot
 * into it (asynchronous: if the slot is non-zero and the flag is still clear, the flag takes its value); read it
 * (synchronous: drains the handle's streams). */
int bgp_flag_reset_dev(bgp_handle* h);
int bgp_flag_merge_dev(bgp_handle* h, const double* flag_slot_dev);
int bgp_flag_read(bgp_handle* h, int* flag_out);

/* The HIP stream the handle's asynchronous calls are enqueued on (which = 0: main, 1: auxiliary), as a hipStream_t.
 * A host framework that moves data for the engine (torch.distributed / RCCL in battgp_amd/sharded.py) wraps it
 * (torch.cuda.ExternalStream) so that its copies and collectives are ORDERED with the engine's kernels by the
 * stream itself - no host synchronisation between a broadcast and the kernels that produce / consume its buffer. */
void* bgp_get_stream(bgp_handle* h, int which);

/* E_K <- E_K L_KK^-T for a row block E_K[me, nbk] against a factored panel's diagonal block. */
int bgp_solve_panel_dev(bgp_handle* h, double* E_dev, int64_t lde, int64_t me, const double* Lkk_dev,
                        int64_t ld, int nbk, const double* inv_dev);

/* bgp_gemm_nt_sub_dev without the trailing synchronisation. */
int bgp_gemm_nt_sub_async_dev(bgp_handle* h, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                              const double* B_dev, int64_t ldb, int64_t m, int64_t n, int64_t k, int lower);

/* All rank-k updates of one step of the sharded factorisation in one call: for i < count
 *   C_i[rows_i, ncols_i] -= P[p_off_i .. , 0..k) P[p_off_i .. p_off_i + ncols_i, 0..k)^T    (lower trapezoid)
 * with C_i = store_dev + desc[5 i] (leading dimension desc[5 i + 4]), rows_i = desc[5 i + 1], ncols_i = desc[5 i + 2],
 * p_off_i = desc[5 i + 3] (desc on the host): every local panel keeps only the rows from its own diagonal down, so
 * each has its own leading dimension.  The launches alternate between the engine's two streams, so that the
 * partial last round of workgroups of one panel's update overlaps the next panel's; deep updates (k >= 256)
 * accumulate through L2 atomics like the single-GPU trailing update.  abort_flag_dev (may be NULL): an 8-byte flag slot
 * (see bgp_factor_pack_panel_async_dev) - non-zero turns the launches into no-ops.  Asynchronous (bgp_sync). */
int bgp_update_panels_dev(bgp_handle* h, double* store_dev, const int64_t* desc, int count, const double* P_dev,
                          int64_t ldp, int k, const double* abort_flag_dev);

/* ---- building blocks of the sharded LML gradient: Sigma^-1 formed IN PLACE over the distributed factor
 * (battgp_amd/sharded.py::ShardedExactGP.lml_grad; the single-GPU bgp_lml_grad runs the same steps on its slabs).
 * All asynchronous on the handle's stream. ---- */

/* General form of the MFMA product  C[m, n] (op)= A[m, k] B[n, k]^T  (all column-major; k a multiple of 16; m, n, lda,
 * ldb even; A, B 16-byte aligned):  mode 0  C -= A B^T;  mode 1  C = A B^T (C must not overlap A or B unless n <= 64);
 * mode 2  C -= A B^T accumulated by one L2 atomic per element (no C read; deep k).  (A `C += A B^T` is mode 2 / 0 against
 * a negated copy of B: bgp_block_copy_dev with scale -1.)
 * lower != 0: only tiles touching i >= j.  btri != 0 (mode 1): B is lower triangular (B[j, kk] = 0 for kk > j) and
 * the zero half of the k-range is skipped. */
int bgp_gemm_nt_async_dev(bgp_handle* h, int mode, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                          const double* B_dev, int64_t ldb, int64_t m, int64_t n, int64_t k, int lower, int btri);

/* dst = scale * op(src[rows, cols]) (column-major): trans != 0 writes the transposed block dst[c + r ldd], else
 * dst[r + c ldd]; tri != 0 takes the entries above the diagonal of src (r < c) as zero.  src == dst is allowed for
 * trans == 0. */
int bgp_block_copy_dev(bgp_handle* h, const double* src_dev, int64_t lds, int64_t rows, int64_t cols, double* dst_dev,
                       int64_t ldd, int trans, double scale, int tri);

/* out_dev[nbk, nbk] (ld = nbk, clean lower triangle, zeros above) = inv(L_kk) of a factored panel's diagonal block
 * (panel_dev, ld) from its nbk/64 inverted diagonal tiles inv_dev (what bgp_factor_pack_panel_async_dev left). */
int bgp_panel_inverse_dev(bgp_handle* h, const double* panel_dev, int64_t ld, int nbk, const double* inv_dev, double* out_dev);

/* out_dev[c] = sum_{r < rows} A[r + c ld] x[r],  c < ncols (a multiple of 64): A^T x of a tall column block. */
int bgp_gemv_t_dev(bgp_handle* h, const double* A_dev, int64_t ld, int64_t rows, int ncols, const double* x_dev, double* out_dev);

/* Number of accumulators of the gradient reduction, and the reduction over ONE lower trapezoid of P = Sigma^-1 that
 * starts on the diagonal: rows [r0, r0 + nrows) x columns [r0, r0 + ncols) of the matrix, element (i, j) at
 * P_dev[(i - r0) + (j - r0) ldp] (a column panel of the sharded store).  acc_dev[0 .. nacc) is increased by
 * sum' (alpha_i alpha_j - P_ij) dSigma_ij/d(.) of that block (accumulate == 0: overwritten).  X_dev [N, D] row-major,
 * alpha_dev [>= min(N, r0 + nrows)].  1 <= ncols <= nrows <= ldp; the block need not be a whole number of the kernel's
 * 512 x 32 tiles - nothing outside the described rows and columns is read.  bgp_grad_finish turns the (all-reduced) accumulators, copied to the host, into
 * d lml / d theta in the layout of hyp - the same arithmetic as bgp_lml_grad. */
int bgp_grad_nacc(void);
int bgp_grad_reduce_block_dev(bgp_handle* h, const double* X_dev, int64_t N, int D, int64_t r0, int64_t nrows, int64_t ncols,
                              const double* P_dev, int64_t ldp, const double* alpha_dev, double* acc_dev, int accumulate);
int bgp_grad_finish(bgp_handle* h, const double* acc_host, int D, double* grad_out, int ngrad);

/* out_host[0] = sum_{i<n} log A[i + i*ld] (half log-determinant of a factored diagonal block). Synchronous. */
int bgp_diag_logsum_dev(bgp_handle* h, const double* A_dev, int64_t ld, int64_t n, double* out_host);

/* out_dev[m] = sum_{i<n} E[m + i*lde] * (vec_dev ? vec_dev[i] : E[m + i*lde]),  m < M. */
int bgp_rowdot_dev(bgp_handle* h, const double* E_dev, int64_t lde, int64_t M, int64_t n, const double* vec_dev,
                   double* out_dev);

/* out_dev[m] = max(k(xq_m, xq_m) - ssq_dev[m], min_var) (no floor if min_var < 0): predictive variance from
 * the row sums of squares of V^T, with the prior variance of the handle's kernel
 * (K0: s_w t^3/3 + s_r, the diag branch of src/gp/wiener_kernel.py:15-16). */
int bgp_var_finish_dev(bgp_handle* h, const double* Xq_dev, int64_t M, int D, const double* ssq_dev, double min_var,
                       double* out_dev);

/* Diagnostic: one wavefront on the handle's AUXILIARY stream records nsamp pairs (wall clock [100 MHz ticks],
 * shader clock [cycles]) into out_dev[2 * nsamp] (uint64), spinning `spin` FMAs between samples - launched next to
 * work on the main stream it shows the shader clock the SMU grants WHILE that work runs (tools/fill_gap_probe.py).
 * Asynchronous (bgp_sync). */
int bgp_debug_clock_samples_dev(bgp_handle* h, uint64_t* out_dev, int nsamp, int spin);

/* Diagnostic / tests: `extra_rows` unused rows are appended to every column of the factor buffer from the next fit on
 * (leading dimension N + 64 + riding rows + extra_rows; even, >= 0; 0 restores the default).  Small problems then address
 * the buffer with the element strides of the BASELINE sizes (row + col * ld beyond 2^31 / 2^32), which is how the CPU
 * suite checks the index arithmetic of the whole single-GPU path without an N > 46 341 problem
 * (tests/emu/huge_ld_check.py).  Frees the resident problem; a pooled handle that is revived starts at 0 again. */
int bgp_debug_set_ld_pad(bgp_handle* h, int64_t extra_rows);

/* Wait for everything enqueued on the handle's streams. */
int bgp_sync(bgp_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* BATTGP_H */
