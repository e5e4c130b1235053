/* hdpgpc_hip.h - C-ABI of libhdpgpc_hip.so: the GP-emission hot path of HDP-GPC on MI355X (gfx950).
 *
 * The reference (AdrianPerezHerrero/HDP-GPC) is pure Python and has no FFI layer; every entry point
 * below replaces the NumPy/SciPy/torch-CPU arithmetic of one Python method (cited per function, paths
 * relative to the reference's hdpgpc/hdpgpc/).  INTEGRATION.md shows the ctypes binding a maintainer
 * would add on the reference side.
 *
 * Conventions
 *  - every `const double*` / `double*` / `int32_t*` argument is a DEVICE pointer owned by the caller unless
 *    its name ends in `_host`; nothing is allocated, freed or synchronised inside a call;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - return value: 0 = work enqueued; -1 = bad argument; -2 = size not supported by this build;
 *    >= 1000 = 1000 + hipError_t of a failed launch;
 *  - numerical failure is reported per item in an `info` array (LAPACK convention: 0 = ok, j > 0 = the
 *    j-th pivot was not positive), so the Python layer can raise torch.linalg.LinAlgError exactly where
 *    torch.linalg.cholesky would;
 *  - all arithmetic is fp64; matrices are row-major and dense.
 */
#ifndef HDPGPC_HIP_H
#define HDPGPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HGP_ABI_VERSION 6   /* 6: + hgp_chol_inverse_ws_f64, candidate / no-smoother flags of the chain finish; 5: + hgp_pairs_plan_set_score_output, hgp_debug_exp_neg_f64; 3: + hgp_pairs_plan_set_accuracy (solve-based per-pair path), assignment tail, warp fit, member-step lists; 4: + batched chain gather / finish */
/* largest T (basis length) and T* (segment length) served by the register-resident wave kernels */
#define HGP_MAX_T_WAVE 128
/* largest T served at all: 128 < T <= 256 runs on cooperative kernels (one workgroup of 4-8 waves per matrix / pair) */
#define HGP_MAX_T_COOP 256

int hgp_abi_version(void);

/* diagnostics: C[16,16] = A[16,16] B[16,16] by one wave through v_mfma_f64_16x16x4_f64, using the operand
 * and accumulator lane maps the kernels assume (tests/test_gpu_parity.py checks it with asymmetric data). */
int hgp_debug_mfma_f64(const double* A, const double* B, double* C, void* stream);

/* diagnostics: out[i] = exp(-h[i]) through the kernels' own exponential (exp_neg4, tile_f64.hpp: the RBF entries of the
 * per-pair kernels' E and K** are built with it); n a multiple of 4, h >= 0 (tests/test_gpu_parity.py: <= 3e-16 relative). */
int hgp_debug_exp_neg_f64(const double* h, int n, double* out, void* stream);

/* a1 - (ConstantKernel(c) * RBF(ell) + WhiteKernel(noise))(X, Y)   [scikit-learn kernels.py; built at
 * GPI_HDP.py:164-166, called at GPI.py:54-58,124,126,474-476].  y == NULL is the one-argument call
 * (white noise on the diagonal, exact c on the diagonal); otherwise no white noise.  K_out: [nx, ny]. */
int hgp_gram_rbf_f64(const double* x, int nx, const double* y, int ny, double c, double ell, double noise,
                     double* K_out, void* stream);

/* a3 - GPI_model._chol_spd (GPI_model.py:83-87), batched: for each of the b matrices A[T,T] (in place)
 *   L = chol(0.5 (A + A^T) + (add_diag + jitter_rel * max(mean_i |A_ii + add_diag|, eps)) I), lower, zeros above.
 * Optional outputs (may be NULL): Linv[b,T,T] = L^{-1}; logdet[b] = log det of the regularised matrix. */
int hgp_potrf_batched_f64(double* A, int T, int b, double jitter_rel, double add_diag, double* Linv,
                          double* logdet, int32_t* info, void* stream);

/* The inverse of the a3 factor only (A is not modified): Linv[b,T,T] = chol(0.5 (A + A^T) + shift I)^{-1}, T <= 256
 * (one wavefront per block column for T <= 128, one workgroup per block column above).
 * Replaces torch.linalg.solve / inv / cholesky_solve on symmetric positive-definite matrices in the state recursion
 * (GPI.py:144-145,267,295; GPI_model.py:1316,1330): S^{-1} = Linv^T Linv. */
int hgp_chol_inverse_batched_f64(const double* A, int T, int b, double jitter_rel, double add_diag, double* Linv,
                                 int32_t* info, void* stream);
/* The same with a caller-provided workspace work[b,T,T] (used for T > 128 only; NULL = none): large batches factor every matrix
 * ONCE into the workspace and take L^-1 from L by block columns instead of one factorisation per block column. */
int hgp_chol_inverse_ws_f64(const double* A, int T, int b, double jitter_rel, double add_diag, double* Linv, double* work,
                            int32_t* info, void* stream);

/* a4 + a6 - GPI_model._gaussian_score_shared_cov (GPI_model.py:92-113) over the groups that
 * GPI_model.compute_sq_err_all builds on a shared grid (GPI_model.py:516-533).
 * Work item g (one per group chunk) scores `item_cnt[g]` segments against one state:
 *     S    = Sigma + item_mat[g] * sigma_stride          (T x T, leading dimension T)
 *     m    = mean  + (item_mean ? item_mean[g] : item_mat[g]) * mean_stride   (T)   (mean == NULL: zero mean)
 *     cov  = 0.5 (S + S^T) + item_add[g] I               ("first" inflation, GPI_model.py:527-529)
 *     cov += jitter_rel * max(mean |diag cov|, eps) I    (a3)
 *   for j < item_cnt[g]:  n = seg_ids ? seg_ids[item_off[g] + j] : item_off[g] + j
 *     out_quad[n]   = (Y[n] - m)^T cov^{-1} (Y[n] - m)
 *     out_logdet[n] = log det cov          (may be NULL; the reference's score omits it, GPI_model.py:113)
 *     out_info[n]   = LAPACK info of cov   (may be NULL)
 * The reference's score is -0.5 * out_quad - 0.5 * T * log(2 pi). */
int hgp_score_groups_f64(const double* Y, int ldy, const double* mean, long mean_stride, const double* Sigma,
                         long sigma_stride, int T, const int32_t* item_mat, const int32_t* item_mean,
                         const double* item_add, const int32_t* item_off, const int32_t* item_cnt, int n_items,
                         const int32_t* seg_ids, double jitter_rel, double* out_quad, double* out_logdet,
                         int32_t* out_info, void* stream);

/* a6, member segments - GPI_model.compute_sq_err_all gives every member segment of a cluster its own LDS step
 * (GPI_model.py:508-531): one Sigma_i, one right-hand side.  Same arithmetic as hgp_score_groups_f64 with one
 * segment per item, on a leaner kernel (T <= 128):
 *   cov_i = 0.5 (S + S^T) + seg_add[i] I + jitter_rel max(mean|diag|, eps) I,  S = Sigma + seg_mat[i] * sigma_stride
 *   out_quad[i] = (Y[i] - m)^T cov_i^{-1} (Y[i] - m),   m = mean + (seg_mean ? seg_mean[i] : seg_mat[i]) * mean_stride
 * symmetric != 0: the caller guarantees every Sigma to equal its transpose bit for bit (true for the MNIW scale
 * recursion, GPI_model.py:1332-1336); only the upper triangle is then read (about half the HBM bytes). */
int hgp_score_each_f64(const double* Y, int ldy, const double* mean, long mean_stride, const double* Sigma,
                       long sigma_stride, int T, const int32_t* seg_mat, const int32_t* seg_mean, const double* seg_add,
                       int n, double jitter_rel, int symmetric, double* out_quad, double* out_logdet, int32_t* out_info,
                       void* stream);

/* a2 + a5 - the per-(segment, cluster) general path: IterativeGaussianProcess.pred_dist (GPI.py:457-503)
 * followed by the score of GPI_model.log_sq_error (GPI_model.py:250-286), for an N x K batch.
 *
 * A plan holds what depends only on the clusters.  theta_host[K,3] = (c, ell, noise) per cluster lives on
 * the HOST (the reference keeps it in scikit-learn kernel objects); Ts_max is the longest segment the plan will
 * be asked to score; dev_buf is caller-owned device scratch of at least
 * hgp_pairs_plan_device_bytes(T, Ts_max, K) bytes and must outlive the plan. */
typedef struct hgp_pairs_plan hgp_pairs_plan;
size_t hgp_pairs_plan_device_bytes(int T, int Ts_max, int K);
int hgp_pairs_plan_create(hgp_pairs_plan** plan, int T, int Ts_max, int K, const double* theta_host, void* dev_buf,
                          size_t dev_bytes);
void hgp_pairs_plan_destroy(hgp_pairs_plan* plan);
/* (Re)compute the per-cluster operators from the current state: x_basis[T], mean[K,T] (= C f, the prior mean
 * on the basis grid), Sigma[K,T,T].  Per cluster: K~ = ker(xb,xb) + 1e-4 max(mean|diag Sigma|, eps) I
 * (GPI.py:474,488-489), its Cholesky inverse, and M = c^2 (K~^{-1} Sigma K~^{-1} - K~^{-1}) so that
 * cov_f = K** + E^T M E  (GPI.py:500) with E = exp(-0.5 ((xb_i - x_j)/ell)^2).  info[K]: Cholesky status. */
int hgp_pairs_plan_update(hgp_pairs_plan* plan, const double* x_basis, const double* mean, const double* Sigma,
                          int32_t* info, void* stream);
/* Device pointer to the per-cluster scalars [K,8] written by hgp_pairs_plan_update: c, ell, noise, iso flag
 * (Sigma iso-diagonal, GPI.py:497), mean(diag Sigma), jitter of K~, kinv = ||K~^{-1}||_inf, 0.
 * The per-pair kernel evaluates cov_f through the explicit operator M; its rounding error relative to the
 * reference's triangular solves grows like eps * (c * kinv)^2 (about 1e-10 for the reference's length-scale
 * 1.2 on a unit-spaced grid, where parity is 1e-11; quickly worse for smoother kernels); hgp_pairs_plan_update compares it
 * with the tolerance of hgp_pairs_plan_set_accuracy ON THE DEVICE and routes such clusters to the solve-based kernel
 * (scalar 7 = the flag): no caller has to check anything. */
const double* hgp_pairs_plan_scalars(const hgp_pairs_plan* plan);
/* Which clusters take the solve-based evaluation (the reference's operation order, GPI.py:489-501: S = cholesky_solve(K*, L)
 * by blocked substitution per pair, cov_f = K** + S^T (Sigma S - K*)) instead of the explicit operator M:
 *   tol > 0 : the clusters with eps * (c * kinv)^2 > tol   (default 1e-9; decided on the device inside hgp_pairs_plan_update)
 *   tol == 0: every cluster;   tol < 0: none (explicit operator everywhere).
 * Takes effect at the next hgp_pairs_plan_update. */
int hgp_pairs_plan_set_accuracy(hgp_pairs_plan* plan, double tol);
/* What hgp_loglik_pairs_f64 stores in out_quad:  on == 0 (default): d^T cov^{-1} d;  on != 0: the reference's score itself,
 * -0.5 d^T cov^{-1} d - 0.5 Ts log(2 pi)  (GPI_model.py:285, no log-determinant) - written by the pair kernels, so that the caller
 * needs no arithmetic of its own behind the launch.  Takes effect at the next hgp_loglik_pairs_f64. */
int hgp_pairs_plan_set_score_output(hgp_pairs_plan* plan, int on);
/* x[N,Ts], y[N,Ts]: segment grids and values.  first_noise[N,K] (may be NULL): additive diagonal of the
 * `first` branch (GPI_model.py:271-273).  Outputs [N,K]: out_quad = d^T cov^{-1} d, out_logdet (may be NULL),
 * out_info (may be NULL).  cov carries the reference's regularisation: +1e-6 I (GPI.py:501, dense Sigma only),
 * + first_noise, + 1e-8 mean|diag| I (GPI_model.py:83-87). */
/* sel[N] (may be NULL): segment n is scored against cluster sel[n] only - the per-segment LDS step of
 * GPI_model.compute_sq_err_all's irregular-grid loop (GPI_model.py:535-545); outputs and first_noise are then [N].
 * A plan serves one stream at a time (its workspace and, for Ts > 128, the overflow areas handed out inside the kernel belong to
 * the call; the hand-out flags are reset at the start of every call).  The fall-back list of segments whose E is not
 * block-tridiagonal and its two counters are plan-owned scratch as well: two calls on one plan must be ordered on one stream. */
int hgp_loglik_pairs_f64(const hgp_pairs_plan* plan, const double* x, const double* y, int N, int Ts,
                         const double* first_noise, const int32_t* sel, double* out_quad, double* out_logdet,
                         int32_t* out_info, void* stream);

/* Auxiliary: C[b] = alpha op(A[b]) op(B[b]) + beta C[b] (row-major, any M x N x Kd) on v_mfma_f64_16x16x4_f64.
 * Replaces the torch.matmul / torch.linalg.multi_dot calls of the matrix-valued terms below. */
int hgp_gemm_batched_f64(int transA, int transB, int M, int N, int Kd, double alpha, const double* A, int lda, long strideA,
                         const double* B, int ldb, long strideB, double beta, double* C, int ldc, long strideC, int batch,
                         void* stream);

/* workspace (bytes) of the two matrix-valued likelihood terms below for b items of size T; only read for 128 < T <= 256
 * (composition of the batched kernels) - for T <= 128 each term is ONE fused kernel, one wavefront per item, and ws may
 * be NULL */
size_t hgp_matrix_lik_ws_bytes(int T, int b);

/* a8 - GPI_model.log_lat_error (GPI_model.py:288-323), batched over b LDS steps:
 *   r = f_cur - A f_prev;  L = _chol_spd(Gamma);  out = -0.5 (r^T Gamma^{-1} r + tr(A^T Gamma^{-1} A covprev))
 * (the caller adds -0.5 T log 2pi).  f_cur, f_prev [b,T]; A, Gamma, covprev [b,T,T]. */
int hgp_lat_error_f64(const double* f_cur, const double* f_prev, const double* A, const double* Gamma, const double* covprev,
                      int T, int b, double* out, int32_t* info, void* ws, size_t ws_bytes, void* stream);

/* a9 - matrix_normal_inv_wishart.log_likelihood_MNIW (GPI_model.py:1346-1362), batched:
 *   L = chol(0.5 (Sigma + Sigma^T) + 1e-8 I);  D = M - m_mean
 *   out = -0.5 sum (D m_r_cov) o (Sigma^{-1} D) - 0.5 tr(Sigma^{-1} scale)
 * M, Sigma [b,T,T]; the prior (m_mean, m_r_cov, scale) is read with stride prior_stride (0 = shared by all items);
 * m_r_cov == NULL means the identity (the only value on the hot path, GPI_model.py:481-484); scale_is_diagonal != 0 promises
 * a diagonal `scale` (the prior's sigma I on the hot path): tr(Sigma^-1 scale) is then taken from the columns of L^-1 alone. */
int hgp_mniw_loglik_f64(const double* M, const double* Sigma, const double* m_mean, const double* m_r_cov,
                        const double* scale, int scale_is_diagonal, long prior_stride, int T, int b, double* out, int32_t* info,
                        void* ws, size_t ws_bytes, void* stream);

/* a11 - WarpPriorAMTGP._rbf_cov (amtgp_warping_system.py:160-173): omega^2 exp(-0.5 dx^2/rho^2) + diag_add I on the
 * grid normalised to [0,1] when normalize != 0.  The batch score is then hgp_score_groups_f64 with jitter_rel = 0. */
int hgp_warp_cov_f64(const double* x, int T, double rho, double omega, double diag_add, int normalize, double* K_out,
                     void* stream);

/* BASELINE configs[4] - rank-1 update of a Cholesky factor, batched, T <= 256, in place:
 *   L <- chol(alpha[b] L L^T + beta[b] v v^T)   (alpha, beta may be NULL = 1).
 * The MNIW scale recursion new_scale = a scale + b e e^T (GPI_model.py:1332-1336) has exactly this form. */
int hgp_chol_rank1_f64(double* L, const double* v, const double* alpha, const double* beta, int T, int b, int32_t* info,
                       void* stream);

/* 8f-1 (first "next" row) - glue of one member step of the LDS recursion (GPI_model.full_pass_weighted,
 * GPI_model.py:377-406), fused so that a captured step is GEMMs + inverses + two of these launches.
 * hgp_lds_chain_gather_f64: row pos[0] of the state stacks A, Gamma, C, Sigma, cov_f_sm, cov_f ([L,T,T]) and f_star,
 *   f_star_sm ([L,T]) into out[6 T T + 2 T] in that order (replaces the per-step list indexing of GPI_model.py:300-318);
 *   if Y != NULL also y_out[T] = Y[pos[0] - y_row0] (the observation of the member this step includes).
 * hgp_lds_chain_finish_f64: element-wise tail of the two matrix_normal_inv_wishart.posterior updates
 *   (GPI_model.py:1326-1336; item 0 = internal (A, Gamma), item 1 = observation (C, Sigma)):
 *     bad = any(info1, info2 != 0)                      (then the previous distributions are kept, GPI_model.py:1068-1071)
 *     means' = ((n0 - 2) means + part) / (n0 - 1);  R' = Snew;  scales' = ((n0 - 2) scales + ee) / (n0 - 1)
 *     n0' = n0 + 1 (unless bad);  Nf' = Nf + 1;  scl = n0' / (n0' - 2);  ann = annealing ? 1 / Nf'^2 : 0
 *     A[pos+1] = means'[0]; C[pos+1] = means'[1]; Gamma[pos+1] = scales'[0] scl + Gamma[0] ann; Sigma likewise
 *                                                        (bayesian_new_params, GPI_model.py:1076-1106)
 *     W = (means', R', scales') [3,2,T,T];  n0, Nf, bad_count[2], pos updated in place (pos += 1).
 *     bad_count[0] counts the steps that kept their previous distributions; info0[2] (may be NULL) is the status of the
 *     step's Kalman / pair-smoother factorisations: the first step (row index) where one failed is latched in bad_count[1].
 *     sync: one int32 the caller zero-initialises once (inter-block counter, left at zero). */
int hgp_lds_chain_gather_f64(const double* stA, const double* stG, const double* stC, const double* stS, const double* stPsm,
                             const double* stP, const double* stF, const double* stFsm, const int64_t* pos, int T,
                             double* out, const double* Y, long y_row0, double* y_out, void* stream);
/* hgp_rts_chain_f64: the sequential part of GPI.backward (GPI.py:240-270) for all n states in one launch, T <= 96
 * (-2 above).  J[n-1,T,T] = c_t A_t^T P_t^{-1}, P[n-1,T,T] = A_t c_t A_t^T + Gamma_t and AM[n-1,T] = A_t m_t come from
 * the FILTERED states (batched by the caller); in place, for t = n-2 .. 0:
 *     M[t] += J[t] (M[t+1] - AM[t]);   Cv[t] += J[t] (Cv[t+1] - P[t]) J[t]^T. */
int hgp_rts_chain_f64(const double* J, const double* P, const double* AM, double* M, double* Cv, int n, int T, void* stream);
/* hgp_lds_chain_scatter_f64: f_star[pos+1] = f_star_sm[pos+1] = f_post; cov_f[pos+1] = cov_f_sm[pos+1] = c_post
 *   (include_sample, GPI_model.py:317); f_star_sm[pos] = f_sm_prev, cov_f_sm[pos] = P_sm_prev (backwards_pair,
 *   GPI_model.py:705-716). */
int hgp_lds_chain_scatter_f64(const double* f_post, const double* c_post, const double* f_sm_prev, const double* P_sm_prev,
                              double* stF, double* stFsm, double* stP, double* stPsm, const int64_t* pos, int T, void* stream);
/* out[b] = R[b] + factor * max(mean |diag S[b]|, eps) I  - the jitter matrix_normal_inv_wishart.posterior adds to the
 * right covariance before inverting it (GPI_model.py:1312-1316). */
int hgp_add_diag_mean_f64(const double* R, const double* S, int T, int b, double factor, double* out, void* stream);
/* C[b] = alpha op(A[b]) op(B[b]) + beta D[b]  (D: leading dimension ldd, batch stride strideD, 0 = shared). */
int hgp_gemm_add_batched_f64(int transA, int transB, int M, int N, int Kd, double alpha, const double* A, int lda, long strideA,
                             const double* B, int ldb, long strideB, double beta, const double* D, int ldd, long strideD,
                             double* C, int ldc, long strideC, int batch, void* stream);
int hgp_lds_chain_finish_f64(int T, const double* part, const double* ee, const double* Snew, const int32_t* info1,
                             const int32_t* info2, const int32_t* info0, double* W, double* n0, double* Nf, int32_t* bad_count,
                             double* stA, double* stG, double* stC, double* stS, int64_t* pos, int annealing, int32_t* sync,
                             void* stream);

/* 8f-1, one launch per dependency level of the member step.  hgp_gemm_list_f64 executes a DEVICE-resident list of
 * heterogeneous products  C = alpha op(A) op(B) + beta D (+ add_eye on the diagonal), any M x N x K, vectors as N = 1;
 * C2 (may be NULL) receives a second copy of the result.  total_tiles = sum over the items of ceil(M/16) ceil(N/16).
 * A step's lists are built once (every pointer is fixed for the life of a chain) and replayed from a hipGraph. */
typedef struct hgp_gemm_item {
  const double* A;
  const double* B;
  const double* D;
  double* C;
  double* C2;
  int M, N, K, lda, ldb, ldc, ldd, tA, tB;
  double alpha, beta, add_eye;
} hgp_gemm_item;
int hgp_gemm_list_f64(const hgp_gemm_item* items_dev, int n_items, int total_tiles, void* stream);
/* The same with a tile map in device memory, tile_map[t] = (item << 16) | tile-within-item for every output tile of the list in
 * order: long lists (the levels of many chains side by side) without the per-wave walk over the items.  A prefix of the map
 * (total_tiles = tiles of the first k items) runs the first k items. */
int hgp_gemm_list_mapped_f64(const hgp_gemm_item* items_dev, int n_items, const uint32_t* tile_map_dev, int total_tiles, void* stream);
/* Cholesky inverse with the right-hand sides riding the factorisation (T <= 128): for every matrix of the batch
 *   L = chol(0.5 (A + A^T) + shift I);  Linv = L^-1 (may be NULL);  rhs_out = L^-1 op(rhs)  (rhs NULL: none; rhs_on[m] == 0: not for m)
 * so that  B^T A^-1 = rhs_out^T Linv  costs ONE product after the factorisation (GPI.py:144-145,295; GPI_model.py:1329-1330). */
int hgp_chol_inverse_rhs_batched_f64(const double* A, int T, int b, double jitter_rel, double add_diag, double* Linv, const double* rhs,
                                     const int32_t* rhs_on, int rhs_trans, double* rhs_out, int32_t* info, void* stream);
/* hgp_lds_chain_gather_f64 plus the jittered right covariances of the two MNIW updates, Rp[2,T,T] = W[1] + 1e-2 max(mean|diag W[2]|, eps) I;
 * hgp_lds_chain_scatter_f64 + hgp_lds_chain_finish_f64 in one launch, with (y1 - y2)(y1 - y2)^T formed inside; info1[4] = status of
 * the first inversion (P, S_k, R0', R1'), info2[2] of the second. */
int hgp_lds_chain_gather2_f64(const double* stA, const double* stG, const double* stC, const double* stS, const double* stPsm,
                              const double* stP, const double* stF, const double* stFsm, const int64_t* pos, int T, double* out,
                              const double* Y, long y_row0, double* y_out, const double* W, double* Rp, void* stream);
int hgp_lds_chain_finish2_f64(int T, const double* f_post, const double* c_post, const double* f_sm_prev, const double* P_sm_prev,
                              const double* y, const double* part, const double* Snew, const int32_t* info1, const int32_t* info2,
                              double* W, double* n0, double* Nf, int32_t* bad_count, double* stA, double* stG, double* stC, double* stS,
                              double* stF, double* stFsm, double* stP, double* stPsm, int64_t* pos, int annealing, int32_t* sync,
                              void* stream);
/* The same two launches for a BATCH of independent chains (clusters x leads x proposals of the variational loop: the recursion
 * is sequential per chain, so throughput comes from running many chains side by side): one descriptor per chain in DEVICE
 * memory, blockIdx.y = chain.  Field meaning as the arguments of the single-chain calls above. */
typedef struct hgp_chain_gather_desc {
  const double* st[8];   /* stacks A, G, C, S, Psm, P ([L,T,T]), F, Fsm ([L,T]) */
  const int64_t* pos;
  double* out;           /* [6 T T + 2 T] */
  const double* Y;       /* observations of the run, row pos - y_row0 is read (y_row0 < 0: row 0, Y is the observation) */
  double* y_out;
  const double* W;       /* [3,2,T,T] means, right covariances, scales of the two MNIW distributions */
  double* Rp;            /* [2,T,T] */
  long y_row0;
  int T;
} hgp_chain_gather_desc;
typedef struct hgp_chain_finish_desc {
  const double* f_post; const double* c_post; const double* f_sm_prev; const double* P_sm_prev; const double* y;
  const double* part; const double* Snew;
  const int32_t* info1; const int32_t* info2;
  double* W; double* n0; double* Nf;
  int32_t* bad_count;
  double* stA; double* stG; double* stC; double* stS; double* stF; double* stFsm; double* stP; double* stPsm;
  int64_t* pos;
  int32_t* sync;
  int T, annealing;      /* annealing: bit 0 = annealed scales (GPI_model.py:1083-1091); bit 1 = candidate step (new rows are
                          * written at pos + 1, W / n0 / Nf / pos unchanged); bit 2 = previous smoothed state not rewritten */
} hgp_chain_finish_desc;
/* A device-resident list of plain copies dst[0..n) = src[0..n) run as ONE launch (the online step gathers the inputs of its
 * batched a8 / a9 calls from rows of many clusters' stacks).  max_n = the largest n of the list. */
typedef struct hgp_copy_item { const double* src; double* dst; long n; } hgp_copy_item;
int hgp_copy_list_f64(const hgp_copy_item* items_dev, int n_items, long max_n, void* stream);
int hgp_lds_chain_gather2_batched_f64(const hgp_chain_gather_desc* descs_dev, int n_chains, int T, void* stream);
int hgp_lds_chain_finish2_batched_f64(const hgp_chain_finish_desc* descs_dev, int n_chains, int T, void* stream);
/* a10 helper - || G^{-1} y ||^2 for the lower triangle G of a [T, ld] matrix.  IterativeGaussianProcess.
 * log_marginal_likelihood as written passes K itself as the "factor" to cho_solve (GPI.py:1043); this reproduces
 * that call with G = tril(K).  out[1]. */
int hgp_trsv_lower_quad_f64(const double* G, int ld, const double* y, int T, double* out, void* stream);
/* SURVEY 8f-3 - messages of the switching variable on the device (GPI_HDP.forward / backward / coupled_state_coef,
 * GPI_HDP.py:3546-3700), full recursion.  q[N,K]: log-observations (the [N,K] score matrix after LogLik); log_pi[K] =
 * compute_trans_pi, log_trans[K,K] = compute_trans_A (both stay host-side control plane).  Outputs: fmsg[N,K],
 * marg[N] (forward messages and their normalisers), bmsg[N,K] (backward messages, normalised without the last state as
 * in GPI_HDP.py:3645), log_resp_pair[N,K,K] (may be NULL; row 0 = -inf as in the reference).  K <= 64 (-2 above). */
int hgp_hmm_messages_f64(const double* q, const double* log_pi, const double* log_trans, int N, int K, double* fmsg,
                         double* marg, double* bmsg, double* log_resp_pair, void* stream);
/* The whole local step of the switching variable for a BATCH of B score matrices q[B,N,K] that share log_pi / log_trans (the
 * online step's candidates, GPI_HDP.py:586-630 once per candidate): LogLik normalisation per matrix -> qnorm[B,N,K], messages
 * -> fmsg / bmsg [B,N,K], marg [B,N], then the hard assignment: labels[B,N] = first arg-max of log(fmsg * bmsg) per row,
 * pair_first[B,N] (may be NULL) = first arg-max of the K x K pair table of row n (0 for row 0), last_log[B,K] (may be NULL)
 * = log(fmsg * bmsg) of the last row.  The [N,K,K] pair table is never written.  K <= 64. */
int hgp_hmm_local_terms_f64(const double* q, const double* log_pi, const double* log_trans, int N, int K, int B, double* qnorm,
                            double* fmsg, double* marg, double* bmsg, int64_t* labels, int64_t* pair_first, double* last_log,
                            void* stream);
/* SURVEY 8f-3, assignment tail.  hgp_loglik_rows_f64 = GPI_HDP.LogLik(axis=1) (GPI_HDP.py:632-661): out[n,:] = q[n,:] -
 * max_k q[n,k], rowmax[n] (may be NULL) = that maximum; if any row maximum is infinite the input is returned unchanged, as the
 * reference does.  hgp_assign_f64 = GPI_HDP._safe_exp (the one-hot arg-max, GPI_HDP.py:338-343) of log(fmsg * bmsg):
 * labels[N] (int64, may be NULL) = first arg-max per row, resp[N,K] (may be NULL) = its one-hot row. */
int hgp_loglik_rows_f64(const double* q, int N, int K, double* out, double* rowmax, void* stream);
int hgp_assign_f64(const double* fmsg, const double* bmsg, int N, int K, int64_t* labels, double* resp, void* stream);
/* SURVEY 8f-4 - Warping_system.compute_warp_batch (amtgp_warping_system.py:548-735): B independent monotone time-warps
 * g_b(t) of the observations Yt[b] onto the model mean Ym, each parameterised by n_ctrl control values (softplus increments,
 * cumulative sum, normalised to the grid's range) and fitted by `iters` Adam steps (lr, torch defaults) on
 *   0.5 |Yt[b](g_b) - Ym|^2 / (noise + 1e-12) + lam_s |D2 (g_b - x)|^2 + lam_a |g_b - x|^2,
 * weighted by weights[b] / sum(weights) (NULL = uniform).  One wavefront per sample, the whole optimisation in one launch.
 * x[T] strictly increasing (T <= 256), Yt[B,T,D], Ym with batch stride ym_stride (0 = one mean shared by the batch), D <= 4,
 * n_ctrl <= 32; u0[n_ctrl] (may be NULL = zeros) is the warm start of every sample.  Outputs: u_out[B,n_ctrl], xw_out[B,T] =
 * g_b - x, yw_out[B,T,D] = Yt[b](g_b), loss_out[B,iters,4] (may be NULL) = per-sample loss, data, smoothness and amplitude terms
 * at every iteration.  The warp-prior score of the result is hgp_warp_cov_f64 + hgp_score_groups_f64 (a11). */
int hgp_warp_batch_f64(const double* x, const double* Yt, const double* Ym, long ym_stride, int T, int B, int D, int n_ctrl,
                       int iters, double noise, double lam_s, double lam_a, double lr, const double* weights, const double* u0,
                       double* u_out, double* xw_out, double* yw_out, double* loss_out, void* stream);
/* a10, the same call in full: alpha[T] = G^{-T} G^{-1} y  (= scipy cho_solve((G, True), y) with G = tril(K), GPI.py:1043);
 * quad[1] = || G^{-1} y ||^2 (may be NULL). */
int hgp_trsv_lower_solve_f64(const double* G, int ld, const double* y, int T, double* alpha, double* quad, void* stream);
/* a10 gradient (GPI.py:1046-1051): out3[k] = 0.5 tr((alpha alpha^T - Kinv) dK/dtheta_k), theta = (log c, log ell, log noise),
 * with scikit-learn's kernel gradients (ConstantKernel * RBF: c R and c R d^2/ell^2; WhiteKernel: noise I). */
int hgp_lml_grad_f64(const double* x, const double* alpha, const double* Kinv, int T, double c, double ell, double noise,
                     double* out3, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HDPGPC_HIP_H */
