/*
 * apmf.h - C-ABI of libapmf_hip.so: the MI355X (gfx950) hot path of active PMF.
 *
 * The reference (autonlab/active-matrix-factorization, python-pmf/) has no FFI of
 * its own: its native layer is three Cython modules called from Python classes.
 * Each entry point below replaces the reference function(s) cited next to it
 * (paths relative to /root/reference/python-pmf/).  The Python mirror of the
 * reference's class surface (the .py modules of active_matrix_factorization_amd) binds these
 * with ctypes; INTEGRATION.md shows the stub a maintainer of the reference adds.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - Every function returns int: 0 = APMF_OK, < 0 = apmf_status.  No exception
 *    crosses the boundary.  apmf_last_error() gives the message for the calling
 *    thread's last failure.
 *  - Pointers named d_* are DEVICE pointers (hipMalloc'd by apmf_malloc, or any
 *    other allocator on the same device, e.g. a torch tensor's data_ptr()).
 *    Pointers named h_* are HOST pointers, borrowed for the call only.
 *  - All matrices are row-major and dense.  Grids of candidates use a row pitch
 *    `ld` (in elements) >= m, a multiple of APMF_GRID_ALIGN (apmf_grid_ld()).
 *  - One apmf_ctx = one device + one stream.  A ctx is not thread-safe; use one
 *    per thread.  Work is enqueued on the ctx stream; calls that return host
 *    values synchronise that stream, the others are asynchronous.
 *  - Never create a ctx before fork(): a HIP context does not survive it.
 */
#ifndef APMF_H
#define APMF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum apmf_status {
    APMF_OK = 0,
    APMF_EINVAL = -1,  /* bad argument (NULL, shape, unsupported latent_d ...) -> ValueError */
    APMF_EHIP = -2,    /* a HIP runtime call failed -> RuntimeError                          */
    APMF_ENOMEM = -3,  /* device or host allocation failed -> MemoryError                    */
    APMF_ENOTPD = -4,  /* a matrix that must be positive definite is not -> LinAlgError      */
    APMF_ENODEV = -5   /* no usable gfx950 device -> RuntimeError                            */
} apmf_status;

#define APMF_GRID_ALIGN 64  /* row pitch of candidate grids, in elements */
#define APMF_MAX_D 64       /* largest latent_d the kernels are built for  */

typedef struct apmf_ctx apmf_ctx;
typedef struct apmf_ratings apmf_ratings;

/* ---- context, memory, stream plumbing --------------------------------------- */
int apmf_version(void);
const char *apmf_last_error(void);
int apmf_device_count(int *count);
int apmf_ctx_create(int device, apmf_ctx **out);
int apmf_ctx_destroy(apmf_ctx *ctx);
/* run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream);
 * NULL switches back to the ctx's own stream */
int apmf_ctx_set_stream(apmf_ctx *ctx, void *hip_stream);
int apmf_ctx_sync(apmf_ctx *ctx);
/* give the context's cache of freed device blocks (up to 16 GB, kept to spare the classes' many same-sized
 * allocations a hipMalloc each) back to the driver; synchronises the stream first.  No reference counterpart. */
int apmf_ctx_trim(apmf_ctx *ctx);
int apmf_grid_ld(int m); /* m rounded up to APMF_GRID_ALIGN */

/* In-library timing of the dominant kernel of each instrumented entry point (the sweep
 * kernel of apmf_mn_score_grid_*, the gradient kernel of apmf_pmf_gradient/fit_lls): between
 * begin and end every such launch is bracketed by a pair of HIP events on the ctx stream,
 * up to max_records launches.  end() synchronises and returns the durations in ms. */
int apmf_profile_begin(apmf_ctx *ctx, int max_records);
int apmf_profile_end(apmf_ctx *ctx, float *h_ms, int cap, int *h_count);

int apmf_malloc(apmf_ctx *ctx, size_t bytes, void **d_out);
int apmf_free(apmf_ctx *ctx, void *d_ptr);
int apmf_memset(apmf_ctx *ctx, void *d_ptr, int value, size_t bytes);
int apmf_h2d(apmf_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int apmf_d2h(apmf_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int apmf_d2d(apmf_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* strided 2-D copies (pitches and width in bytes), e.g. the (n, m) cross block of the
 * (n+m, n+m) cov_useritems into a grid of pitch ld */
int apmf_h2d_2d(apmf_ctx *ctx, void *d_dst, size_t dpitch, const void *h_src, size_t spitch,
                size_t width, size_t height);
int apmf_d2h_2d(apmf_ctx *ctx, void *h_dst, size_t dpitch, const void *d_src, size_t spitch,
                size_t width, size_t height);
/* device to device, on the stream, no host synchronisation (e.g. H fitted (n, d) / (m, d) factor pairs
 * stacked into the H (n+m, d) means of a batch of approximations) */
int apmf_d2d_2d(apmf_ctx *ctx, void *d_dst, size_t dpitch, const void *d_src, size_t spitch,
                size_t width, size_t height);
/* d_out = d_x + alpha * d_y, product and sum rounded separately like NumPy's `x + alpha * y`
 * (the trial points of the learning-rate searches, pmf_cy.pyx:271-272); in place allowed */
int apmf_axpy_f64(apmf_ctx *ctx, double *d_out, const double *d_x, double alpha, const double *d_y,
                  size_t count);
/* element-wise f64 -> f32 / f32 -> f64 conversion on the device */
int apmf_cast_f64_f32(apmf_ctx *ctx, float *d_dst, const double *d_src, size_t count);
int apmf_cast_f32_f64(apmf_ctx *ctx, double *d_dst, const float *d_src, size_t count);

/* ---- K1: PMF MAP fit over the observed ratings (pmf_cy.pyx) --------------------- */
/* ratings container: replaces `self.ratings` iteration of pmf_cy.pyx:184,217,230.
 * h_ratings is the reference's (N, 3) float64 array of [i, j, r] rows.  Builds, on the
 * device, CSR-by-user and CSR-by-item (int32 indices, float64 values) whose per-row
 * order is the storage order of h_ratings (stable), so every per-row sum is taken in
 * the reference's order. */
int apmf_ratings_create(apmf_ctx *ctx, const double *h_ratings, int64_t N, int n, int m,
                        apmf_ratings **out);
/* The same container straight from the dense inputs of the reference's drivers: d_real (n, ld_real)
 * float64 and d_known (n, ld_known) uint8 on the device, known entries taken in row-major order -
 * the order choose_training.py's make_ratings() writes `_ratings` in (choose_training.py:13-17),
 * so the result is identical to apmf_ratings_create on that list.  Built by device kernels
 * (counts, ordered compaction per row and per column); no (N, 3) list, no Python sets. */
int apmf_ratings_create_dense(apmf_ctx *ctx, const double *d_real, int ld_real, const uint8_t *d_known,
                              int ld_known, int n, int m, apmf_ratings **out);
int apmf_ratings_destroy(apmf_ctx *ctx, apmf_ratings *r);
int apmf_ratings_count(const apmf_ratings *r, int64_t *N, int *n, int *m);

/* gradient(), pmf_cy.pyx:204-223.  d_U (n,d), d_V (m,d) float64 in; d_gU, d_gV out. */
int apmf_pmf_gradient(apmf_ctx *ctx, const apmf_ratings *r, const double *d_U, const double *d_V,
                      int d, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                      int subtract_mean, double mean_rating, double *d_gU, double *d_gV);
/* Row-sharded gradient (multi-GPU): only rows [row_lo, row_hi) of one side
 * (0 = users -> rows of grad_u, 1 = items -> rows of grad_v) are computed and written into
 * the full-size d_g; every rank holds the whole ratings container and the whole U, V. */
int apmf_pmf_gradient_rows(apmf_ctx *ctx, const apmf_ratings *r, int side, int row_lo, int row_hi,
                           const double *d_U, const double *d_V, int d, double sigma_sq,
                           double sigma_u_sq, double sigma_v_sq, int subtract_mean,
                           double mean_rating, double *d_g);
/* The three sums log_likelihood() is made of, restricted to a shard: h_terms3[0] = squared
 * error over the ratings of users [user_lo, user_hi), [1] = ||U[user_lo:user_hi]||^2,
 * [2] = ||V[item_lo:item_hi]||^2.  Summed over the shards they give pmf_cy.pyx:191-193. */
int apmf_pmf_ll_terms(apmf_ctx *ctx, const apmf_ratings *r, const double *d_U, const double *d_V, int d,
                      int subtract_mean, double mean_rating, int user_lo, int user_hi, int item_lo,
                      int item_hi, double *h_terms3);
/* apmf_pmf_ll_terms with the three sums left ON THE DEVICE (d_terms3, three doubles) and no host
 * synchronisation: the partial sums of a rank travel to the others inside the trial's collective. */
int apmf_pmf_ll_terms_dev(apmf_ctx *ctx, const apmf_ratings *r, const double *d_U, const double *d_V, int d,
                          int subtract_mean, double mean_rating, int user_lo, int user_hi, int item_lo,
                          int item_hi, double *d_terms3);
/* log_likelihood(), pmf_cy.pyx:170-193 (h_ll) and the bare sum of squared errors that
 * update_sigma() needs, pmf_cy.pyx:225-234 (h_sq_error; either may be NULL). */
int apmf_pmf_log_likelihood(apmf_ctx *ctx, const apmf_ratings *r, const double *d_U,
                            const double *d_V, int d, double sigma_sq, double sigma_u_sq,
                            double sigma_v_sq, int subtract_mean, double mean_rating,
                            double *h_ll, double *h_sq_error);
/* fit_lls(), pmf_cy.pyx:257-291: batch gradient ascent with the x1.25 / x0.5 learning-rate
 * search, entirely on the device.  d_U, d_V are updated in place.  Runs until convergence
 * or until max_steps accepted steps (max_steps < 0: no limit).  h_ll_trace[0..min(steps,
 * max_trace)) receives the LL the reference's generator would yield.  *h_lr: in = starting
 * learning rate, out = learning rate after the last step (so a caller can resume);
 * *h_old_ll (may be NULL): in = the LL the search compares against (NaN: evaluate it at
 * entry, as fit_lls does before its loop), out = LL of the last accepted step - carrying it
 * across calls reproduces the generator exactly even when a caller changes the sigmas
 * between steps (fit_with_sigmas_lls, pmf_cy.pyx:385-403); *h_converged: the reference's
 * `converged` flag. */
int apmf_pmf_fit_lls(apmf_ctx *ctx, const apmf_ratings *r, double *d_U, double *d_V, int d,
                     double sigma_sq, double sigma_u_sq, double sigma_v_sq, int subtract_mean,
                     double mean_rating, double *h_lr, double *h_old_ll,
                     double min_learning_rate, double stop_thresh, int64_t max_steps,
                     double *h_ll_trace, int64_t max_trace, int64_t *h_steps, int *h_converged);
/* H fit_lls searches as ONE chain of launches - the MAP re-fits of a lookahead with refit_lookahead
 * (mn_active_pmf.py:660-665 / active_pmf.py:660-665: deepcopy, add_rating(i, j, v), do_fit() per hypothesis;
 * bayes_pmf.py:457-468 with fit_first).  Hypothesis h is the shared container `r` plus the rating
 * (h_ei[h], h_ej[h], h_ev[h]) taken as the last entry of user row i / item row j - where add_rating() and a
 * rebuilt container put it - and starts from its own factors d_U (H, n, d), d_V (H, m, d), updated in place.
 * Row order, partial sums and decisions are those of apmf_pmf_fit_lls on the rebuilt container: results are
 * bit-identical to H separate calls.  h_mean_rating (H; only read with subtract_mean): mean of the H extended
 * rating lists.  Outputs (each nullable): final log-likelihood, accepted steps, converged flag per hypothesis. */
int apmf_pmf_batch_fit_lls(apmf_ctx *ctx, const apmf_ratings *r, int d, int H, const int32_t *h_ei,
                           const int32_t *h_ej, const double *h_ev, double *d_U, double *d_V, double sigma_sq,
                           double sigma_u_sq, double sigma_v_sq, int subtract_mean, const double *h_mean_rating,
                           double learning_rate, double min_learning_rate, double stop_thresh, int64_t max_steps,
                           double *h_ll, int64_t *h_steps, int *h_converged);
/* One epoch of fit_minibatches(), pmf_cy.pyx:324-344, on ALREADY SHUFFLED ratings
 * (the host keeps np.random.shuffle so the RNG stream matches): for each consecutive
 * batch  inc = momentum*inc + grad*(lr/len(batch)); params += inc.  d_U, d_V, d_uinc,
 * d_vinc are updated in place. */
int apmf_pmf_minibatch_epoch(apmf_ctx *ctx, const double *h_shuffled_ratings, int64_t N, int n,
                             int m, int64_t batch_size, double lr, double momentum, double *d_U,
                             double *d_V, double *d_uinc, double *d_vinc, int d, double sigma_sq,
                             double sigma_u_sq, double sigma_v_sq, int subtract_mean,
                             double mean_rating);
/* predicted_matrix(), pmf_cy.pyx:410-420: d_out (n, ld) = U V^T (+ mean_rating). */
int apmf_pmf_predicted_matrix(apmf_ctx *ctx, const double *d_U, const double *d_V, int n, int m,
                              int d, double add, double *d_out, int ld);

/* ---- K2: candidate-grid scoring under the matrix-normal posterior ----------------
 * Replaces MNActivePMF.approx_pred_mean_var / approx_pred_means_vars / pred_variance /
 * _total_variance (mn_active_pmf.py:300-330, 501-511, 597-598) and the per-candidate
 * matrix_normal_exps_cy.exp_dotprod_sq (matrix_normal_exps_cy.pyx:126-154) they call.
 *
 * A posterior state is  mean M (n+m, d),  Sigma = cov_useritems,  Omega = cov_latents.
 * Only what the path reads is passed: the (n, m) cross block Sigma[i, n+j] as a grid of
 * pitch ld, and the diagonal Sigma[t, t] (n+m).  B states are scored per launch (B >= 1):
 * every array has a leading batch dimension with the strides implied by its shape.
 *   d_mean   (B, n+m, d)      d_cross (B, n, ld)     d_diag (B, n+m)     d_omega (B, d, d)
 * Outputs (each may be NULL):
 *   d_var    (B, n, ld)   Var[U_i . V_j]            d_pmean (B, n, ld)  E[U_i . V_j]
 *   d_sum    (B) float64  sum of Var over the candidates (mask applied)  [_total_variance]
 *   d_best_val (B) / d_best_idx (B, int64 flat i*m+j): arg-extremum over the candidates
 *       (want = +1 max, -1 min, 0 none); ties -> lowest flat index; idx = -1 if none.
 * d_mask (n, ld) uint8, shared by all states, 1 = candidate (unrated), NULL = all.
 * Masked-out entries of d_var/d_pmean are still written (the value is well defined).
 */
int apmf_mn_score_grid_f32(apmf_ctx *ctx, int B, int n, int m, int d, int ld, const float *d_mean,
                           const float *d_cross, const float *d_diag, const float *d_omega,
                           const uint8_t *d_mask, float *d_var, float *d_pmean, double *d_sum,
                           int want, float *d_best_val, int64_t *d_best_idx);
int apmf_mn_score_grid_f64(apmf_ctx *ctx, int B, int n, int m, int d, int ld,
                           const double *d_mean, const double *d_cross, const double *d_diag,
                           const double *d_omega, const uint8_t *d_mask, double *d_var,
                           double *d_pmean, double *d_sum, int want, double *d_best_val,
                           int64_t *d_best_idx);
/* Arbitrary pool of candidates (what _get_key_vals iterates, mn_active_pmf.py:731-762):
 * d_pi, d_pj (P) int32 -> d_pmean, d_var (P) float64, single state. */
int apmf_mn_score_pool_f64(apmf_ctx *ctx, int n, int m, int d, int ld, const double *d_mean,
                           const double *d_cross, const double *d_diag, const double *d_omega,
                           const int32_t *d_pi, const int32_t *d_pj, int64_t P, double *d_pmean,
                           double *d_var);
/* After an f32 sweep: collect every candidate whose f32 score is within rel_tol of the
 * extremum (d_var of one state as written by apmf_mn_score_grid_f32), up to cap entries,
 * so the caller can re-score that short list in float64 (apmf_mn_score_pool_f64) and make
 * the selection exact.  h_count receives the number found (may exceed cap). */
int apmf_grid_collect_near_f32(apmf_ctx *ctx, int n, int m, int ld, const float *d_var,
                               const uint8_t *d_mask, int want, float best, float rel_tol,
                               int32_t *d_pi, int32_t *d_pj, int cap, int *h_count);

/* After a float64 sweep: every candidate whose score equals `best` exactly (the ties the
 * reference's first-in-pool-order rule has to break, mn_active_pmf.py:729), up to cap entries;
 * *h_count receives the number found (may exceed cap). */
int apmf_grid_collect_eq_f64(apmf_ctx *ctx, int n, int m, int ld, const double *d_var,
                             const uint8_t *d_mask, double best, int32_t *d_pi, int32_t *d_pj, int cap,
                             int *h_count);

/* ---- K2-full: same under the full-covariance Gaussian --------------------------------
 * Replaces ActivePMF.approx_pred_mean_var / approx_pred_means_vars / pred_variance
 * (active_pmf.py:301-322, 392-400, 502-524) + normal_exps_cy.exp_dotprod_sq
 * (normal_exps_cy.pyx:111-135).  d_mean (K), d_cov (K, K) float64, K = (n+m)*d with the
 * reference's index maps u[k,i] = i*d+k, v[k,j] = n*d + j*d + k (active_pmf.py:141-142). */
int apmf_full_score_grid_f64(apmf_ctx *ctx, int n, int m, int d, int ld, const double *d_mean,
                             const double *d_cov, const uint8_t *d_mask, double *d_var,
                             double *d_pmean, double *d_sum, int want, double *d_best_val,
                             int64_t *d_best_idx);

/* ---- K3: Bayesian-PMF Gibbs row updates (bayes_pmf.py) --------------------------------
 * One half-sweep of samples(), bayes_pmf.py:280-296 = `rows` independent sample_feature()
 * calls (bayes_pmf.py:189-216): for each row t of the chosen side
 *     F = oth[idx(t)],  cov = inv(alpha + beta F^T F),  mean = cov (beta F^T (r - mean_sub) + alpha mu)
 *     out[t] = chol(cov) z[t] + mean                        (lower Cholesky of the covariance)
 * side 0: rows = users, oth = item features (m, d); side 1: rows = items, oth = users.
 * d_alpha (d,d), d_mu (d), d_z (rows, d) standard-normal draws made by the host RNG in the
 * reference's order, d_out (rows, d).  A row with no rating -> APMF_EINVAL (the reference
 * raises KeyError, bayes_pmf.py:287,296); a non-PD system -> APMF_ENOTPD. */
int apmf_gibbs_half_sweep(apmf_ctx *ctx, const apmf_ratings *r, int side, const double *d_oth,
                          int d, const double *d_alpha, const double *d_mu, double beta,
                          double mean_sub, const double *d_z, double *d_out);

/* Row-sharded half-sweep: only rows [row_lo, row_hi) of `side` are resampled (d_z and d_out
 * are indexed by the global row). */
int apmf_gibbs_half_sweep_rows(apmf_ctx *ctx, const apmf_ratings *r, int side, int row_lo,
                               int row_hi, const double *d_oth, int d, const double *d_alpha,
                               const double *d_mu, double beta, double mean_sub,
                               const double *d_z, double *d_out);

/* H half-sweeps in one launch (the hypothesis is a grid dimension): the lookahead of
 * BayesianPMF.exp_variance (bayes_pmf.py:457-525, 560-598), where every hypothesis "R_ij = v" is
 * deepcopy -> add_rating -> samples().  All arrays carry a leading H: d_oth (H, other rows, d),
 * d_alpha (H, d, d), d_mu (H, d), d_mean_sub (H), d_z / d_out (H, rows, d).  d_ei / d_ej / d_ev (H,
 * nullable together): ONE rating per hypothesis on top of the shared container, taken as the last
 * entry of its row - where add_rating() (pmf_cy.pyx:128-156) + a rebuilt container puts it. */
int apmf_gibbs_half_sweep_batched(apmf_ctx *ctx, const apmf_ratings *r, int side, int H,
                                  const double *d_oth, int d, const double *d_alpha, const double *d_mu,
                                  double beta, const double *d_mean_sub, const double *d_z, double *d_out,
                                  const int32_t *d_ei, const int32_t *d_ej, const double *d_ev);

/* ---- K4: statistics over S posterior samples (bayes_pmf.py:433-448, 528-538) -----------
 * predict / pred_variance / prob_ge_cutoff: per (i, j) the mean, the variance (ddof 0) and
 * the fraction >= cutoff of  U_s[i] . V_s[j] + add  over s < S.
 * d_Us (S, n, d), d_Vs (S, m, d) float64; outputs (n, ld) float64, each may be NULL. */
int apmf_sample_stats_f64(apmf_ctx *ctx, int S, int n, int m, int d, int ld, const double *d_Us,
                          const double *d_Vs, double add, double cutoff, double *d_mean,
                          double *d_var, double *d_prob);

/* Gather (n, ld) grid entries at a pool of (i, j): d_out[p] = d_grid[pi[p]*ld + pj[p]]. */
int apmf_grid_gather_f64(apmf_ctx *ctx, const double *d_grid, int ld, const int32_t *d_pi,
                         const int32_t *d_pj, int64_t P, double *d_out);
/* Arg-extremum of a pool vector with the reference's tie rule (first in pool order:
 * python max()/min() over zip(pool, vals), mn_active_pmf.py:729; np.argmax, bayes_pmf.py:712). */
int apmf_pool_argext_f64(apmf_ctx *ctx, const double *d_vals, int64_t P, int want,
                         int64_t *h_pos, double *h_val);

/* ---- dense symmetric linear algebra of the variational fit (libapmf_dense.so) -----------
 * Exported by a second shared library (rocSOLVER / rocBLAS underneath) that shares the
 * apmf_ctx of libapmf_hip.so.  d_mat is an N x N float64 matrix, N = n + m (or d).
 *
 * project_psd(), mn_active_pmf.py:42-67 / active_pmf.py:36-50: symmetrise in place; if the
 * smallest eigenvalue is < min_eig rebuild the matrix with its spectrum clipped at min_eig
 * (then symmetrise again).  *h_projected tells which branch ran. */
int apmf_dense_project_psd(apmf_ctx *ctx, double *d_mat, int N, double min_eig, int *h_projected);
/* np.linalg.inv and the log|det| of np.linalg.slogdet (matrix_normal_exps_cy.pyx:180-182,
 * 477-485) by LU (getrf / getri, like NumPy): d_inv (N x N, nullable) and *h_logdet
 * (nullable).  Singular -> APMF_ENOTPD (NumPy raises LinAlgError("Singular matrix")). */
int apmf_dense_inv_logdet(apmf_ctx *ctx, const double *d_mat, int N, double *d_inv, double *h_logdet);

/* ---- K5: KL objective and gradient of the matrix-normal variational fit (libapmf_dense.so)
 * d_mean (n+m, d), d_cov_useritems (n+m, n+m), d_cov_latents (d, d), float64, device.
 * mn_kl_divergence(), matrix_normal_exps_cy.pyx:159-213 (quirks kept: the item-covariance
 * trace term is dropped and the item prior uses sigma_u_sq). */
int apmf_mn_kl_value(apmf_ctx *ctx, const apmf_ratings *r, int d, const double *d_mean,
                     const double *d_cov_useritems, const double *d_cov_latents, double sigma_sq,
                     double sigma_u_sq, double sigma_v_sq, double *h_kl);
/* matrixnormal_gradient() / _mnormal_grad(), matrix_normal_exps_cy.pyx:216-485: gradient
 * w.r.t. one triangular half written into both halves, like the reference. */
int apmf_mn_kl_gradient(apmf_ctx *ctx, const apmf_ratings *r, int d, const double *d_mean,
                        const double *d_cov_useritems, const double *d_cov_latents, double sigma_sq,
                        double sigma_u_sq, double sigma_v_sq, double *d_g_mean,
                        double *d_g_cov_useritems, double *d_g_cov_latents);

/* fit_normal_kls() / fit_normal(), mn_active_pmf.py:236-288, with the approximation resident on
 * the device: projected gradient descent on the KL with the reference's learning-rate search
 * (x1.25 on an improving trial, x0.5 otherwise; converged when the KL improves by < .005 or
 * lr < 1e-10).  d_mean / d_cov_useritems / d_cov_latents are updated in place.  *h_lr and
 * *h_old_kl carry the search state between calls (pass NaN as *h_old_kl the first time):
 * max_steps = 1 per `yield` of the generator, -1 for the whole fit.  The KL after every
 * accepted step goes to h_kl_trace (first trace_cap entries). */
int apmf_mn_fit_normal(apmf_ctx *ctx, const apmf_ratings *r, int d, double *d_mean,
                       double *d_cov_useritems, double *d_cov_latents, double sigma_sq,
                       double sigma_u_sq, double sigma_v_sq, double min_eig, double *h_lr,
                       double *h_old_kl, int64_t max_steps, double *h_kl_trace, int64_t trace_cap,
                       int64_t *h_steps, int *h_converged);

/* ---- batched variational fits: the one-step lookahead (libapmf_dense.so) ----------------------
 * Replaces the body of _exp_with_rij's calculate_fn mapped over the pool (mn_active_pmf.py:660-672,
 * 749-755): for every hypothesis "entry (i, j) turns out to be v"
 *     deepcopy(model) -> add_rating(i, j, v) -> fit_normal() -> fn(model)
 * A batch holds H matrix-normal approximations of the same shape resident on the device; every
 * launch of the fit advances all of them (the hypothesis is a grid dimension), and the accept /
 * reject search state of each is device state.  A hypothesis may carry ONE rating on top of the
 * shared ratings container - it is treated as the last entry of its user row and item row, which
 * is where add_rating() (pmf_cy.pyx:128-156) followed by a rebuilt container puts it, so a
 * batched fit is bit-identical to the same fit run alone.  H = 1 without an extra rating is
 * MNActivePMF.fit_normal itself (apmf_mn_fit_normal above is that case).
 * `r` must outlive the batch.  trace_cap: KL values kept per hypothesis (0: none). */
typedef struct apmf_mn_batch apmf_mn_batch;
int apmf_mn_batch_create(apmf_ctx *ctx, const apmf_ratings *r, int d, int H, int64_t trace_cap,
                         apmf_mn_batch **out);
int apmf_mn_batch_destroy(apmf_ctx *ctx, apmf_mn_batch *b);
int apmf_mn_batch_size(const apmf_mn_batch *b, int *H, int *n, int *m, int *d);
/* Current point of every hypothesis: d_mean (n+m, d), d_cov_useritems (n+m, n+m), d_cov_latents
 * (d, d), each nullable (= leave as is).  per_hypothesis bit 0 / 1 / 2: the mean / cov_useritems /
 * cov_latents argument holds H consecutive arrays instead of one shared by all hypotheses. */
int apmf_mn_batch_set_state(apmf_ctx *ctx, apmf_mn_batch *b, const double *d_mean,
                            const double *d_cov_useritems, const double *d_cov_latents,
                            int per_hypothesis);
/* The extra rating (h_i[h], h_j[h], h_v[h]) of every hypothesis; h_i[h] < 0 (or h_i == NULL): none.
 * The entry must not be rated in `r` (add_rating raises ValueError; not checked here). */
int apmf_mn_batch_set_extra(apmf_ctx *ctx, apmf_mn_batch *b, const int32_t *h_i, const int32_t *h_j,
                            const double *h_v);
/* Copy the current point of hypothesis h out (device to device; each nullable). */
int apmf_mn_batch_get_state(apmf_ctx *ctx, apmf_mn_batch *b, int h, double *d_mean,
                            double *d_cov_useritems, double *d_cov_latents);
/* fit_normal_kls() / fit_normal() (mn_active_pmf.py:236-288) for all H hypotheses.  h_lr, h_old_kl
 * (H each) carry the search state in and out (NaN old_kl: evaluate the KL at entry, as the
 * reference does before its loop); at most max_steps accepted steps per hypothesis (< 0: until
 * convergence, 0: only evaluate).  h_kl_trace (H, trace_cap), h_steps (H), h_converged (H): nullable. */
int apmf_mn_batch_fit(apmf_ctx *ctx, apmf_mn_batch *b, double sigma_sq, double sigma_u_sq,
                      double sigma_v_sq, double min_eig, double *h_lr, double *h_old_kl,
                      int64_t max_steps, double *h_kl_trace, int64_t *h_steps, int *h_converged);
/* Counters of the last apmf_mn_batch_fit (each nullable): iterations of the search loop, trial points
 * evaluated, the project_psd calls (mn_active_pmf.py:42-67) that really clipped eigenvalues - how many
 * were finished through the few eigenpairs below min_eig, how many through the full eigendecomposition -
 * the trial points whose "nothing to clip" was certified by an eigenvalue bound (Weyl, from the inverse
 * the gradient computes) without factoring the shifted matrix, and the gradients evaluated (one per accepted
 * step plus the first). */
int apmf_mn_batch_fit_stats(const apmf_mn_batch *b, int64_t *iterations, int64_t *trials, int64_t *proj_low,
                            int64_t *proj_full, int64_t *cert_by_bound, int64_t *gradients);
/* Kernel launches the batch engine has enqueued so far in this process (all batches, all streams): the
 * difference around a call is its launch count.  A hypothesis of the reference's lookahead
 * (mn_active_pmf.py:660-668) is a chain of dependent launches here; launches per hypothesis is tracked. */
int apmf_mn_batch_launch_count(int64_t *launches);
/* kl_divergence() of every current point (mn_kl_divergence, matrix_normal_exps_cy.pyx:159-213). */
int apmf_mn_batch_kl(apmf_ctx *ctx, apmf_mn_batch *b, double sigma_sq, double sigma_u_sq,
                     double sigma_v_sq, double *h_kl);
/* matrixnormal_gradient() at the current point of hypothesis h, into caller buffers. */
int apmf_mn_batch_gradient(apmf_ctx *ctx, apmf_mn_batch *b, int h, double sigma_sq, double sigma_u_sq,
                           double sigma_v_sq, double *d_g_mean, double *d_g_cov_useritems,
                           double *d_g_cov_latents);
/* _total_variance() (mn_active_pmf.py:597-598) and _approx_entropy() (:513-521) of every
 * hypothesis: h_out (H).  The sweep is K2 with B = H, straight from the resident states. */
int apmf_mn_batch_total_variance(apmf_ctx *ctx, apmf_mn_batch *b, double *h_out);
int apmf_mn_batch_entropy(apmf_ctx *ctx, apmf_mn_batch *b, double *h_out);
/* K2 over hypotheses [h0, h0 + B) from the resident states; outputs as apmf_mn_score_grid_f64
 * (use_f32 = 0: double grids / best value) or _f32 (use_f32 = 1: float), leading dimension B. */
int apmf_mn_batch_score_grid(apmf_ctx *ctx, apmf_mn_batch *b, int h0, int B, int use_f32, int ld,
                             const uint8_t *d_mask, void *d_var, void *d_pmean, double *d_sum,
                             int want, void *d_best_val, int64_t *d_best_idx);

/* approx_pred_mean_var over an arbitrary pool (mn_active_pmf.py:300-315) for hypothesis h, from
 * its resident state: d_pi, d_pj (P) int32 -> d_pmean, d_var (P) float64 (each nullable). */
int apmf_mn_batch_score_pool(apmf_ctx *ctx, apmf_mn_batch *b, int h, const int32_t *d_pi,
                             const int32_t *d_pj, int64_t P, double *d_pmean, double *d_var);

/* normal_gradient() / _normal_grad(), normal_exps_cy.pyx:140-303: gradient of the KL of the
 * full-covariance approximation (ActivePMF.fit_normal) w.r.t. mean (K) and cov (K x K),
 * K = (n+m) d, stacked as u[k,i] = i d + k, v[k,j] = n d + j d + k (active_pmf.py:141-142).
 * The reference's arithmetic is kept, summed increments included.  latent_d <= 32. */
int apmf_full_kl_gradient(apmf_ctx *ctx, const apmf_ratings *r, int d, const double *d_mean,
                          const double *d_cov, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                          double *d_g_mean, double *d_g_cov);

/* ActivePMF.kl_divergence (active_pmf.py:202-246) and fit_normal_kls / fit_normal (:250-288) of the
 * full-covariance approximation, resident on the device: d_mean (K), d_cov (K x K), K = (n+m) d.
 * Calling convention of apmf_mn_fit_normal. */
int apmf_full_kl_value(apmf_ctx *ctx, const apmf_ratings *r, int d, const double *d_mean,
                       const double *d_cov, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                       double *h_kl);
int apmf_full_fit_normal(apmf_ctx *ctx, const apmf_ratings *r, int d, double *d_mean, double *d_cov,
                         double sigma_sq, double sigma_u_sq, double sigma_v_sq, double min_eig,
                         double *h_lr, double *h_old_kl, int64_t max_steps, double *h_kl_trace,
                         int64_t trace_cap, int64_t *h_steps, int *h_converged);

/* H variational fits of the FULL-covariance approximation as one chain of launches (csrc/k8_fullbatch.hip): the
 * one-step lookahead of the full-covariance class (active_pmf.py:635-704: deepcopy, add_rating(i, j, v),
 * fit_normal per hypothesis and rating value) for a whole pool at once, and - with H = 1 - ActivePMF.fit_normal
 * itself, so that the batched lookahead equals the per-hypothesis one bit for bit.  Hypothesis h is the shared
 * container `r` plus ONE extra rating, taken as the last entry of user row i / item row j (where add_rating() and a
 * rebuilt container put it); its state (mean (K), cov (K x K), K = (n+m) d) stays on the device. */
typedef struct apmf_full_batch apmf_full_batch;
int apmf_full_batch_create(apmf_ctx *ctx, const apmf_ratings *r, int d, int H, apmf_full_batch **out);
int apmf_full_batch_destroy(apmf_ctx *ctx, apmf_full_batch *b);
/* every hypothesis starts from (d_mean, d_cov) - the deep copies of the reference - or, per_hypothesis != 0, from
 * its own (H x K, H x K x K) */
int apmf_full_batch_set_state(apmf_ctx *ctx, apmf_full_batch *b, const double *d_mean, const double *d_cov,
                              int per_hypothesis);
/* the hypotheses' extra ratings (host arrays of H; h_i[h] < 0: none; NULL: none at all) */
int apmf_full_batch_set_extra(apmf_ctx *ctx, apmf_full_batch *b, const int32_t *h_i, const int32_t *h_j,
                              const double *h_v);
int apmf_full_batch_get_state(apmf_ctx *ctx, apmf_full_batch *b, int h, double *d_mean, double *d_cov);
/* fit_normal_kls (active_pmf.py:250-288) of every hypothesis: h_lr / h_old_kl (H each) in = learning rate and KL to
 * resume from (NaN: evaluate it), out = where the search stands; max_steps < 0: to convergence (an accepted step
 * improving the KL by < 0.005, or lr < 1e-10).  h_kl_trace (H x trace_cap, nullable): the KL after each accepted
 * step of this call.  h_steps / h_converged (H, nullable). */
int apmf_full_batch_fit(apmf_ctx *ctx, apmf_full_batch *b, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                        double min_eig, double *h_lr, double *h_old_kl, int64_t max_steps, double *h_kl_trace,
                        int64_t trace_cap, int64_t *h_steps, int *h_converged);
/* counters of the last fit: search iterations, trial points, trials whose project_psd really clipped eigenvalues */
int apmf_full_batch_fit_stats(const apmf_full_batch *b, int64_t *iterations, int64_t *trials, int64_t *projected);
/* kl_divergence() (active_pmf.py:202-246), log|det cov| (_approx_entropy, :519-521) and _total_variance()
 * (:597-598: the K2-full sweep with the fused sum) of every current point */
int apmf_full_batch_kl(apmf_ctx *ctx, apmf_full_batch *b, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                       double *h_kl);
int apmf_full_batch_logdet(apmf_ctx *ctx, apmf_full_batch *b, double *h_logdet);
int apmf_full_batch_total_variance(apmf_ctx *ctx, apmf_full_batch *b, double *h_out);

/* Optional device-side standard-normal draws for the Gibbs sweeps (Philox4x32-10 + Box-Muller,
 * counter = offset + index / 2): d_out[0..count) ~ N(0, 1).  Not NumPy's stream - BayesianPMF
 * uses it only with device_rng switched on (the reference draws np.random.normal per row,
 * bayes_pmf.py:212-215). */
int apmf_fill_normal_f64(apmf_ctx *ctx, double *d_out, int64_t count, uint64_t seed, uint64_t offset);

/* ---- exchange steps of the sharded path (libapmf_comm.so, RCCL over xGMI) -----------------------
 * Replace the reference's multiprocessing.Pool pickling of whole models (active_pmf.py:739-770,
 * bayes_pmf.py:385-422) by collectives on device pointers, enqueued on the context's stream.
 * One process per GPU; rank 0 draws the 128-byte id (apmf_comm_unique_id) and the host side hands
 * it to the other ranks (sharding.py does so over a TCP socket). */
typedef struct apmf_comm apmf_comm;
int apmf_comm_unique_id(void *h_id128);
int apmf_comm_create(apmf_ctx *ctx, int rank, int world, const void *h_id128, apmf_comm **out);
int apmf_comm_destroy(apmf_comm *c);
int apmf_comm_rank(const apmf_comm *c, int *rank, int *world);
/* d_recv (world x bytes_per_rank) <- every rank's d_send, in rank order: the row blocks of U, V,
 * of the freshly sampled Gibbs side, of diag(Sigma); the per-rank (sum, value, index) records */
int apmf_comm_allgather(apmf_ctx *ctx, apmf_comm *c, const void *d_send, void *d_recv,
                        size_t bytes_per_rank);
/* `count` (<= 16) all-gathers enqueued as one RCCL group - one launch: d_recv[k] (world x bytes_per_rank[k]) <-
 * every rank's d_send[k].  The exchange of one sharded fit_lls trial (pmf_cy.pyx:266-291 row-sharded): the
 * U / V row blocks of both speculative next trial points and the three partial sums of the log-likelihood. */
int apmf_comm_allgather_multi(apmf_ctx *ctx, apmf_comm *c, int count, const void *const *d_send,
                              void *const *d_recv, const size_t *bytes_per_rank);
/* RCCL's version code, for the diagnostic line a rank prints when its communicator comes up */
int apmf_comm_version(int *version);
/* element-wise sum over the ranks (the three sums of log_likelihood, pmf_cy.pyx:191-193) */
int apmf_comm_allreduce_f64(apmf_ctx *ctx, apmf_comm *c, const double *d_send, double *d_recv,
                            size_t count);
/* every rank's stream has reached this point; synchronises the calling host thread with it */
int apmf_comm_barrier(apmf_ctx *ctx, apmf_comm *c);

/* ---- the Gaussian-Wishart hyper-parameter step of the Gibbs sampler on the device ----------------
 * BayesianPMF.sample_hyperparam + sample_wishart (bayes_pmf.py:157-186, 41-59; the scalar-inner-
 * product quirk of :171-176 kept): (mu, alpha) given the feature matrix d_feats (N, d), d <= 32,
 * hyper-prior d_wi (d, d), b0, df, d_mu0 (d).  Wishart draw by Bartlett's decomposition.
 * d_variates: NULL = draw from the Philox stream (seed, offset; a call consumes < 1024 counters), or
 * [chi-squares (d) | N(0,1) of the strict lower triangle, row-major (d (d-1) / 2) | z (d)] - what NumPy
 * draws, in this order, in the Bartlett branch of sample_wishart and for mu.
 * d_status: device int (zeroed by the caller), set to 1 on a non-positive-definite matrix, no host
 * synchronisation; NULL: the call synchronises and returns APMF_ENOTPD itself. */
int apmf_gibbs_sample_hyperparam(apmf_ctx *ctx, const double *d_feats, int N, int d, const double *d_wi,
                                 double b0, double df, const double *d_mu0, const double *d_variates,
                                 uint64_t seed, uint64_t offset, double *d_mu, double *d_alpha,
                                 int *d_status);
/* H chains at once with SUPPLIED variates - the hyper-parameter steps of the batched exp_variance lookahead
 * (bayes_pmf.py:457-525: one Gibbs chain per hypothesis): d_feats (H, N, d), d_variates (H, nvar), d_mu (H, d),
 * d_alpha (H, d, d); d_status: one device int (zeroed by the caller), set if a matrix was not positive definite.
 * direct_dof = 0: Bartlett variates [chi (d) | normals (d (d-1) / 2) | z (d)]; direct_dof > 0: the reference's
 * small-dof branch of sample_wishart (bayes_pmf.py:47-52), variates [G (d x dof, row-major) | z (d)]. */
int apmf_gibbs_sample_hyperparam_batched(apmf_ctx *ctx, int H, const double *d_feats, int N, int d,
                                         const double *d_wi, double b0, double df, const double *d_mu0,
                                         const double *d_variates, int direct_dof, double *d_mu,
                                         double *d_alpha, int *d_status);

#ifdef __cplusplus
}
#endif
#endif /* APMF_H */
