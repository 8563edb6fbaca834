/*
 * apmf_oracle.c - CPU restatement of the reference's python-pmf hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The product (active_matrix_factorization_amd) never imports or links it.
 *
 * Parity pinning: every function here is checked against the golden vectors in
 * the tests/golden npz files, which were produced by the real reference (Cython build
 * of /root/reference/python-pmf, see tests/golden/make_golden.py).
 *
 * Plain C, float64 everywhere (the reference's dtype), scalar, single thread,
 * same per-rating / per-candidate operation order as the reference.  Citations
 * are relative to /root/reference/python-pmf/.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fPIC -shared).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* numpy's float add.reduce over a (strided) 1-D run: pairwise summation with an
 * 8-way unrolled base case (numpy/_core/src/umath/loops_utils.h.src).  Used
 * wherever the reference calls ndarray.sum()/trace() so that the oracle rounds
 * the same way (mn_active_pmf.py:311-312, pmf_cy.pyx:188-189). */
static double np_sum(const double *a, long n, long stride)
{
    if (n < 8) {
        double res = 0.;
        for (long i = 0; i < n; i++) res += a[i * stride];
        return res;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (i = 0; i < 8; i++) r[i] = a[i * stride];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[(i + k) * stride];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i * stride];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_sum(a, n2, stride) + np_sum(a + n2 * stride, n - n2, stride);
    }
}

static double dot_seq(const double *x, const double *y, int d)
{
    double s = 0.;
    for (int k = 0; k < d; k++) s += x[k] * y[k];
    return s;
}

/* ------------------------------------------------------------------------- */
/* PMF MAP model                                                              */
/* ------------------------------------------------------------------------- */

/* prediction_for, pmf_cy.pyx:158-168 (np.dot -> sequential dot here) */
static double predict(const double *U, const double *V, int d, long i, long j,
                      int subtract_mean, double mean_rating)
{
    double p = dot_seq(U + i * d, V + j * d, d);
    return subtract_mean ? p + mean_rating : p;
}

/* log_likelihood, pmf_cy.pyx:170-193 */
double orc_pmf_log_likelihood(const double *ratings, long N, const double *U,
                              const double *V, int n, int m, int d, double sig,
                              double sig_u, double sig_v, int subtract_mean,
                              double mean_rating)
{
    double sq_error = 0.;
    for (long t = 0; t < N; t++) {
        long i = (long)ratings[3 * t], j = (long)ratings[3 * t + 1];
        double r_hat = predict(U, V, d, i, j, subtract_mean, mean_rating);
        double e = ratings[3 * t + 2] - r_hat;
        sq_error += e * e;
    }
    /* np.sum(users * users): pairwise over the flattened product */
    long nu = (long)n * d, nv = (long)m * d;
    double *tmp = (double *)malloc(sizeof(double) * (nu > nv ? nu : nv));
    for (long k = 0; k < nu; k++) tmp[k] = U[k] * U[k];
    double user_norm2 = np_sum(tmp, nu, 1);
    for (long k = 0; k < nv; k++) tmp[k] = V[k] * V[k];
    double item_norm2 = np_sum(tmp, nv, 1);
    free(tmp);
    return (-sq_error / (2. * sig) - user_norm2 / (2. * sig_u) - item_norm2 / (2. * sig_v));
}

/* sum of squared errors only (update_sigma, pmf_cy.pyx:225-234) */
double orc_pmf_sq_error(const double *ratings, long N, const double *U,
                        const double *V, int d, int subtract_mean, double mean_rating)
{
    double sq_error = 0.;
    for (long t = 0; t < N; t++) {
        long i = (long)ratings[3 * t], j = (long)ratings[3 * t + 1];
        double e = ratings[3 * t + 2] - predict(U, V, d, i, j, subtract_mean, mean_rating);
        sq_error += e * e;
    }
    return sq_error;
}

/* gradient, pmf_cy.pyx:204-223: prior term, then the ratings in storage order */
void orc_pmf_gradient(const double *ratings, long N, const double *U, const double *V,
                      int n, int m, int d, double sig, double sig_u, double sig_v,
                      int subtract_mean, double mean_rating, double *gU, double *gV)
{
    for (long k = 0; k < (long)n * d; k++) gU[k] = -U[k] / sig_u;
    for (long k = 0; k < (long)m * d; k++) gV[k] = -V[k] / sig_v;
    for (long t = 0; t < N; t++) {
        long i = (long)ratings[3 * t], j = (long)ratings[3 * t + 1];
        double r_hat = predict(U, V, d, i, j, subtract_mean, mean_rating);
        double w = (ratings[3 * t + 2] - r_hat) / sig;
        for (int k = 0; k < d; k++) {
            gU[i * d + k] += V[j * d + k] * w;
            gV[j * d + k] += U[i * d + k] * w;
        }
    }
}

/* fit_lls, pmf_cy.pyx:257-291.  U, V are updated in place; returns the number
 * of accepted steps, the LL after each is written to ll_trace[0..max_trace). */
long orc_pmf_fit_lls(const double *ratings, long N, double *U, double *V, int n, int m,
                     int d, double sig, double sig_u, double sig_v, int subtract_mean,
                     double mean_rating, double lr, double min_lr, double stop_thresh,
                     double *ll_trace, long max_trace, long max_steps)
{
    long nu = (long)n * d, nv = (long)m * d, steps = 0;
    double *gU = (double *)malloc(sizeof(double) * nu), *gV = (double *)malloc(sizeof(double) * nv);
    double *nU = (double *)malloc(sizeof(double) * nu), *nV = (double *)malloc(sizeof(double) * nv);
    double old_ll = orc_pmf_log_likelihood(ratings, N, U, V, n, m, d, sig, sig_u, sig_v,
                                           subtract_mean, mean_rating);
    int converged = 0;
    while (!converged && (max_steps < 0 || steps < max_steps)) {
        orc_pmf_gradient(ratings, N, U, V, n, m, d, sig, sig_u, sig_v, subtract_mean,
                         mean_rating, gU, gV);
        for (;;) {
            for (long k = 0; k < nu; k++) nU[k] = U[k] + lr * gU[k];
            for (long k = 0; k < nv; k++) nV[k] = V[k] + lr * gV[k];
            double new_ll = orc_pmf_log_likelihood(ratings, N, nU, nV, n, m, d, sig, sig_u,
                                                   sig_v, subtract_mean, mean_rating);
            if (new_ll > old_ll) {
                memcpy(U, nU, sizeof(double) * nu);
                memcpy(V, nV, sizeof(double) * nv);
                lr *= 1.25;
                if (new_ll - old_ll < stop_thresh) converged = 1;
                if (steps < max_trace) ll_trace[steps] = new_ll;
                steps++;
                old_ll = new_ll;
                break;
            } else {
                lr *= .5;
                if (lr < min_lr) { converged = 1; break; }
            }
        }
    }
    free(gU); free(gV); free(nU); free(nV);
    return steps;
}

/* ------------------------------------------------------------------------- */
/* Matrix-normal moments, matrix_normal_exps_cy.pyx                           */
/* mean: (n+m, d) row-major; R = cov_useritems (n+m, n+m); C = cov_latents    */
/* ------------------------------------------------------------------------- */

/* quadexpect, matrix_normal_exps_cy.pyx:28-69 */
double orc_mn_quadexpect(const double *mean, const double *R, const double *C, int nm,
                         int d, int a_i, int a_j, int b_i, int b_j, int c_i, int c_j,
                         int d_i, int d_j)
{
    double ma = mean[a_i * d + a_j], mb = mean[b_i * d + b_j];
    double mc = mean[c_i * d + c_j], md = mean[d_i * d + d_j];
    double cov_ab = R[a_i * nm + b_i] * C[a_j * d + b_j];
    double cov_ac = R[a_i * nm + c_i] * C[a_j * d + c_j];
    double cov_ad = R[a_i * nm + d_i] * C[a_j * d + d_j];
    double cov_bc = R[b_i * nm + c_i] * C[b_j * d + c_j];
    double cov_bd = R[b_i * nm + d_i] * C[b_j * d + d_j];
    double cov_cd = R[c_i * nm + d_i] * C[c_j * d + d_j];
    return (ma * mb * mc * md
            + ma * mb * cov_cd + ma * mc * cov_bd + ma * md * cov_bc
            + mb * mc * cov_ad + mb * md * cov_ac + mc * md * cov_ab
            + cov_ab * cov_cd + cov_ac * cov_bd + cov_ad * cov_bc);
}

/* exp_squared, matrix_normal_exps_cy.pyx:73-91 */
double orc_mn_exp_squared(const double *mean, const double *R, const double *C, int nm,
                          int d, int a_i, int a_j, int b_i, int b_j)
{
    double ma = mean[a_i * d + a_j], mb = mean[b_i * d + b_j];
    double cov_ab = R[a_i * nm + b_i] * C[a_j * d + b_j];
    double var_a = R[a_i * nm + a_i] * C[a_j * d + a_j];
    double var_b = R[b_i * nm + b_i] * C[b_j * d + b_j];
    return (4 * ma * mb * cov_ab + 2 * (cov_ab * cov_ab)
            + (ma * ma + var_a) * (mb * mb + var_b));
}

/* exp_a2bc, matrix_normal_exps_cy.pyx:95-121 */
double orc_mn_exp_a2bc(const double *mean, const double *R, const double *C, int nm,
                       int d, int a_i, int a_j, int b_i, int b_j, int c_i, int c_j)
{
    double ma = mean[a_i * d + a_j], mb = mean[b_i * d + b_j], mc = mean[c_i * d + c_j];
    double var_a = R[a_i * nm + a_i] * C[a_j * d + a_j];
    double cov_ab = R[a_i * nm + b_i] * C[a_j * d + b_j];
    double cov_ac = R[a_i * nm + c_i] * C[a_j * d + c_j];
    double cov_bc = R[b_i * nm + c_i] * C[b_j * d + c_j];
    return ((ma * ma + var_a) * (mb * mc + cov_bc) + 2 * ma * mc * cov_ab
            + 2 * ma * mb * cov_ac + 2 * cov_ab * cov_ac);
}

/* tripexpect, matrix_normal_exps_cy.pyx:9-23 */
double orc_mn_tripexpect(const double *mean, const double *R, const double *C, int nm,
                         int d, int a_i, int a_j, int b_i, int b_j, int c_i, int c_j)
{
    double ma = mean[a_i * d + a_j], mb = mean[b_i * d + b_j], mc = mean[c_i * d + c_j];
    return (ma * mb * mc + ma * R[b_i * nm + c_i] * C[b_j * d + c_j]
            + mb * R[a_i * nm + c_i] * C[a_j * d + c_j]
            + mc * R[a_i * nm + b_i] * C[a_j * d + b_j]);
}

/* exp_dotprod_sq, matrix_normal_exps_cy.pyx:126-154 */
double orc_mn_exp_dotprod_sq(int num_users, const double *mean, const double *R,
                             const double *C, int nm, int d, int i, int j)
{
    double e = 0;
    int j_ = num_users + j;
    for (int k = 0; k < d; k++) {
        e += orc_mn_exp_squared(mean, R, C, nm, d, i, k, j_, k);
        for (int l = k + 1; l < d; l++)
            e += 2 * orc_mn_quadexpect(mean, R, C, nm, d, i, k, j_, k, i, l, j_, l);
    }
    return e;
}

/* approx_pred_mean_var, mn_active_pmf.py:300-315 */
void orc_mn_pred_mean_var(int num_users, const double *mean, const double *R,
                          const double *C, int nm, int d, int i, int j, double *out_mean,
                          double *out_var, double *out_e2)
{
    int j_ = num_users + j;
    double prod[128];
    for (int k = 0; k < d; k++) prod[k] = mean[i * d + k] * mean[j_ * d + k];
    double mn = np_sum(prod, d, 1) + R[i * nm + j_] * np_sum(C, d, d + 1);
    double e2 = orc_mn_exp_dotprod_sq(num_users, mean, R, C, nm, d, i, j);
    *out_mean = mn;
    *out_var = e2 - mn * mn;
    if (out_e2) *out_e2 = e2;
}

/* approx_pred_means_vars, mn_active_pmf.py:317-330 (+ e2 for tolerance scaling) */
void orc_mn_pred_means_vars(int n, int m, int d, const double *mean, const double *R,
                            const double *C, double *p_mean, double *p_var, double *p_e2)
{
    int nm = n + m;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            double e2;
            orc_mn_pred_mean_var(n, mean, R, C, nm, d, i, j, &p_mean[(long)i * m + j],
                                 &p_var[(long)i * m + j], &e2);
            if (p_e2) p_e2[(long)i * m + j] = e2;
        }
}

/* Same quantities from a *compact* posterior state: only the (n, m) cross block
 * of cov_useritems plus its diagonal are supplied (what the device kernels
 * consume).  Operation order identical to the functions above. */
void orc_mn_pred_means_vars_compact(int n, int m, int d, const double *mean,
                                    const double *cross, const double *diag,
                                    const double *C, double *p_mean, double *p_var)
{
    double trC = np_sum(C, d, d + 1);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            const double *a = mean + (long)i * d, *b = mean + (long)(n + j) * d;
            double sij = cross[(long)i * m + j], sii = diag[i], sjj = diag[n + j];
            double e = 0;
            for (int k = 0; k < d; k++) {
                double ma = a[k], mb = b[k];
                double cov_ab = sij * C[k * d + k];
                double var_a = sii * C[k * d + k], var_b = sjj * C[k * d + k];
                e += (4 * ma * mb * cov_ab + 2 * (cov_ab * cov_ab)
                      + (ma * ma + var_a) * (mb * mb + var_b));
                for (int l = k + 1; l < d; l++) {
                    /* a=(i,k) b=(j_,k) c=(i,l) d=(j_,l) */
                    double mc = a[l], md = b[l];
                    double c_ab = sij * C[k * d + k], c_ac = sii * C[k * d + l];
                    double c_ad = sij * C[k * d + l], c_bc = sij * C[k * d + l];
                    double c_bd = sjj * C[k * d + l], c_cd = sij * C[l * d + l];
                    e += 2 * (ma * mb * mc * md
                              + ma * mb * c_cd + ma * mc * c_bd + ma * md * c_bc
                              + mb * mc * c_ad + mb * md * c_ac + mc * md * c_ab
                              + c_ab * c_cd + c_ac * c_bd + c_ad * c_bc);
                }
            }
            double prod[128];
            for (int k = 0; k < d; k++) prod[k] = a[k] * b[k];
            double mn = np_sum(prod, d, 1) + sij * trC;
            p_mean[(long)i * m + j] = mn;
            p_var[(long)i * m + j] = e - mn * mn;
        }
}

/* ------------------------------------------------------------------------- */
/* Full-covariance Gaussian moments, normal_exps_cy.pyx                       */
/* index maps u[k,i] = i*d + k, v[k,j] = n*d + j*d + k (active_pmf.py:141-142) */
/* ------------------------------------------------------------------------- */

double orc_full_tripexpect(const double *mean, const double *cov, long K, long a, long b, long c)
{
    return mean[a] * mean[b] * mean[c] + mean[a] * cov[b * K + c] + mean[b] * cov[a * K + c]
           + mean[c] * cov[a * K + b];
}

/* quadexpect, normal_exps_cy.pyx:40-72 */
double orc_full_quadexpect(const double *mean, const double *cov, long K, long a, long b,
                           long c, long d)
{
    double ma = mean[a], mb = mean[b], mc = mean[c], md = mean[d];
    return (ma * mb * mc * md
            + ma * mb * cov[c * K + d] + ma * mc * cov[b * K + d] + ma * md * cov[b * K + c]
            + mb * mc * cov[a * K + d] + mb * md * cov[a * K + c] + mc * md * cov[a * K + b]
            + cov[a * K + b] * cov[c * K + d] + cov[a * K + c] * cov[b * K + d]
            + cov[a * K + d] * cov[b * K + c]);
}

/* exp_squared, normal_exps_cy.pyx:76-85 */
double orc_full_exp_squared(const double *mean, const double *cov, long K, long a, long b)
{
    double cab = cov[a * K + b];
    return 4 * mean[a] * mean[b] * cab + 2 * (cab * cab)
           + (mean[a] * mean[a] + cov[a * K + a]) * (mean[b] * mean[b] + cov[b * K + b]);
}

/* exp_a2bc, normal_exps_cy.pyx:89-107 */
double orc_full_exp_a2bc(const double *mean, const double *cov, long K, long a, long b, long c)
{
    double ma = mean[a], mb = mean[b], mc = mean[c];
    return ((ma * ma + cov[a * K + a]) * (mb * mc + cov[b * K + c])
            + 2 * ma * mc * cov[a * K + b] + 2 * ma * mb * cov[a * K + c]
            + 2 * cov[a * K + b] * cov[a * K + c]);
}

/* exp_dotprod_sq, normal_exps_cy.pyx:111-135 */
double orc_full_exp_dotprod_sq(int n, int d, const double *mean, const double *cov, long K,
                               int i, int j)
{
    double e = 0;
    for (int k = 0; k < d; k++) {
        long uki = (long)i * d + k, vkj = (long)n * d + (long)j * d + k;
        e += orc_full_exp_squared(mean, cov, K, uki, vkj);
        for (int l = k + 1; l < d; l++)
            e += 2 * orc_full_quadexpect(mean, cov, K, uki, vkj, (long)i * d + l,
                                         (long)n * d + (long)j * d + l);
    }
    return e;
}

/* approx_pred_means_vars, active_pmf.py:301-322 */
void orc_full_pred_means_vars(int n, int m, int d, const double *mean, const double *cov,
                              double *p_mean, double *p_var, double *p_e2)
{
    long K = (long)(n + m) * d;
    double t[128];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            for (int k = 0; k < d; k++) {
                long us = (long)i * d + k, vs = (long)n * d + (long)j * d + k;
                t[k] = mean[us] * mean[vs] + cov[us * K + vs];
            }
            double mn = np_sum(t, d, 1);
            double e2 = orc_full_exp_dotprod_sq(n, d, mean, cov, K, i, j);
            p_mean[(long)i * m + j] = mn;
            p_var[(long)i * m + j] = e2 - mn * mn;
            if (p_e2) p_e2[(long)i * m + j] = e2;
        }
}

/* ------------------------------------------------------------------------- */
/* Bayesian PMF Gibbs row update, bayes_pmf.py:189-216                        */
/* ------------------------------------------------------------------------- */

/* general inverse by LU with partial pivoting (what numpy.linalg.inv's dgesv does) */
static int inv_lu(const double *A, double *Ainv, int d)
{
    double *a = (double *)malloc(sizeof(double) * d * d);
    int *piv = (int *)malloc(sizeof(int) * d);
    memcpy(a, A, sizeof(double) * d * d);
    for (int i = 0; i < d; i++) {
        int p = i;
        for (int r = i + 1; r < d; r++)
            if (fabs(a[r * d + i]) > fabs(a[p * d + i])) p = r;
        piv[i] = p;
        if (a[p * d + i] == 0.) { free(a); free(piv); return -1; }
        if (p != i)
            for (int c = 0; c < d; c++) { double t = a[i * d + c]; a[i * d + c] = a[p * d + c]; a[p * d + c] = t; }
        for (int r = i + 1; r < d; r++) {
            a[r * d + i] /= a[i * d + i];
            for (int c = i + 1; c < d; c++) a[r * d + c] -= a[r * d + i] * a[i * d + c];
        }
    }
    for (int col = 0; col < d; col++) {
        double *x = (double *)malloc(sizeof(double) * d);
        for (int r = 0; r < d; r++) x[r] = (r == col) ? 1. : 0.;
        for (int i = 0; i < d; i++)
            if (piv[i] != i) { double t = x[i]; x[i] = x[piv[i]]; x[piv[i]] = t; }
        for (int r = 0; r < d; r++)
            for (int c = 0; c < r; c++) x[r] -= a[r * d + c] * x[c];
        for (int r = d - 1; r >= 0; r--) {
            for (int c = r + 1; c < d; c++) x[r] -= a[r * d + c] * x[c];
            x[r] /= a[r * d + r];
        }
        for (int r = 0; r < d; r++) Ainv[r * d + col] = x[r];
        free(x);
    }
    free(a); free(piv);
    return 0;
}

/* lower Cholesky factor (numpy.linalg.cholesky) */
static int chol_lower(const double *A, double *L, int d)
{
    memset(L, 0, sizeof(double) * d * d);
    for (int j = 0; j < d; j++) {
        double s = A[j * d + j];
        for (int k = 0; k < j; k++) s -= L[j * d + k] * L[j * d + k];
        if (!(s > 0.)) return -2;
        L[j * d + j] = sqrt(s);
        for (int i = j + 1; i < d; i++) {
            double t = A[i * d + j];
            for (int k = 0; k < j; k++) t -= L[i * d + k] * L[j * d + k];
            L[i * d + j] = t / L[j * d + j];
        }
    }
    return 0;
}

/* sample_feature, bayes_pmf.py:189-216, with the standard-normal draw z supplied.
 * oth: other side's features (rows, d); idx/rat: this row's rated indices and
 * ratings (count cnt); mean_sub: subtracted from the ratings (mean_rating when
 * subtract_mean, else 0). */
int orc_sample_feature(const double *oth, int d, const long *idx, const double *rat,
                       long cnt, const double *alpha, const double *mu, double beta,
                       double mean_sub, const double *z, double *out)
{
    double *A = (double *)calloc((size_t)d * d, sizeof(double));
    double *cov = (double *)malloc(sizeof(double) * d * d);
    double *L = (double *)malloc(sizeof(double) * d * d);
    double *rhs = (double *)calloc(d, sizeof(double));
    double *mean = (double *)malloc(sizeof(double) * d);
    for (long t = 0; t < cnt; t++) {
        const double *f = oth + idx[t] * d;
        double r = rat[t] - mean_sub;
        for (int k = 0; k < d; k++) {
            rhs[k] += f[k] * r;
            for (int l = 0; l < d; l++) A[k * d + l] += f[k] * f[l];
        }
    }
    for (int k = 0; k < d; k++) {
        double am = 0;
        for (int l = 0; l < d; l++) { A[k * d + l] = alpha[k * d + l] + beta * A[k * d + l]; am += alpha[k * d + l] * mu[l]; }
        rhs[k] = beta * rhs[k] + am;
    }
    int rc = inv_lu(A, cov, d);
    if (rc == 0) {
        for (int k = 0; k < d; k++) {
            double s = 0;
            for (int l = 0; l < d; l++) s += cov[k * d + l] * rhs[l];
            mean[k] = s;
        }
        rc = chol_lower(cov, L, d);
    }
    if (rc == 0)
        for (int k = 0; k < d; k++) {
            double s = 0;
            for (int l = 0; l <= k; l++) s += L[k * d + l] * z[l];
            out[k] = s + mean[k];
        }
    free(A); free(cov); free(L); free(rhs); free(mean);
    return rc;
}

/* One half-sweep of samples(), bayes_pmf.py:280-296: every row of one side is
 * resampled against `oth` (CSR by row: ptr/idx/rat), z is (rows, d). */
int orc_gibbs_half_sweep(long rows, int d, const long *ptr, const long *idx,
                         const double *rat, const double *oth, const double *alpha,
                         const double *mu, double beta, double mean_sub, const double *z,
                         double *out)
{
    for (long r = 0; r < rows; r++) {
        int rc = orc_sample_feature(oth, d, idx + ptr[r], rat + ptr[r], ptr[r + 1] - ptr[r],
                                    alpha, mu, beta, mean_sub, z + r * d, out + r * d);
        if (rc) return rc;
    }
    return 0;
}

/* predict / pred_variance / prob_ge_cutoff over S posterior samples,
 * bayes_pmf.py:433-448, 528-538: per entry mean, np.var (ddof 0, two-pass) and
 * exceedance frequency of U_s V_s^T + add.  Any output may be NULL. */
void orc_sample_stats(const double *Us, const double *Vs, int S, int n, int m, int d,
                      double add, double cutoff, double *mean_nm, double *var_nm,
                      double *prob_nm)
{
    double *x = (double *)malloc(sizeof(double) * S);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            double tot = 0;
            long cnt = 0;
            for (int s = 0; s < S; s++) {
                x[s] = dot_seq(Us + ((long)s * n + i) * d, Vs + ((long)s * m + j) * d, d) + add;
                tot += x[s];
                cnt += (x[s] >= cutoff);
            }
            double mu = tot / S, v = 0;
            for (int s = 0; s < S; s++) v += (x[s] - mu) * (x[s] - mu);
            long o = (long)i * m + j;
            if (mean_nm) mean_nm[o] = mu;
            if (var_nm) var_nm[o] = v / S;
            if (prob_nm) prob_nm[o] = (double)cnt / (double)S;
        }
    free(x);
}

/* ------------------------------------------------------------------------- */
/* CPU-baseline helper: score a grid of candidates following the reference's   */
/* per-candidate path (pred_variance -> approx_pred_mean_var -> exp_dotprod_sq) */
/* on rows [row_lo, row_hi).  Returns the number of candidates scored.         */
/* ------------------------------------------------------------------------- */
long orc_mn_score_rows(int n, int m, int d, const double *mean, const double *R,
                       const double *C, int row_lo, int row_hi, double *var_out)
{
    int nm = n + m;
    long cnt = 0;
    for (int i = row_lo; i < row_hi; i++)
        for (int j = 0; j < m; j++) {
            double mn, var;
            orc_mn_pred_mean_var(n, mean, R, C, nm, d, i, j, &mn, &var, 0);
            var_out[(long)(i - row_lo) * m + j] = var;
            cnt++;
        }
    return cnt;
}

/* The same per-candidate path with the rows dealt to `threads` OpenMP threads: the all-cores */
/* leg of bench.py's cpu_baseline.  Candidates are independent, so this is   */
/* what the reference's multiprocessing.Pool would give at best.                              */
long orc_mn_score_rows_omp(int n, int m, int d, const double *mean, const double *R,
                           const double *C, int row_lo, int row_hi, double *var_out, int threads)
{
    int nm = n + m;
    int i;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (i = row_lo; i < row_hi; i++)
        for (int j = 0; j < m; j++) {
            double mn, var;
            orc_mn_pred_mean_var(n, mean, R, C, nm, d, i, j, &mn, &var, 0);
            var_out[(long)(i - row_lo) * m + j] = var;
        }
    return (long)(row_hi - row_lo) * m;
}
