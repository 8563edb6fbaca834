// k7_hyper.hip - the Gaussian-Wishart hyper-parameter step of the Bayesian-PMF Gibbs sampler on
// the device: BayesianPMF.sample_hyperparam + sample_wishart
// (bayes_pmf.py:157-186, 41-59), so that a samples() yield has no host hop between its sweeps.
//
//   x_bar = mean(F), S_bar = cov(F) (ddof 1),
//   WI    = inv( inv(W0) + N S_bar + b0 N / (b0 + N) * <mu0 - x_bar, mu0 - x_bar> )      [sic: the reference
//           takes np.dot of two 1-D vectors - a SCALAR, added to every element - bayes_pmf.py:171-176]
//   WI    = WI / 2 + (WI / 2)'
//   alpha ~ Wishart(WI, df + N)   by Bartlett's decomposition:  X = chol(WI) T,  alpha = X X',
//           T lower triangular, T_ii = sqrt(chi2(dof - i)), T_ij ~ N(0, 1) below the diagonal
//   mu    = chol( inv((b0 + N) alpha) ) z + (b0 mu0 + N x_bar) / (b0 + N),   z ~ N(0, 1)^d
//
// One block.  The statistics of F are fixed-order reductions (deterministic); the d x d algebra runs
// in LDS, one lane per row (d <= 32).  The random variates are either SUPPLIED (chi-squares, normals
// for T in row-major order of the strict lower triangle, z - what NumPy draws in this order, so the
// arithmetic can be checked exactly against the host) or generated here from the Philox4x32-10 stream
// BayesianPMF's device generator uses (normals by Box-Muller, chi-squares by Marsaglia-Tsang).
// The reference's small-dof branch of sample_wishart (a direct sum of dof outer products) is a
// different way to draw from the same distribution; the device always uses Bartlett.
#include "common.h"

namespace {

constexpr int HY_MAXD = 32;

__device__ __forceinline__ void philox7(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// counter -> (N(0,1), N(0,1), U(0,1])
__device__ __forceinline__ void draw(uint64_t seed, uint64_t ctr, double &n0, double &n1, double &u)
{
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x13198A2Eu, 0x03707344u};
    philox7(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t a = (((uint64_t)c[0] << 32) | c[1]) >> 11, b = (((uint64_t)c[2] << 32) | c[3]) >> 11;
    const double u1 = ((double)a + 1.) * (1. / 9007199254740992.), u2 = ((double)b + 1.) * (1. / 9007199254740992.);
    const double rad = sqrt(-2. * log(u1)), ang = 6.283185307179586476925 * u2;
    n0 = rad * cos(ang);
    n1 = rad * sin(ang);
    u = u2;       // reused as the acceptance uniform: independent of rad
}

// chi-square with k > 1 degrees of freedom = 2 Gamma(k / 2) (Marsaglia & Tsang 2000); counters ctr, ctr+1, ...
__device__ double chi_square(double k, uint64_t seed, uint64_t &ctr)
{
    const double a = k / 2., dd = a - 1. / 3., cc = 1. / sqrt(9. * dd);
    for (int it = 0; it < 64; it++) {
        double x, y, u;
        draw(seed, ctr++, x, y, u);
        double v = 1. + cc * x;
        if (v <= 0.) continue;
        v = v * v * v;
        // the acceptance uniform must be independent of x: take it from a fresh counter
        double x2, y2, u2;
        draw(seed, ctr++, x2, y2, u2);
        if (log(u2) < .5 * x * x + dd - dd * v + dd * log(v)) return 2. * dd * v;
    }
    return 2. * dd;      // unreachable in practice (acceptance > 95 % per round)
}

// ---- d x d helpers in LDS (leading dimension P), executed by the whole block; lane = row --------
__device__ bool chol_block(double *A, int d, int P, int t)
{
    bool ok = true;
    for (int j = 0; j < d; j++) {
        __syncthreads();
        const double piv = A[j * P + j];
        if (!(piv > 0.)) { ok = false; break; }
        const double s = sqrt(piv);
        __syncthreads();
        if (t == j) A[j * P + j] = s;
        if (t > j && t < d) A[t * P + j] /= s;
        __syncthreads();
        if (t > j && t < d) {
            const double lij = A[t * P + j];
            for (int c = j + 1; c <= t; c++) A[t * P + c] -= lij * A[c * P + j];
        }
    }
    __syncthreads();
    return ok;
}

// Ainv <- (L L')^-1 from the lower factor in L; X is scratch
__device__ void inv_from_chol(const double *L, double *X, double *Ainv, int d, int P, int t)
{
    if (t < d) {       // column t of X = L^-1
        for (int i = 0; i < d; i++) {
            double v = (i == t) ? 1. : 0.;
            for (int k = t; k < i; k++) v -= L[i * P + k] * X[k * P + t];
            X[i * P + t] = i < t ? 0. : v / L[i * P + i];
        }
    }
    __syncthreads();
    if (t < d)
        for (int j = 0; j < d; j++) {
            double v = 0;
            for (int k = (t > j ? t : j); k < d; k++) v += X[k * P + t] * X[k * P + j];
            Ainv[t * P + j] = v;
        }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_hyper(const double *__restrict__ F, int N, int d,
                                               const double *__restrict__ W0, double b0, double df,
                                               const double *__restrict__ mu0,
                                               const double *__restrict__ variates, uint64_t seed, uint64_t offset,
                                               double *__restrict__ mu_out, double *__restrict__ alpha_out,
                                               int *__restrict__ status)
{
    extern __shared__ double sm[];
    const int P = d + 1, t = threadIdx.x;
    double *xbar = sm;                 // d
    double *part = xbar + HY_MAXD;     // 8 x 32 partials
    double *A = part + 8 * HY_MAXD;    // d x P each
    double *B = A + d * P, *C = B + d * P, *T = C + d * P;
    double *vec = T + d * P;           // d: z, then misc
    __shared__ int bad;
    if (t == 0) bad = 0;
    // ---- x_bar: thread (g, k) sums rows g, g + 8, ... of column k
    {
        const int k = t & 31, g = t >> 5;
        double s = 0.;
        if (k < d)
            for (int r = g; r < N; r += 8) s += F[(size_t)r * d + k];
        part[g * HY_MAXD + k] = s;
        __syncthreads();
        if (t < d) {
            double tot = 0.;
            for (int gg = 0; gg < 8; gg++) tot += part[gg * HY_MAXD + t];
            xbar[t] = tot / N;
        }
        __syncthreads();
    }
    // ---- N * S_bar = N / (N - 1) * sum (x - x_bar)(x - x_bar)'   into B
    for (int e = t; e < d * d; e += 256) {
        const int k = e / d, l = e % d;
        const double xk = xbar[k], xl = xbar[l];
        double s = 0.;
        for (int r = 0; r < N; r++) s += (F[(size_t)r * d + k] - xk) * (F[(size_t)r * d + l] - xl);
        B[k * P + l] = s / (N - 1.) * N;
    }
    // ---- inv(W0) into A (via C, T)
    for (int e = t; e < d * d; e += 256) C[(e / d) * P + e % d] = W0[e];
    __syncthreads();
    if (!chol_block(C, d, P, t) && t == 0) bad = 1;
    inv_from_chol(C, T, A, d, P, t);
    // ---- A <- inv(W0) + N S_bar + b0 N / (b0 + N) * <mu0 - x_bar, mu0 - x_bar>  (scalar, every element)
    double sc = 0.;
    for (int k = 0; k < d; k++) sc += (mu0[k] - xbar[k]) * (mu0[k] - xbar[k]);
    sc *= (b0 * N) / (b0 + N);
    for (int e = t; e < d * d; e += 256) {
        const int o = (e / d) * P + e % d;
        A[o] = A[o] + B[o] + sc;
    }
    __syncthreads();
    // ---- WI = inv(A); WI = WI / 2 + (WI / 2)'   into B
    if (!chol_block(A, d, P, t) && t == 0) bad = 1;
    inv_from_chol(A, T, C, d, P, t);
    for (int e = t; e < d * d; e += 256) {
        const int k = e / d, l = e % d;
        B[k * P + l] = C[k * P + l] / 2. + C[l * P + k] / 2.;
    }
    __syncthreads();
    // ---- chol(WI) in B
    if (!chol_block(B, d, P, t) && t == 0) bad = 1;
    // ---- Bartlett factor T: variates supplied [chi (d) | normals (d (d - 1) / 2) | z (d)] or generated
    const double dof = df + N;
    const int nlow = d * (d - 1) / 2;
    for (int e = t; e < d * d; e += 256) T[(e / d) * P + e % d] = 0.;
    __syncthreads();
    if (variates) {
        if (t < d) T[t * P + t] = sqrt(variates[t]);
        for (int e = t; e < nlow; e += 256) {
            int i = (int)((1. + sqrt(1. + 8. * e)) / 2.);
            while (i * (i - 1) / 2 > e) i--;
            while ((i + 1) * i / 2 <= e) i++;
            T[i * P + (e - i * (i - 1) / 2)] = variates[d + e];
        }
        if (t < d) vec[t] = variates[d + nlow + t];
    } else {
        // counters: [0, 4 d) chi-squares (thread 0, sequential rejection), then one per normal pair
        if (t == 0) {
            uint64_t ctr = offset;
            for (int i = 0; i < d; i++) T[i * P + i] = sqrt(chi_square(dof - i, seed, ctr));
        }
        const uint64_t base = offset + 256;       // far beyond what the rejection loops can consume
        for (int e = t; e < (nlow + 1) / 2; e += 256) {
            double a, b, u;
            draw(seed, base + e, a, b, u);
            for (int q = 0; q < 2; q++) {
                const int idx = 2 * e + q;
                if (idx >= nlow) break;
                int i = (int)((1. + sqrt(1. + 8. * idx)) / 2.);
                while (i * (i - 1) / 2 > idx) i--;
                while ((i + 1) * i / 2 <= idx) i++;
                T[i * P + (idx - i * (i - 1) / 2)] = q ? b : a;
            }
        }
        if (t < (d + 1) / 2) {
            double a, b, u;
            draw(seed, base + (uint64_t)(nlow + 1) / 2 + t, a, b, u);
            vec[2 * t] = a;
            if (2 * t + 1 < d) vec[2 * t + 1] = b;
        }
    }
    __syncthreads();
    // ---- X = chol(WI) T   into C;  alpha = X X'   into A
    for (int e = t; e < d * d; e += 256) {
        const int i = e / d, j = e % d;
        double v = 0.;
        for (int k = j; k <= i; k++) v += B[i * P + k] * T[k * P + j];
        C[i * P + j] = v;
    }
    __syncthreads();
    for (int e = t; e < d * d; e += 256) {
        const int i = e / d, j = e % d;
        double v = 0.;
        const int kmax = i < j ? i : j;
        for (int k = 0; k <= kmax; k++) v += C[i * P + k] * C[j * P + k];
        A[i * P + j] = v;
        alpha_out[e] = v;
    }
    __syncthreads();
    // ---- lam = chol(inv((b0 + N) alpha));  mu = lam z + (b0 mu0 + N x_bar) / (b0 + N)
    for (int e = t; e < d * d; e += 256) B[(e / d) * P + e % d] = (b0 + N) * A[(e / d) * P + e % d];
    __syncthreads();
    if (!chol_block(B, d, P, t) && t == 0) bad = 1;
    inv_from_chol(B, T, C, d, P, t);
    if (!chol_block(C, d, P, t) && t == 0) bad = 1;
    if (t < d) {
        double v = (b0 * mu0[t] + N * xbar[t]) / (b0 + N);
        for (int k = 0; k <= t; k++) v += C[t * P + k] * vec[k];
        mu_out[t] = v;
    }
    __syncthreads();
    if (t == 0 && bad) *status = 1;
}

} // namespace

extern "C" int apmf_gibbs_sample_hyperparam(apmf_ctx *ctx, const double *d_feats, int N, int d, const double *d_wi,
                                            double b0, double df, const double *d_mu0, const double *d_variates,
                                            uint64_t seed, uint64_t offset, double *d_mu, double *d_alpha,
                                            int *d_status)
{
    APMF_REQUIRE(ctx && d_feats && d_wi && d_mu0 && d_mu && d_alpha, "NULL operand");
    APMF_REQUIRE(N >= 2, "need at least two feature rows (np.cov)");
    APMF_REQUIRE(d >= 1 && d <= HY_MAXD, "latent_d=%d: the device hyper-parameter step supports 1..%d", d, HY_MAXD);
    // d_status (device int, zeroed by the caller, nullable): set to 1 if a matrix of the step was not
    // positive definite.  Asynchronous: no host round trip here, the caller checks once per yield.
    int *status_ptr = d_status;
    if (!status_ptr) {
        void *w;
        APMF_TRY(apmf_ws(ctx, 2, 64, &w));
        status_ptr = (int *)w;
        APMF_CHECK_HIP(hipMemsetAsync(status_ptr, 0, sizeof(int), ctx->stream));
    }
    const size_t smem = (HY_MAXD + 8 * HY_MAXD + 4 * (size_t)d * (d + 1) + HY_MAXD) * sizeof(double);
    hipLaunchKernelGGL(k_hyper, dim3(1), dim3(256), smem, ctx->stream, d_feats, N, d, d_wi, b0, df, d_mu0, d_variates,
                       seed, offset, d_mu, d_alpha, status_ptr);
    APMF_CHECK_LAUNCH();
    if (d_status) return APMF_OK;
    int status = 0;
    APMF_TRY(apmf_d2h(ctx, &status, status_ptr, sizeof(int)));
    if (status) {
        apmf_set_error("a matrix of the hyper-parameter step is not positive definite");
        return APMF_ENOTPD;
    }
    return APMF_OK;
}
