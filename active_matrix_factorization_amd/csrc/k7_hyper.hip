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
constexpr int HY_STAGE = 4096;   // doubles of the feature matrix staged per pass (a multiple of 8 rows)

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

// lane `src` 's value of a wave-resident double, as a wave-uniform value
__device__ __forceinline__ double lane_value7(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// d <= 16: the same two helpers with the matrix in the REGISTERS of wave 0 (lane = row, other lanes'
// entries by v_readlane; the same operations on every element in the same order).  The LDS loops
// below are chains of dependent LDS round trips - 22 us per 15 x 15 factorisation, and this kernel
// runs five of them and three inversions.
__device__ bool chol_block16(double *A, int d, int P, int t)
{
    __shared__ int s_ok;
    __syncthreads();
    if (t < 64) {
        double r[16];
#pragma unroll
        for (int k = 0; k < 16; k++) r[k] = (t < d && k < d) ? A[t * P + k] : (k == t ? 1. : 0.);
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const double piv = lane_value7(r[j], j);
            ok = ok && piv > 0.;
            const double sq = sqrt(piv);
            r[j] = t == j ? sq : r[j] / sq;
#pragma unroll
            for (int c = j + 1; c < 16; c++) r[c] -= r[j] * lane_value7(r[j], c);
        }
        if (t < d)
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k <= t) A[t * P + k] = r[k];
        if (t == 0) s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return s_ok != 0;
}

__device__ void inv_from_chol16(const double *L, double *Ainv, int d, int P, int t)
{
    __syncthreads();
    if (t < 64) {
        double r[16], x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) r[k] = (t < d && k <= t) ? L[t * P + k] : (k == t ? 1. : 0.);
#pragma unroll
        for (int i = 0; i < 16; i++) {       // lane c owns column c of X = L^-1
            double v = i == t ? 1. : 0.;
#pragma unroll
            for (int m = 0; m < i; m++) v -= lane_value7(r[m], i) * x[m];
            x[i] = i < t ? 0. : v / lane_value7(r[i], i);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {       // (X' X)[t][j]; terms below max(t, j) are zero
            double v = 0.;
#pragma unroll
            for (int k = 0; k < 16; k++) v += x[k] * lane_value7(x[k], j);
            if (t < d && j < d) Ainv[t * P + j] = v;
        }
    }
    __syncthreads();
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

__device__ __forceinline__ bool chol_any(double *A, int d, int P, int t)
{
    return d <= 16 ? chol_block16(A, d, P, t) : chol_block(A, d, P, t);
}

__device__ __forceinline__ void inv_any(const double *L, double *X, double *Ainv, int d, int P, int t)
{
    if (d <= 16) inv_from_chol16(L, Ainv, d, P, t);
    else inv_from_chol(L, X, Ainv, d, P, t);
}

// blockIdx.x = chain (apmf_gibbs_sample_hyperparam_batched: H feature matrices, H sets of supplied variates, H
// outputs; the hyper-priors are shared).  direct_dof > 0: the reference's small-dof branch of sample_wishart
// (bayes_pmf.py:47-52: X = chol(WI) G with G a d x dof matrix of normals) instead of Bartlett's factor; the variates
// are then [G (d x dof, row-major) | z (d)].
__global__ __launch_bounds__(256) void k_hyper(const double *__restrict__ F, int N, int d,
                                               const double *__restrict__ W0, double b0, double df,
                                               const double *__restrict__ mu0,
                                               const double *__restrict__ variates, uint64_t seed, uint64_t offset,
                                               double *__restrict__ mu_out, double *__restrict__ alpha_out,
                                               int *__restrict__ status, int direct_dof = 0, size_t nvar = 0)
{
    extern __shared__ double sm[];
    const int P = d + 1, t = threadIdx.x;
    {
        const size_t hb = blockIdx.x;
        F += hb * (size_t)N * d;
        if (variates) variates += hb * nvar;
        mu_out += hb * d;
        alpha_out += hb * (size_t)d * d;
    }
    double *xbar = sm;                 // d
    double *part = xbar + HY_MAXD;     // 8 x 32 partials
    double *A = part + 8 * HY_MAXD;    // d x P each
    double *B = A + d * P, *C = B + d * P, *T = C + d * P;
    double *vec = T + d * P;           // d: z, then misc
    __shared__ int bad;
    if (t == 0) bad = 0;
    // The two passes over the feature matrix go through LDS, HY_ROWS rows at a time: the block fetches a
    // chunk with all its loads in flight (the next chunk while this one is summed) and the sums then
    // walk LDS in the same row order as before.  Read from global memory inside the summation loops
    // these were N dependent round trips per thread: 300 us of a 312 us kernel at N = 1682.
    double *stage = vec + HY_MAXD;     // HY_ROWS x d
    const int HY_ROWS = HY_STAGE / d / 8 * 8;
    constexpr int SQ = HY_STAGE / 256;
    double sv[SQ];
    auto chunk_load = [&](int r0) {
        const int lim = min(HY_ROWS, N - r0) * d;
#pragma unroll
        for (int q = 0; q < SQ; q++) {
            const int e = t + 256 * q;
            const double x = F[(size_t)r0 * d + (e < lim ? e : 0)];
            sv[q] = e < lim ? x : 0.;
        }
    };
    auto chunk_store = [&]() {
#pragma unroll
        for (int q = 0; q < SQ; q++) {
            const int e = t + 256 * q;
            if (e < HY_ROWS * d) stage[e] = sv[q];
        }
    };
    // ---- x_bar: thread (g, k) sums rows g, g + 8, ... of column k
    {
        const int k = t & 31, g = t >> 5;
        double s = 0.;
        chunk_load(0);
        for (int r0 = 0; r0 < N; r0 += HY_ROWS) {
            __syncthreads();
            chunk_store();
            if (r0 + HY_ROWS < N) chunk_load(r0 + HY_ROWS);
            __syncthreads();
            const int rows = min(HY_ROWS, N - r0);
            if (k < d)
#pragma unroll 8
                for (int rr = g; rr < rows; rr += 8) s += stage[rr * d + k];
        }
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
    {
        const int k = t / d, l = t % d;          // d * d <= 1024: up to four entries per thread
        double s4[4] = {0., 0., 0., 0.};
        chunk_load(0);
        for (int r0 = 0; r0 < N; r0 += HY_ROWS) {
            __syncthreads();
            chunk_store();
            if (r0 + HY_ROWS < N) chunk_load(r0 + HY_ROWS);
            __syncthreads();
            const int rows = min(HY_ROWS, N - r0);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = t + 256 * u;
                if (e < d * d) {
                    const int kk = u == 0 ? k : e / d, ll = u == 0 ? l : e % d;
                    const double xk = xbar[kk], xl = xbar[ll];
                    double s = s4[u];
#pragma unroll 8
                    for (int rr = 0; rr < rows; rr++) s += (stage[rr * d + kk] - xk) * (stage[rr * d + ll] - xl);
                    s4[u] = s;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = t + 256 * u;
            if (e < d * d) B[(e / d) * P + e % d] = s4[u] / (N - 1.) * N;
        }
    }
    // ---- inv(W0) into A (via C, T)
    for (int e = t; e < d * d; e += 256) C[(e / d) * P + e % d] = W0[e];
    __syncthreads();
    if (!chol_any(C, d, P, t) && t == 0) bad = 1;
    inv_any(C, T, A, d, P, t);
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
    if (!chol_any(A, d, P, t) && t == 0) bad = 1;
    inv_any(A, T, C, d, P, t);
    for (int e = t; e < d * d; e += 256) {
        const int k = e / d, l = e % d;
        B[k * P + l] = C[k * P + l] / 2. + C[l * P + k] / 2.;
    }
    __syncthreads();
    // ---- chol(WI) in B
    if (!chol_any(B, d, P, t) && t == 0) bad = 1;
    // ---- Bartlett factor T: variates supplied [chi (d) | normals (d (d - 1) / 2) | z (d)] or generated
    const double dof = df + N;
    const int nlow = d * (d - 1) / 2;
    for (int e = t; e < d * d; e += 256) T[(e / d) * P + e % d] = 0.;
    __syncthreads();
    if (variates && direct_dof > 0) {
        // X = chol(WI) G into the staging area (the feature matrix is done with it): d x dof <= HY_STAGE
        for (int e = t; e < d * direct_dof; e += 256) {
            const int i = e / direct_dof, c = e % direct_dof;
            double v = 0.;
            for (int k = 0; k <= i; k++) v += B[i * P + k] * variates[(size_t)k * direct_dof + c];
            stage[e] = v;
        }
        if (t < d) vec[t] = variates[(size_t)d * direct_dof + t];
    } else if (variates) {
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
        // counters (< 1024 per call, see apmf.h): chi-square i starts at offset + 16 i, one thread each (its
        // rejection loop takes two per round and accepts > 95 % of the rounds: running into the next
        // one's range needs 8 rejections in a row); the normals start at offset + 512, one per pair
        if (t < d) {
            uint64_t ctr = offset + 16 * (uint64_t)t;
            T[t * P + t] = sqrt(chi_square(dof - t, seed, ctr));
        }
        const uint64_t base = offset + 16 * (uint64_t)HY_MAXD;
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
    // ---- X = chol(WI) T   into C;  alpha = X X'   into A   (direct scheme: X is d x dof, in the staging area)
    if (variates && direct_dof > 0) {
        for (int e = t; e < d * d; e += 256) {
            const int i = e / d, j = e % d;
            double v = 0.;
            for (int c = 0; c < direct_dof; c++) v += stage[i * direct_dof + c] * stage[j * direct_dof + c];
            A[i * P + j] = v;
            alpha_out[e] = v;
        }
    } else {
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
    }
    __syncthreads();
    // ---- lam = chol(inv((b0 + N) alpha));  mu = lam z + (b0 mu0 + N x_bar) / (b0 + N)
    for (int e = t; e < d * d; e += 256) B[(e / d) * P + e % d] = (b0 + N) * A[(e / d) * P + e % d];
    __syncthreads();
    if (!chol_any(B, d, P, t) && t == 0) bad = 1;
    inv_any(B, T, C, d, P, t);
    if (!chol_any(C, d, P, t) && t == 0) bad = 1;
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
    const size_t smem = (HY_MAXD + 8 * HY_MAXD + 4 * (size_t)d * (d + 1) + HY_MAXD + (size_t)HY_STAGE) * sizeof(double);
    if (smem > 64 * 1024)
        APMF_CHECK_HIP(hipFuncSetAttribute((const void *)k_hyper, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
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

// H chains at once with SUPPLIED variates (the batched exp_variance lookahead, bayes_pmf.py:457-525: every
// hypothesis' Gibbs chain takes its hyper-parameter step here instead of on the host, with the random numbers NumPy
// drew for it): d_feats (H, N, d), d_variates (H, nvar), d_mu (H, d), d_alpha (H, d, d).  direct_dof = 0: Bartlett
// variates [chi (d) | normals (d (d - 1) / 2) | z (d)]; > 0: the small-dof scheme, [G (d x dof) | z (d)].
extern "C" int apmf_gibbs_sample_hyperparam_batched(apmf_ctx *ctx, int H, const double *d_feats, int N, int d,
                                                    const double *d_wi, double b0, double df, const double *d_mu0,
                                                    const double *d_variates, int direct_dof, double *d_mu,
                                                    double *d_alpha, int *d_status)
{
    APMF_REQUIRE(ctx && d_feats && d_wi && d_mu0 && d_variates && d_mu && d_alpha && d_status, "NULL operand");
    APMF_REQUIRE(H >= 1 && N >= 2, "need at least one chain and two feature rows (np.cov)");
    APMF_REQUIRE(d >= 1 && d <= HY_MAXD, "latent_d=%d: the device hyper-parameter step supports 1..%d", d, HY_MAXD);
    APMF_REQUIRE(direct_dof >= 0 && (size_t)d * direct_dof <= (size_t)HY_STAGE, "direct Wishart scheme: d x dof = %d x %d "
                 "exceeds the staging area", d, direct_dof);
    const size_t nvar = direct_dof > 0 ? (size_t)d * direct_dof + d : (size_t)d + (size_t)d * (d - 1) / 2 + d;
    const size_t smem = (HY_MAXD + 8 * HY_MAXD + 4 * (size_t)d * (d + 1) + HY_MAXD + (size_t)HY_STAGE) * sizeof(double);
    if (smem > 64 * 1024)
        APMF_CHECK_HIP(hipFuncSetAttribute((const void *)k_hyper, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_hyper, dim3(H), dim3(256), smem, ctx->stream, d_feats, N, d, d_wi, b0, df, d_mu0, d_variates,
                       (uint64_t)0, (uint64_t)0, d_mu, d_alpha, d_status, direct_dof, nvar);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}
