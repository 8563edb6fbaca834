// k4_samples.hip - statistics over S posterior samples of (U, V): BayesianPMF.predict /
// pred_variance / prob_ge_cutoff (bayes_pmf.py:433-448, 528-538), float64.
//
// The reference materialises S dense (n, m) products U_s V_s' and reduces over s with
// np.mean / np.var / a >= count.  Here every candidate (i, j) keeps running statistics of
//   x_s = U_s[i] . V_s[j] + add
// in registers; nothing of size S x n x m ever exists.  The contraction (2 d S flops per
// candidate, S ~ 128) makes this the one compute-bound kernel of the path: each thread
// owns a RT x 2 micro-tile (RT rows, 2 adjacent columns) so that every V operand fetched
// (coalesced, from a transposed copy Vt[s][k][j]) feeds RT fused multiply-adds and every
// U operand is a wave-uniform scalar.
// Variance is accumulated around the first sample (shifted sums), which matches numpy's
// two-pass np.var to rounding.
#include "common.h"

namespace {

constexpr int RT = 8;

// Vt[s][k][j] = Vs[s][j][k], pitch ld, padding zeroed
__global__ __launch_bounds__(256) void k_transpose_v(int S, int m, int d, int ld,
                                                     const double *__restrict__ Vs,
                                                     double *__restrict__ Vt)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (j >= ld) return;
    for (int k = 0; k < d; k++)
        Vt[((size_t)s * d + k) * ld + j] = j < m ? Vs[((size_t)s * m + j) * d + k] : 0.;
}

__global__ __launch_bounds__(256) void k_sample_stats(int S, int n, int m, int d, int ld,
                                                      const double *__restrict__ Us,
                                                      const double *__restrict__ Vt, double add,
                                                      double cutoff, double *__restrict__ o_mean,
                                                      double *__restrict__ o_var,
                                                      double *__restrict__ o_prob)
{
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int j0 = (blockIdx.x * 64 + lane) * 2;
    const int i0 = (blockIdx.y * 4 + wave) * RT;
    if (i0 >= n) return;  // wave-uniform
    const bool colok = j0 < ld;
    double c[RT][2], s1[RT][2], s2[RT][2];
    int cnt[RT][2];
#pragma unroll
    for (int r = 0; r < RT; r++)
#pragma unroll
        for (int v = 0; v < 2; v++) { c[r][v] = 0.; s1[r][v] = 0.; s2[r][v] = 0.; cnt[r][v] = 0; }

    for (int s = 0; s < S; s++) {
        double x[RT][2];
#pragma unroll
        for (int r = 0; r < RT; r++) x[r][0] = x[r][1] = 0.;
        const double *vt = Vt + (size_t)s * d * ld + j0;
        const double *us = Us + ((size_t)s * n + i0) * d;  // wave-uniform
        for (int k = 0; k < d; k++) {
            double2 v = colok ? *reinterpret_cast<const double2 *>(vt + (size_t)k * ld) : make_double2(0., 0.);
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const double u = (i0 + r < n) ? us[(size_t)r * d + k] : 0.;
                x[r][0] += u * v.x;
                x[r][1] += u * v.y;
            }
        }
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
            for (int v = 0; v < 2; v++) {
                const double xv = x[r][v] + add;
                if (s == 0) c[r][v] = xv;
                const double dlt = xv - c[r][v];
                s1[r][v] += dlt;
                s2[r][v] += dlt * dlt;
                cnt[r][v] += (xv >= cutoff) ? 1 : 0;
            }
    }
    if (!colok) return;
    const double invS = 1. / (double)S;
#pragma unroll
    for (int r = 0; r < RT; r++) {
        if (i0 + r >= n) break;
        const size_t o = (size_t)(i0 + r) * ld + j0;
        double mu[2], var[2], pr[2];
#pragma unroll
        for (int v = 0; v < 2; v++) {
            const double md = s1[r][v] * invS;
            mu[v] = c[r][v] + md;
            var[v] = s2[r][v] * invS - md * md;
            if (var[v] < 0.) var[v] = 0.;
            pr[v] = (double)cnt[r][v] / (double)S;
        }
        if (o_mean) *reinterpret_cast<double2 *>(o_mean + o) = make_double2(mu[0], mu[1]);
        if (o_var) *reinterpret_cast<double2 *>(o_var + o) = make_double2(var[0], var[1]);
        if (o_prob) *reinterpret_cast<double2 *>(o_prob + o) = make_double2(pr[0], pr[1]);
    }
}

} // namespace

extern "C" int apmf_sample_stats_f64(apmf_ctx *ctx, int S, int n, int m, int d, int ld,
                                     const double *d_Us, const double *d_Vs, double add,
                                     double cutoff, double *d_mean, double *d_var, double *d_prob)
{
    APMF_REQUIRE(ctx && d_Us && d_Vs, "NULL operand");
    APMF_REQUIRE(S >= 1 && n >= 1 && m >= 1 && d >= 1, "bad shape");
    APMF_REQUIRE(ld >= m && ld % APMF_GRID_ALIGN == 0, "ld=%d must be >= m and a multiple of %d", ld,
                 APMF_GRID_ALIGN);
    if (!d_mean && !d_var && !d_prob) return APMF_OK;
    void *w;
    APMF_TRY(apmf_ws(ctx, 0, (size_t)S * d * ld * sizeof(double), &w));
    double *Vt = (double *)w;
    hipLaunchKernelGGL(k_transpose_v, dim3((ld + 255) / 256, S), dim3(256), 0, ctx->stream, S, m, d, ld,
                       d_Vs, Vt);
    APMF_CHECK_LAUNCH();
    dim3 grid((ld + 127) / 128, (n + 4 * RT - 1) / (4 * RT));
    hipLaunchKernelGGL(k_sample_stats, grid, dim3(256), 0, ctx->stream, S, n, m, d, ld, d_Us, Vt, add,
                       cutoff, d_mean, d_var, d_prob);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}
