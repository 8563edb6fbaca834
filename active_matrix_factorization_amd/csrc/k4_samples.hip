// k4_samples.hip - statistics over S posterior samples of (U, V): BayesianPMF.predict /
// pred_variance / prob_ge_cutoff (bayes_pmf.py:433-448, 528-538), float64.
//
// The reference materialises S dense (n, m) products U_s V_s' and reduces over s with
// np.mean / np.var / a >= count.  Here every candidate (i, j) keeps running statistics of
//   x_s = U_s[i] . V_s[j] + add
// in registers; nothing of size S x n x m ever exists.  The contraction (2 d S flops per
// candidate, S ~ 128) makes this the one compute-bound kernel of the path, and it is a batch
// of small GEMMs: per sample a wave forms a 16-row x 32-column tile of U_s V_s' with
// v_mfma_f64_16x16x4_f64 (the latent dimension zero-padded to a multiple of 4), both operands
// gathered straight from the row-major sample arrays (lane l supplies row/column l & 15,
// k = 4 q + (l >> 4): a 16 x d block per operand, every cache line of it fully used), and
// folds the 8 values a lane receives (rows (l >> 4) + 4 reg, column l & 15 of each 16 x 16
// tile) into its running sums.
// Variance is accumulated around the first sample (shifted sums), which matches numpy's
// two-pass np.var to rounding.
#include "common.h"

namespace {

using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int CT = 2;   // 16 x 16 tiles per wave, side by side

__global__ __launch_bounds__(256) void k_sample_stats(int S, int n, int m, int d, int ld,
                                                      const double *__restrict__ Us,
                                                      const double *__restrict__ Vs, double add,
                                                      double cutoff, double *__restrict__ o_mean,
                                                      double *__restrict__ o_var,
                                                      double *__restrict__ o_prob)
{
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int ri = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.y * 16;
    const int col0 = (blockIdx.x * 4 + wave) * (16 * CT);
    if (col0 >= ld) return;   // wave-uniform

    // operand addresses are clamped into range and the value zeroed afterwards: every lane
    // issues every load
    const int arow = min(row0 + ri, n - 1);
    const bool arow_ok = row0 + ri < n;
    int bcol[CT];
    bool bcol_ok[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) {
        bcol[t] = min(col0 + 16 * t + ri, m - 1);
        bcol_ok[t] = col0 + 16 * t + ri < m;
    }

    double c[CT][4], s1[CT][4], s2[CT][4];
    int cnt[CT][4];
#pragma unroll
    for (int t = 0; t < CT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) { c[t][r] = 0.; s1[t][r] = 0.; s2[t][r] = 0.; cnt[t][r] = 0; }

    // The walk over (sample, chunk of 16 latent components) is software-pipelined: the
    // operands of the next chunk are in flight while the matrix core works on this one
    // (without it every sample costs a full L2 round trip: the kernel was latency-bound).
    const int nchunk = (d + 15) >> 4;
    auto load_chunk = [&](int s, int ch, double (&av)[4], double (&bv)[CT][4]) {
        const double *us = Us + ((size_t)s * n + arow) * d;
        const double *vs = Vs + (size_t)s * m * d;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = 16 * ch + 4 * q + kq;
            const int kc = min(k, d - 1);
            const bool kok = k < d;
            const double x = us[kc];
            av[q] = (kok && arow_ok) ? x : 0.;
#pragma unroll
            for (int t = 0; t < CT; t++) {
                const double y = vs[(size_t)bcol[t] * d + kc];
                bv[t][q] = (kok && bcol_ok[t]) ? y : 0.;
            }
        }
    };
    double ca[4], cb[CT][4], na[4], nb[CT][4];
    load_chunk(0, 0, ca, cb);
    f64x4 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) acc[t] = (f64x4){0., 0., 0., 0.};
    const int total = S * nchunk;
    int s = 0, ch = 0;
    for (int it = 0; it < total; it++) {
        int ns = s, nch = ch + 1;
        if (nch == nchunk) { nch = 0; ns = s + 1; }
        const bool more = it + 1 < total;          // the last step re-reads its own operands
        load_chunk(more ? ns : s, more ? nch : ch, na, nb);
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int t = 0; t < CT; t++)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[q], cb[t][q], acc[t], 0, 0, 0);
        if (ch == nchunk - 1) {
#pragma unroll
            for (int t = 0; t < CT; t++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double xv = acc[t][r] + add;
                    c[t][r] = (s == 0) ? xv : c[t][r];
                    const double dlt = xv - c[t][r];
                    s1[t][r] += dlt;
                    s2[t][r] = fma(dlt, dlt, s2[t][r]);
                    cnt[t][r] += (xv >= cutoff) ? 1 : 0;
                }
                acc[t] = (f64x4){0., 0., 0., 0.};
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ca[q] = na[q];
#pragma unroll
            for (int t = 0; t < CT; t++) cb[t][q] = nb[t][q];
        }
        s = ns;
        ch = nch;
    }
    const double invS = 1. / (double)S;
#pragma unroll
    for (int t = 0; t < CT; t++) {
        const int col = col0 + 16 * t + ri;     // < ld: ld is a multiple of 64
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = row0 + kq + 4 * r;
            if (row >= n) continue;
            const size_t o = (size_t)row * ld + col;
            const double md = s1[t][r] * invS;
            double var = s2[t][r] * invS - md * md;
            if (var < 0.) var = 0.;
            const bool pad = col >= m;          // padding columns stay zero
            if (o_mean) o_mean[o] = pad ? 0. : c[t][r] + md;
            if (o_var) o_var[o] = pad ? 0. : var;
            if (o_prob) o_prob[o] = pad ? 0. : (double)cnt[t][r] / (double)S;
        }
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
    dim3 grid((ld + 64 * CT - 1) / (64 * CT), (n + 15) / 16);
    {
        ProfScope prof(ctx);
        hipLaunchKernelGGL(k_sample_stats, grid, dim3(256), 0, ctx->stream, S, n, m, d, ld, d_Us, d_Vs, add,
                           cutoff, d_mean, d_var, d_prob);
    }
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}
