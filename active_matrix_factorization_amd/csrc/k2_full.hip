// k2_full.hip - candidate-grid scoring under the full-covariance Gaussian of ActivePMF
// (active_pmf.py:301-322, 392-400, 502-524; normal_exps_cy.exp_dotprod_sq,
// normal_exps_cy.pyx:111-135), float64.
//
// With u = U_i ~ N(a, A), v = V_j ~ N(b, Bm), Cov(u_k, v_l) = C_kl (the d x d blocks of the
// (n+m)d x (n+m)d covariance at the reference's index maps u[k,i] = i d + k,
// v[k,j] = n d + j d + k), the reference's Isserlis sum minus mean^2 equals
//     Var = tr(A Bm) + tr(C C) + a'Bm a + b'A b + 2 b'C a        E = a'b + tr(C)
// The reference reads A and Bm through their upper triangles (k <= l) - mirrored here - and
// C from both off-diagonal blocks, which agree for the symmetric matrices project_psd emits;
// this kernel reads the upper-right block.
// Per candidate the d x d cross block C is the streamed operand (8 d^2 bytes); for a fixed
// user row the d rows of the cross block are contiguous over j, so a wave reads them
// coalesced.  One block = one user row x 256 items; A_i and a_i sit in LDS.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void k_full_grid(int n, int m, int d, int ld, const double *__restrict__ mean,
                                                   const double *__restrict__ cov,
                                                   const uint8_t *__restrict__ mask,
                                                   double *__restrict__ o_var, double *__restrict__ o_mean,
                                                   int want, int want_sum, double *__restrict__ part_sum,
                                                   double *__restrict__ part_val,
                                                   long long *__restrict__ part_idx)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *sA = sm;           // d*d mirrored
    double *sa = sm + d * d;   // d
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const size_t K = (size_t)(n + m) * d;
    for (int t = threadIdx.x; t < d * d; t += 256) {
        const int k = t / d, l = t % d;
        const int kk = min(k, l), ll = max(k, l);
        sA[t] = cov[((size_t)i * d + kk) * K + (size_t)i * d + ll];
    }
    for (int t = threadIdx.x; t < d; t += 256) sa[t] = mean[(size_t)i * d + t];
    __syncthreads();

    double var = 0., mn = 0.;
    const bool in = j < m;
    if (in) {
        const size_t vb = (size_t)n * d + (size_t)j * d;  // first index of v_j
        const double *b = mean + vb;
        double trAB = 0., trCC = 0., aBa = 0., bAb = 0., bCa = 0., ab = 0., trC = 0.;
        for (int k = 0; k < d; k++) {
            const double *Crow = cov + ((size_t)i * d + k) * K + vb;     // C[k][.]
            double Ba_k = 0., Ab_k = 0., Ca_k = 0.;
            for (int l = 0; l < d; l++) {
                const int kk = min(k, l), ll = max(k, l);
                const double Bkl = cov[(vb + kk) * K + vb + ll];
                const double Ckl = Crow[l];
                const double Clk = cov[((size_t)i * d + l) * K + vb + k];
                trAB += sA[k * d + l] * Bkl;
                trCC += Ckl * Clk;
                Ba_k += Bkl * sa[l];
                Ab_k += sA[k * d + l] * b[l];
                Ca_k += Ckl * sa[l];
            }
            aBa += sa[k] * Ba_k;
            bAb += b[k] * Ab_k;
            bCa += b[k] * Ca_k;
            ab += sa[k] * b[k];
            trC += Crow[k];
        }
        var = trAB + trCC + aBa + bAb + 2. * bCa;
        mn = ab + trC;
        if (o_var) o_var[(size_t)i * ld + j] = var;
        if (o_mean) o_mean[(size_t)i * ld + j] = mn;
    } else if (j < ld) {
        if (o_var) o_var[(size_t)i * ld + j] = 0.;
        if (o_mean) o_mean[(size_t)i * ld + j] = 0.;
    }
    if (want || want_sum) {
        const bool cand = in && (!mask || mask[(size_t)i * ld + j]);
        double s = cand ? var : 0.;
        double bv = var;
        long long bi = cand ? (long long)i * m + j : -1;
        s = wave_sum(s);
        if (want) wave_argext(bv, bi, want);
        __shared__ double sh_s[4], sh_v[4];
        __shared__ long long sh_i[4];
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sh_s[wave] = s; sh_v[wave] = bv; sh_i[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; w++) {
                s += sh_s[w];
                if (want) argext_merge(bv, bi, sh_v[w], sh_i[w], want);
            }
            const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            part_sum[slot] = s; part_val[slot] = bv; part_idx[slot] = bi;
        }
    }
}

__global__ __launch_bounds__(256) void k_full_finish(int nblk, int want, const double *__restrict__ part_sum,
                                                     const double *__restrict__ part_val,
                                                     const long long *__restrict__ part_idx,
                                                     double *__restrict__ out_sum, double *__restrict__ out_val,
                                                     int64_t *__restrict__ out_idx)
{
    // single thread-block, fixed order
    __shared__ double ss[256], sv[256];
    __shared__ long long si[256];
    double s = 0, bv = 0;
    long long bi = -1;
    for (int t = threadIdx.x; t < nblk; t += 256) {
        s += part_sum[t];
        if (want) argext_merge(bv, bi, part_val[t], part_idx[t], want);
    }
    ss[threadIdx.x] = s; sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            ss[threadIdx.x] += ss[threadIdx.x + o];
            if (want) {
                double v = sv[threadIdx.x];
                long long i = si[threadIdx.x];
                argext_merge(v, i, sv[threadIdx.x + o], si[threadIdx.x + o], want);
                sv[threadIdx.x] = v; si[threadIdx.x] = i;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (out_sum) *out_sum = ss[0];
        if (out_val) *out_val = sv[0];
        if (out_idx) *out_idx = (int64_t)si[0];
    }
}

} // namespace

extern "C" int apmf_full_score_grid_f64(apmf_ctx *ctx, int n, int m, int d, int ld, const double *d_mean,
                                        const double *d_cov, const uint8_t *d_mask, double *d_var,
                                        double *d_pmean, double *d_sum, int want, double *d_best_val,
                                        int64_t *d_best_idx)
{
    APMF_REQUIRE(ctx && d_mean && d_cov, "NULL operand");
    APMF_REQUIRE(n >= 1 && m >= 1 && d >= 1 && d <= APMF_MAX_D && ld >= m, "bad shape");
    APMF_REQUIRE(want >= -1 && want <= 1, "want must be -1, 0 or +1");
    APMF_REQUIRE(want == 0 || (d_best_val && d_best_idx), "want != 0 needs d_best_val/d_best_idx");
    if (!d_var && !d_pmean && !d_sum && want == 0) return APMF_OK;
    dim3 grid((m + 255) / 256, n);
    const size_t nblk = (size_t)grid.x * grid.y;
    void *w;
    APMF_TRY(apmf_ws(ctx, 1, nblk * 24 + 64, &w));
    double *part_sum = (double *)w;
    double *part_val = part_sum + nblk;
    long long *part_idx = (long long *)(part_val + nblk);
    const size_t smem = ((size_t)d * d + d) * sizeof(double);
    hipLaunchKernelGGL(k_full_grid, grid, dim3(256), smem, ctx->stream, n, m, d, ld, d_mean, d_cov, d_mask,
                       d_var, d_pmean, want, d_sum != nullptr, part_sum, part_val, part_idx);
    APMF_CHECK_LAUNCH();
    if (want || d_sum) {
        hipLaunchKernelGGL(k_full_finish, dim3(1), dim3(256), 0, ctx->stream, (int)nblk, want, part_sum,
                           part_val, part_idx, d_sum, d_best_val, d_best_idx);
        APMF_CHECK_LAUNCH();
    }
    return APMF_OK;
}
