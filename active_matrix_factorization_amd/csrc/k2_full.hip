// k2_full.hip - candidate-grid scoring under the full-covariance Gaussian of ActivePMF
// (active_pmf.py:301-322, 392-400, 502-524; normal_exps_cy.exp_dotprod_sq,
// normal_exps_cy.pyx:111-135), float64.
//
// With u = U_i ~ N(a, A), v = V_j ~ N(b, Bm), Cov(u_k, v_l) = C_kl (the d x d blocks of the
// (n+m)d x (n+m)d covariance at the reference's index maps u[k,i] = i d + k,
// v[k,j] = n d + j d + k), the reference's Isserlis sum minus mean^2 equals
//     Var = tr(A Bm) + tr(C C) + a'Bm a + b'A b + 2 b'C a        E = a'b + tr(C)
//         = <A + a a', Bm> + <A, b b'> + tr(C C) + 2 b'C a
// The reference reads A and Bm through their upper triangles (k <= l) - mirrored here - and
// C from both off-diagonal blocks, which agree for the symmetric matrices project_psd emits;
// this kernel reads the upper-right block.
//
// HBM-bound: per candidate the d x d cross block C is the streamed operand - 8 d^2 bytes read,
// 8 written; everything else is per-row / per-column and amortised:
//   k_full_prep   packs, once per sweep, the user side  rowpack (n, 2 d^2 + d) = [A + a a' | A | a]
//                 and the item side TRANSPOSED  colpack (d^2 + d, ld) = [Bm_j | b_j] (lane-contiguous
//                 over j) from the diagonal blocks of the covariance - which the first version
//                 gathered again, uncoalesced, for every (i, j);
//   k_full_sweep_t<D, ROWS>  (d <= 8, compile-time rank) one wave = ROWS user rows x 64 adjacent
//                 items.  The item pack (D^2 + D doubles) stays in the lane's registers for all its
//                 rows.  The D rows of the cross block that the 64 items share are contiguous (64 D
//                 doubles each): the wave fetches them with lane-consecutive 8-byte nontemporal loads
//                 - those of row r + 1 in flight while row r is contracted - and passes them through
//                 LDS (item stride padded to an odd number of doubles: conflict-free) so that every
//                 lane ends up with its own D x D block in registers.  The user-side pack is read
//                 with wave-uniform (scalar) loads.
//                 Measured (tools/bench_k2full.py): 1000 x 2000, d = 5 (400 MB streamed, past the
//                 Infinity Cache) 0.095 ms = 4.4 TB/s = 0.55 of the HBM peak (first version 0.47 ms);
//                 94 x 425, d = 5 (configs[3]; 8 MB, launch-bound) 11 us.
// d > 8: k_full_grid, one thread per candidate straight from global memory (the first version,
// kept as the general path).
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

// ---- packs ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_full_prep(int n, int m, int d, int ld, const double *__restrict__ mean,
                                                   const double *__restrict__ cov, double *__restrict__ rowpack,
                                                   double *__restrict__ colpack)
{
    const size_t K = (size_t)(n + m) * d;
    const int dd = d * d;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) {
        const int i = t;
        double *rp = rowpack + (size_t)i * (2 * dd + d);
        for (int k = 0; k < d; k++) {
            const double ak = mean[(size_t)i * d + k];
            rp[2 * dd + k] = ak;
            for (int l = 0; l < d; l++) {
                const int kk = min(k, l), ll = max(k, l);
                const double A = cov[((size_t)i * d + kk) * K + (size_t)i * d + ll];
                rp[dd + k * d + l] = A;
                rp[k * d + l] = A + ak * mean[(size_t)i * d + l];
            }
        }
    } else if (t < n + ld) {
        const int j = t - n;
        if (j < m) {
            const size_t vb = (size_t)n * d + (size_t)j * d;
            for (int k = 0; k < d; k++) {
                colpack[(size_t)(dd + k) * ld + j] = mean[vb + k];
                for (int l = 0; l < d; l++) {
                    const int kk = min(k, l), ll = max(k, l);
                    colpack[(size_t)(k * d + l) * ld + j] = cov[(vb + kk) * K + vb + ll];
                }
            }
        } else {
            for (int e = 0; e < dd + d; e++) colpack[(size_t)e * ld + j] = 0.;
        }
    }
}

// ---- the sweep, rank known at compile time: one wave = ROWS user rows x 64 items -------------------
// The item side (Bm_j, b_j: D^2 + D doubles) stays in the lane's registers for all ROWS rows, the
// lane's D x D cross block goes LDS -> registers once per row, and every contraction is unrolled.
template <int D, int ROWS>
__global__ __launch_bounds__(256) void k_full_sweep_t(int n, int m, int ld, const double *__restrict__ cov,
                                                      const double *__restrict__ rowpack,
                                                      const double *__restrict__ colpack,
                                                      const uint8_t *__restrict__ mask,
                                                      double *__restrict__ o_var, double *__restrict__ o_mean,
                                                      int want, int want_sum, double *__restrict__ part_sum,
                                                      double *__restrict__ part_val,
                                                      long long *__restrict__ part_idx)
{
    constexpr int DP = D | 1, DD = D * D;
    __shared__ double sC[4][D][64 * DP];
    const int lane = threadIdx.x & 63, w = wave_id_uniform();
    const int j0 = blockIdx.x * 64, j = j0 + lane;
    const size_t K = (size_t)(n + m) * D;
    const bool jin = j < m;
    const int span = min(64, m - j0) * D;
    double Bm[DD], b[D];
#pragma unroll
    for (int t = 0; t < DD; t++) Bm[t] = colpack[(size_t)t * ld + j];       // zero beyond m
#pragma unroll
    for (int k = 0; k < D; k++) b[k] = colpack[(size_t)(DD + k) * ld + j];
    double s_acc = 0., bv = 0.;
    long long bi = -1;
    // software pipeline: the D x D loads of row r + 1 are in flight while row r is contracted
    double st[D][D];
    auto fetch = [&](int i) {
        if (i < n) {
#pragma unroll
            for (int k = 0; k < D; k++) {
                const double *src = cov + ((size_t)i * D + k) * K + (size_t)n * D + (size_t)j0 * D;
#pragma unroll
                for (int it = 0; it < D; it++) {
                    const int e = it * 64 + lane;
                    st[k][it] = e < span ? __builtin_nontemporal_load(src + e) : 0.;
                }
            }
        }
    };
    fetch((blockIdx.y * 4 + w) * ROWS);
    for (int r = 0; r < ROWS; r++) {
        const int i = (blockIdx.y * 4 + w) * ROWS + r;
        const bool row_ok = i < n;                   // wave-uniform
        __syncthreads();
        if (row_ok) {
#pragma unroll
            for (int k = 0; k < D; k++)
#pragma unroll
                for (int it = 0; it < D; it++) {
                    const int e = it * 64 + lane;
                    const int jj = e / D, l = e - jj * D;
                    sC[w][k][jj * DP + l] = st[k][it];
                }
        }
        __syncthreads();
        if (r + 1 < ROWS) fetch(i + 1);
        if (!row_ok) continue;
        double C[D][D];
#pragma unroll
        for (int k = 0; k < D; k++)
#pragma unroll
            for (int l = 0; l < D; l++) C[k][l] = sC[w][k][lane * DP + l];
        const double *rp = rowpack + (size_t)i * (2 * DD + D);      // wave-uniform: scalar loads
        double pb = 0., aq = 0., trCC = 0., bCa = 0., ab = 0., trC = 0.;
#pragma unroll
        for (int k = 0; k < D; k++) {
            double Ab_k = 0., Ca_k = 0.;
#pragma unroll
            for (int l = 0; l < D; l++) {
                pb += rp[k * D + l] * Bm[k * D + l];
                Ab_k += rp[DD + k * D + l] * b[l];
                Ca_k += C[k][l] * rp[2 * DD + l];
                trCC += C[k][l] * C[l][k];
            }
            aq += b[k] * Ab_k;
            bCa += b[k] * Ca_k;
            ab += rp[2 * DD + k] * b[k];
            trC += C[k][k];
        }
        const double var = pb + aq + trCC + 2. * bCa;
        if (j < ld) {
            if (o_var) o_var[(size_t)i * ld + j] = jin ? var : 0.;
            if (o_mean) o_mean[(size_t)i * ld + j] = jin ? ab + trC : 0.;
        }
        if (want || want_sum) {
            const bool cand = jin && (!mask || mask[(size_t)i * ld + j]);
            if (cand) {
                s_acc += var;
                const long long idx = (long long)i * m + j;
                // rows ascend within the lane: a strict compare keeps the lowest flat index on ties
                if (bi < 0 || (want > 0 ? var > bv : var < bv)) { bv = var; bi = idx; }
            }
        }
    }
    if (want || want_sum) {
        double s = wave_sum(s_acc);
        if (want) wave_argext(bv, bi, want);
        __shared__ double sh_s[4], sh_v[4];
        __shared__ long long sh_i[4];
        __syncthreads();
        if (lane == 0) { sh_s[w] = s; sh_v[w] = bv; sh_i[w] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int x = 1; x < 4; x++) {
                s += sh_s[x];
                if (want) argext_merge(bv, bi, sh_v[x], sh_i[x], want);
            }
            const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            part_sum[slot] = s; part_val[slot] = bv; part_idx[slot] = bi;
        }
    }
}

template <int D>
void launch_full_t(hipStream_t st, dim3 &grid, int n, int m, int ld, int rows, const double *cov, const double *rowpack,
                   const double *colpack, const uint8_t *mask, double *o_var, double *o_mean, int want, int want_sum,
                   double *ps, double *pv, long long *pi)
{
    if (rows == 8)
        hipLaunchKernelGGL((k_full_sweep_t<D, 8>), grid, dim3(256), 0, st, n, m, ld, cov, rowpack, colpack, mask, o_var,
                           o_mean, want, want_sum, ps, pv, pi);
    else
        hipLaunchKernelGGL((k_full_sweep_t<D, 2>), grid, dim3(256), 0, st, n, m, ld, cov, rowpack, colpack, mask, o_var,
                           o_mean, want, want_sum, ps, pv, pi);
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
    // compile-time rank (d <= 8): ROWS user rows per wave, 8 when that still fills the machine
    int rows_t = 0;
    if (d >= 2 && d <= 8) rows_t = (size_t)((ld + 63) / 64) * ((n + 31) / 32) >= 1024 ? 8 : 2;
    if (const char *e = getenv("APMF_K2FULL_KERNEL"))
        if (!strcmp(e, "general")) rows_t = 0;
    dim3 grid = rows_t ? dim3((ld + 63) / 64, (n + 4 * rows_t - 1) / (4 * rows_t)) : dim3((m + 255) / 256, n);
    const size_t nblk = (size_t)grid.x * grid.y;
    const size_t dd = (size_t)d * d;
    const size_t pack_doubles = rows_t ? (size_t)n * (2 * dd + d) + (dd + d) * (size_t)ld : 0;
    void *w;
    APMF_TRY(apmf_ws(ctx, 1, nblk * 24 + pack_doubles * sizeof(double) + 64, &w));
    double *part_sum = (double *)w;
    double *part_val = part_sum + nblk;
    long long *part_idx = (long long *)(part_val + nblk);
    if (rows_t) {
        double *rowpack = (double *)(part_idx + nblk);
        double *colpack = rowpack + (size_t)n * (2 * dd + d);
        hipLaunchKernelGGL(k_full_prep, dim3((n + ld + 255) / 256), dim3(256), 0, ctx->stream, n, m, d, ld, d_mean,
                           d_cov, rowpack, colpack);
        ProfScope prof(ctx);
        const int ws = d_sum != nullptr;
        switch (d) {
#define FULL_T(DV) case DV: launch_full_t<DV>(ctx->stream, grid, n, m, ld, rows_t, d_cov, rowpack, colpack, d_mask, d_var, d_pmean, want, ws, part_sum, part_val, part_idx); break;
        FULL_T(2) FULL_T(3) FULL_T(4) FULL_T(5) FULL_T(6) FULL_T(7) FULL_T(8)
#undef FULL_T
        }
    } else {
        const size_t smem = ((size_t)d * d + d) * sizeof(double);
        ProfScope prof(ctx);
        hipLaunchKernelGGL(k_full_grid, grid, dim3(256), smem, ctx->stream, n, m, d, ld, d_mean, d_cov, d_mask,
                           d_var, d_pmean, want, d_sum != nullptr, part_sum, part_val, part_idx);
    }
    APMF_CHECK_LAUNCH();
    if (want || d_sum) {
        hipLaunchKernelGGL(k_full_finish, dim3(1), dim3(256), 0, ctx->stream, (int)nblk, want, part_sum,
                           part_val, part_idx, d_sum, d_best_val, d_best_idx);
        APMF_CHECK_LAUNCH();
    }
    return APMF_OK;
}
