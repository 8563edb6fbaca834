// k8_fullbatch.hip - H variational fits of the FULL-covariance approximation N(mean, cov) of ActivePMF as one
// chain of launches (libapmf_dense.so; C-ABI apmf_full_batch_*).
//
// What is replaced: the reference's one-step lookahead of the full-covariance class (active_pmf.py:635-704)
// evaluates every hypothesis "entry (i, j) turns out to be v" as deepcopy -> add_rating -> fit_normal
// (active_pmf.py:250-288: projected gradient descent on the KL with a x1.25 / x0.5 learning-rate search),
// one after the other or on a forked pool.  Here the H hypotheses of a pool are H states (mean (K), cov (K x K),
// K = (n + m) d) resident in device memory and every search iteration advances ALL active ones:
//   gradient   cov^-1 from the Cholesky factor the accepted trial left behind (block triangular inverse +
//              W'W on the matrix cores, the kernels below), then normal_exps_cy.pyx:140-303's rating terms
//              (one wave per user / item row; the hypothesis' extra rating is the LAST entry of user row i /
//              item row j - where add_rating() and a rebuilt container put it);
//   trial      cand = cur - lr g (two roundings, like NumPy), symmetrised (project_psd's first step), and the
//              two matrices to factor: cand itself and cand - (min_eig + delta) I;
//   project    "no eigenvalue below min_eig: nothing to clip" is certified by the factorisation of the shifted
//              matrix (chol.h: all 2 H factorisations of an iteration are jobs of the same launches); a trial
//              that fails it is projected by the single-matrix routine (eigendecomposition), as before;
//   value      the KL (active_pmf.py:202-246): E[(u.v)^2] - 2 r E[u.v] + r^2 over the ratings, the prior terms,
//              and log det from the factor's diagonal;
//   decide     on the host, from one small read-back per iteration.
// A hypothesis' arithmetic never depends on the others of the batch: a batched fit is bit-identical to the same
// fit run alone - and ActivePMF.fit_normal itself is the H = 1 case of this code, so the batched lookahead
// equals the per-hypothesis one bit for bit.
#include <math.h>

#include <algorithm>
#include <vector>

#include "chol.h"
#include "common.h"

extern "C" int apmf_dense_project_psd(apmf_ctx *ctx, double *d_mat, int N, double min_eig, int *h_projected);
extern "C" int apmf_dense_inv_logdet(apmf_ctx *ctx, const double *d_mat, int N, double *d_inv, double *h_logdet);
extern "C" int apmf_full_score_grid_f64(apmf_ctx *ctx, int n, int m, int d, int ld, const double *d_mean,
                                        const double *d_cov, const uint8_t *d_mask, double *d_var,
                                        double *d_pmean, double *d_sum, int want, double *d_best_val,
                                        int64_t *d_best_idx);

namespace {

constexpr int FB_D = 32;      // largest latent_d (the gradient kernel's LDS blocks)

// what the kernels see: per-hypothesis arrays are H consecutive blocks; which of the two state slots holds a
// hypothesis' current point is slot[h] (the other one receives the trial point)
struct FRefs {
    double *mean[2], *cov[2];    // two state slots: H x K, H x K x K
    double *gmean, *gcov;        // gradient at the current point
    double *F, *W;               // factor of the point last factored / certificate scratch
    double *linvF, *linvW;       // inverses of the diagonal blocks: [H][nbk][64 x 64] kept, [H][64 x 64]
    int *info;                   // [2 H]: 2 h certificate, 2 h + 1 factor
    const int *slot;             // [H] slot of the current point
    const double *lr;            // [H]
    const int32_t *ei, *ej;      // extra rating (ei < 0: none)
    const double *ev;
    double *out;                 // [H][4]: data term, user prior term, item prior term, log det
    int H, n, m, d, K, nbk;
};

__device__ __forceinline__ double *cur_mean(const FRefs &R, int h) { return R.mean[R.slot[h]] + (size_t)h * R.K; }
__device__ __forceinline__ double *cur_cov(const FRefs &R, int h) { return R.cov[R.slot[h]] + (size_t)h * R.K * R.K; }
__device__ __forceinline__ double *cand_mean(const FRefs &R, int h) { return R.mean[R.slot[h] ^ 1] + (size_t)h * R.K; }
__device__ __forceinline__ double *cand_cov(const FRefs &R, int h) { return R.cov[R.slot[h] ^ 1] + (size_t)h * R.K * R.K; }

// ---- trial point -------------------------------------------------------------------------------------
// cand = sym(cur - lr g): element pairs (i > j) are handled together, x - fl(lr g) with NumPy's two roundings,
// (a + b) / 2 like project_psd's first line; F <- cand, W <- cand - (min_eig + delta) I (lower triangles).
__global__ __launch_bounds__(256) void kf_trial(const FRefs R, const int *__restrict__ sel, double min_eig)
{
    const int h = sel[blockIdx.y];
    const size_t K = (size_t)R.K;
    const double lr = R.lr[h];
    const double *__restrict__ c = cur_cov(R, h), *__restrict__ g = R.gcov + (size_t)h * K * K;
    double *__restrict__ o = cand_cov(R, h), *__restrict__ F = R.F + (size_t)h * K * K, *__restrict__ W = R.W + (size_t)h * K * K;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < K * K) {
        const size_t i = t / K, j = t % K;
        if (j <= i) {
            const double a = __dsub_rn(c[i * K + j], __dmul_rn(lr, g[i * K + j]));
            double s = a;
            if (j < i) {
                const double b = __dsub_rn(c[j * K + i], __dmul_rn(lr, g[j * K + i]));
                s = (b + a) / 2;          // (A[j][i] + A[i][j]) / 2: the operand order of k_symmetrize (i < j there)
                o[j * K + i] = s;
            }
            o[i * K + j] = s;
            F[i * K + j] = s;
            W[i * K + j] = i == j ? s - (min_eig + 1e-10 * fmax(fabs(s), fabs(min_eig))) : s;
        }
    }
    if (t < K) {
        const double *__restrict__ m = cur_mean(R, h), *__restrict__ gm = R.gmean + (size_t)h * K;
        cand_mean(R, h)[t] = __dsub_rn(m[t], __dmul_rn(lr, gm[t]));
    }
}

// F <- the current covariance (lower triangle), for the factorisation of a starting point
__global__ __launch_bounds__(256) void kf_copy_cur(const FRefs R, const int *__restrict__ sel)
{
    const int h = sel[blockIdx.y];
    const size_t K = (size_t)R.K, t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < K * K) R.F[(size_t)h * K * K + t] = cur_cov(R, h)[t];
}

// ---- E[u.v], Var[u.v] of one entry under N(mean, cov) (normal_exps_cy.pyx:111-135 minus mean^2, the
// closed form of k2_full.hip): upper triangles of the diagonal blocks mirrored, the cross block as stored ----
__device__ __forceinline__ void entry_moments(const double *__restrict__ mean, const double *__restrict__ cov, size_t K,
                                              int n, int d, int i, int j, double &mn, double &var)
{
    const size_t ub = (size_t)i * d, vb = (size_t)n * d + (size_t)j * d;
    double trAB = 0., trCC = 0., aBa = 0., bAb = 0., bCa = 0., ab = 0., trC = 0.;
    for (int k = 0; k < d; k++) {
        const double *Crow = cov + (ub + k) * K + vb;
        double Ba_k = 0., Ab_k = 0., Ca_k = 0.;
        for (int l = 0; l < d; l++) {
            const int kk = k < l ? k : l, ll = k < l ? l : k;
            const double Akl = cov[(ub + kk) * K + ub + ll];
            const double Bkl = cov[(vb + kk) * K + vb + ll];
            const double Ckl = Crow[l];
            const double Clk = cov[(ub + l) * K + vb + k];
            trAB += Akl * Bkl;
            trCC += Ckl * Clk;
            Ba_k += Bkl * mean[ub + l];
            Ab_k += Akl * mean[vb + l];
            Ca_k += Ckl * mean[ub + l];
        }
        aBa += mean[ub + k] * Ba_k;
        bAb += mean[vb + k] * Ab_k;
        bCa += mean[vb + k] * Ca_k;
        ab += mean[ub + k] * mean[vb + k];
        trC += Crow[k];
    }
    var = trAB + trCC + aBa + bAb + 2. * bCa;
    mn = ab + trC;
}

// ---- KL value (active_pmf.py:202-246), one block per hypothesis, fixed summation order ---------------------
// which = 0: the current point, 1: the trial point.  out[h] = {sum over the ratings of Var + E^2 - 2 r E + r^2,
// sum_u (mean^2 + cov_kk), the same over v, log det from the factor's diagonal}
__global__ __launch_bounds__(256) void kf_value(const FRefs R, const int *__restrict__ sel, int which,
                                                const int64_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                const double *__restrict__ val)
{
    __shared__ double sh[4][256];
    const int h = sel[blockIdx.x];
    const size_t K = (size_t)R.K;
    const double *__restrict__ mean = which ? cand_mean(R, h) : cur_mean(R, h);
    const double *__restrict__ cov = which ? cand_cov(R, h) : cur_cov(R, h);
    const double *__restrict__ F = R.F + (size_t)h * K * K;
    const int xi = R.ei[h], xj = R.ej[h];
    double a = 0., bu = 0., bv = 0., ld = 0.;
    // thread t takes user rows t, t + 256, ...; a row's ratings in storage order, its extra rating last
    for (int i = threadIdx.x; i < R.n; i += 256) {
        const int64_t beg = ptr[i], end = ptr[i + 1] + (i == xi ? 1 : 0);
        for (int64_t t = beg; t < end; t++) {
            const bool extra = t == ptr[i + 1];
            const int j = extra ? xj : idx[t];
            const double r = extra ? R.ev[h] : val[t];
            double mn, var;
            entry_moments(mean, cov, K, R.n, R.d, i, j, mn, var);
            a += var + mn * mn - 2. * r * mn + r * r;
        }
    }
    const size_t nd = (size_t)R.n * R.d;
    for (size_t k = threadIdx.x; k < K; k += 256) {
        const double v = mean[k] * mean[k] + cov[k * K + k];
        if (k < nd) bu += v;
        else bv += v;
        ld += log(F[k * K + k]);
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = bu; sh[2][threadIdx.x] = bv; sh[3][threadIdx.x] = ld;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int z = 0; z < 4; z++) sh[z][threadIdx.x] += sh[z][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) R.out[(size_t)h * 4 + threadIdx.x] = threadIdx.x == 3 ? 2. * sh[3][0] : sh[threadIdx.x][0];
}

// ---- cov^-1 from the factor: W = L^-1 by block rows, then L^-T L^-1 = W'W --------------------------------
// (the tile kernels of k6_batch.hip's inverse, on plain per-hypothesis arrays)
struct FTile {
    const double *p;
    int ld, r0, c0, re, ce;
};
__device__ __forceinline__ FTile ft_mat(const double *A, int N, int r0, int c0) { return {A, N, r0, c0, N, N}; }
__device__ __forceinline__ FTile ft_dense(const double *B) { return {B, CH_NB, 0, 0, CH_NB, CH_NB}; }
__device__ __forceinline__ void ft_request(TileRegs<64, 64> &t, const FTile &s) { t.load(s.p, s.ld, s.r0, s.c0, s.re, s.ce); }
__device__ __forceinline__ void ft_place(const TileRegs<64, 64> &t, const FTile &s, double (*T)[TP])
{
#pragma unroll
    for (int q = 0; q < t.PER; q++) {
        const int r = t.row(q), c = t.col(q);
        T[r][c] = (s.r0 + r < s.re && s.c0 + c < s.ce) ? t.v[q] : 0.;
    }
}

// launch k (0 .. nbk - 2): tiles (i > k, j <= k):  W_ij <- W_ij - L_ik W_kj  (W_kk = Linv_kk, W_ij = 0 at j = k);
// the tiles of row k + 1 are multiplied by Linv_{k+1,k+1} on the way out.  Diagonal tiles of W are the kept Linv_kk.
__global__ __launch_bounds__(256) void kf_trinv_step(const FRefs R, const int *__restrict__ sel, int k)
{
    __shared__ double A[64][TP], B[64][TP];
    int bx, by;
    xcd_tile_job(bx, by);
    const int h = sel[by];
    const int N = R.K, nb = R.nbk;
    const int i = k + 1 + bx / (k + 1), j = bx % (k + 1);
    const double *__restrict__ L = R.F + (size_t)h * N * N;
    double *__restrict__ W = R.W + (size_t)h * N * N;
    const double *__restrict__ linv = R.linvF + (size_t)h * nb * CH_NB * CH_NB;
    const FTile sa = ft_mat(L, N, i * 64, k * 64);
    const FTile sb = j == k ? ft_dense(linv + (size_t)k * CH_NB * CH_NB) : ft_mat(W, N, k * 64, j * 64);
    const FTile sc = ft_dense(linv + (size_t)(k + 1) * CH_NB * CH_NB);
    TileRegs<64, 64> ta, tb, tc;
    ft_request(ta, sa);
    ft_request(tb, sb);
    f64x4 old[2][2];
    TILE_FOR_EACH(u, v, r) {
        const int rr = min(i * 64 + tile_row(u, r), N - 1), cc = min(j * 64 + tile_col(v), N - 1);
        old[u][v][r] = W[(size_t)rr * N + cc];
    }
    if (i == k + 1) ft_request(tc, sc);
    ft_place(ta, sa, A);
    ft_place(tb, sb, B);
    __syncthreads();
    f64x4 acc[2][2] = {};
    const int rmax = N - i * 64;
    tile_mfma<MM_NN>(acc, A, B, 64, rmax);
    __syncthreads();
    TILE_FOR_EACH(u, v, r) {
        const int rr = i * 64 + tile_row(u, r), cc = j * 64 + tile_col(v);
        acc[u][v][r] = ((j < k && rr < N && cc < N) ? old[u][v][r] : 0.) - acc[u][v][r];
    }
    if (i == k + 1) {
        TILE_FOR_EACH(u, v, r) B[tile_row(u, r)][tile_col(v)] = acc[u][v][r];
        ft_place(tc, sc, A);
        __syncthreads();
        TILE_FOR_EACH(u, v, r) acc[u][v][r] = 0.;
        tile_mfma<MM_NN>(acc, A, B, min(64, (rmax + 3) & ~3), rmax);
    }
    TILE_FOR_EACH(u, v, r) {
        const int rr = i * 64 + tile_row(u, r), cc = j * 64 + tile_col(v);
        if (rr < N && cc < N) W[(size_t)rr * N + cc] = acc[u][v][r];
    }
}

// gcov <- cov^-1 = W'W: tile (ti >= tj) = sum_{k >= ti} W_k,ti' W_k,tj, both triangles written
__global__ __launch_bounds__(256) void kf_lauum(const FRefs R, const int *__restrict__ sel)
{
    __shared__ double A[64][TP], B[64][TP];
    int bx, by;
    xcd_tile_job(bx, by);
    const int h = sel[by];
    const int N = R.K, nb = R.nbk;
    int ti, tj;
    tri_tile(bx, ti, tj);
    const double *__restrict__ W = R.W + (size_t)h * N * N;
    double *__restrict__ G = R.gcov + (size_t)h * N * N;
    const double *__restrict__ linv = R.linvF + (size_t)h * nb * CH_NB * CH_NB;
    f64x4 acc[2][2] = {};
    auto src_a = [&](int k) { return k == ti ? ft_dense(linv + (size_t)k * CH_NB * CH_NB) : ft_mat(W, N, k * 64, ti * 64); };
    auto src_b = [&](int k) { return k == tj ? ft_dense(linv + (size_t)k * CH_NB * CH_NB) : ft_mat(W, N, k * 64, tj * 64); };
    TileRegs<64, 64> ta, tb;
    ft_request(ta, src_a(ti));
    ft_request(tb, src_b(ti));
    for (int k = ti; k < nb; k++) {
        __syncthreads();
        ft_place(ta, src_a(k), A);
        ft_place(tb, src_b(k), B);
        if (k + 1 < nb) {
            ft_request(ta, src_a(k + 1));
            ft_request(tb, src_b(k + 1));
        }
        __syncthreads();
        tile_mfma<MM_TN>(acc, A, B, min(64, (N - k * 64 + 3) & ~3), N - ti * 64, N - tj * 64);
    }
    TILE_FOR_EACH(u, v, r) {
        const int rr = ti * 64 + tile_row(u, r), cc = tj * 64 + tile_col(v);
        if (rr < N && cc < N && (ti != tj || cc <= rr)) {
            G[(size_t)rr * N + cc] = acc[u][v][r];
            if (rr != cc) G[(size_t)cc * N + rr] = acc[u][v][r];
        }
    }
}

// ---- gradient (normal_exps_cy.pyx:140-303) ----------------------------------------------------------------
// gcov holds cov^-1 (symmetric): gcov <- -inv off the diagonal, -inv / 2 + 1 / (2 prior) on it; gmean <- mean / prior
__global__ __launch_bounds__(256) void kf_grad_init(const FRefs R, const int *__restrict__ sel, double sig_u, double sig_v)
{
    const int h = sel[blockIdx.y];
    const size_t K = (size_t)R.K, nd = (size_t)R.n * R.d;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= K * K) return;
    const size_t i = t / K, j = t % K;
    double *__restrict__ gcov = R.gcov + (size_t)h * K * K;
    const double inv = gcov[t];
    if (i == j) {
        const double prior = i < nd ? sig_u : sig_v;
        gcov[t] = -inv / 2. + 1. / (2. * prior);
        R.gmean[(size_t)h * K + i] = cur_mean(R, h)[i] / prior;
    } else if (i < j) {
        const double invt = gcov[j * K + i];
        gcov[t] = -(inv + invt) / 2.;
        gcov[j * K + i] = -(invt + inv) / 2.;
    }
}

// one wave per (row, hypothesis): the arithmetic of k5_klgrad.hip's k_full_grad_rows, the hypothesis' extra
// rating taken as the last entry of user row i / item row j
template <int SIDE>
__global__ __launch_bounds__(64) void kf_grad_rows(const FRefs R, const int *__restrict__ sel,
                                                   const int64_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                   const double *__restrict__ val, double inv_sig)
{
    __shared__ double Cuu[FB_D][FB_D + 1], Cuv[FB_D][FB_D + 1], Cvv[FB_D][FB_D + 1], OWN[FB_D][FB_D + 1];
    __shared__ double sa[FB_D], sb[FB_D], se[FB_D], s1[FB_D], s2[FB_D], sA[FB_D], sB[FB_D], sC[FB_D], sD[FB_D];
    const int h = sel[blockIdx.y];
    const int n = R.n, d = R.d;
    const int x = threadIdx.x;
    const int row = blockIdx.x;
    const size_t K = (size_t)R.K;
    const double *__restrict__ mean = cur_mean(R, h), *__restrict__ cov = cur_cov(R, h);
    double *__restrict__ gmean = R.gmean + (size_t)h * K, *__restrict__ gcov = R.gcov + (size_t)h * K * K;
    const int xi = R.ei[h], xj = R.ej[h];
    const bool has_extra = SIDE == 0 ? row == xi : row == xj;
    const int64_t beg = ptr[row], mem_end = ptr[row + 1], end = mem_end + (has_extra ? 1 : 0);
    for (int e = x; e < d * d; e += 64) OWN[e / d][e % d] = 0.;
    double gm = 0.;
    for (int64_t t = beg; t < end; t++) {
        const bool extra = t == mem_end;
        const int other = extra ? (SIDE == 0 ? xj : xi) : idx[t];
        const int i = SIDE == 0 ? row : other, j = SIDE == 0 ? other : row;
        const double r = extra ? R.ev[h] : val[t];
        const size_t ub = (size_t)i * d, vb = (size_t)n * d + (size_t)j * d;
        __syncthreads();
        if (x < d) { sa[x] = mean[ub + x]; sb[x] = mean[vb + x]; }
        for (int e = x; e < d * d; e += 64) {
            const int p = e / d, q = e % d;
            Cuu[p][q] = cov[(ub + p) * K + ub + q];
            Cuv[p][q] = cov[(ub + p) * K + vb + q];
            Cvv[p][q] = cov[(vb + p) * K + vb + q];
        }
        __syncthreads();
        double a = 0, b = 0, pxy = 0, cuv = 0, cuu = 0, cvv = 0, e_x = 0;
        if (x < d) {
            a = sa[x]; b = sb[x]; pxy = a * b; cuv = Cuv[x][x]; cuu = Cuu[x][x]; cvv = Cvv[x][x];
            e_x = pxy + cuv;
            se[x] = e_x;
        }
        __syncthreads();
        double gmu = 0.;
        if (x < d) {
            double Kx = 0., S1 = 0., S2 = 0., Svv = 0., Suu = 0., Svu = 0., Suv = 0.;
            for (int y = x + 1; y < d; y++) {
                const double ay = sa[y], by = sb[y], ey = se[y];
                if (SIDE == 0) Kx += b * ey + ay * Cvv[x][y] + by * Cuv[y][x];
                else Kx += a * ey + ay * Cuv[x][y] + by * Cuu[x][y];
                S1 += pxy * by + a * Cvv[x][y] + b * Cuv[x][y] + by * cuv;
                S2 += pxy * ay + a * Cuv[y][x] + b * Cuu[x][y] + ay * cuv;
                Svv += b * by + Cvv[x][y];
                Suu += a * ay + Cuu[x][y];
                Svu += b * ay + Cuv[y][x];
                Suv += a * by + Cuv[x][y];
            }
            s1[x] = S1; s2[x] = S2; sA[x] = Svv; sB[x] = Suu; sC[x] = Svu; sD[x] = Suv;
            gmu = Kx;
        }
        __syncthreads();
        if (x < d) {
            double L = 0.;
            for (int k = 0; k < x; k++) L += SIDE == 0 ? s1[k] : s2[k];
            if (SIDE == 0) gmu += L + 2. * b * cuv + a * (b * b + cvv) - b * r;
            else gmu += L + 2. * a * cuv + b * (a * a + cuu) - a * r;
            gm += gmu;
        }
        for (int e = x; e < d * d; e += 64) {
            const int p = e / d, q = e % d;
            const double *S = SIDE == 0 ? sA : sB;
            double v;
            if (p < q) v = S[p];
            else if (p > q) v = S[q];
            else v = SIDE == 0 ? (sb[p] * sb[p] + Cvv[p][p]) / 2. : (sa[p] * sa[p] + Cuu[p][p]) / 2.;
            OWN[p][q] += v;
        }
        if (SIDE == 0) {
            double esum = 0.;
            for (int y = 0; y < d; y++) esum += se[y];
            for (int e = x; e < d * d; e += 64) {
                const int p = e / d, q = e % d;
                double uv, vu;
                if (q > p) { uv = sC[p]; vu = sD[p]; }
                else if (q < p) { uv = sD[q]; vu = sC[q]; }
                else uv = vu = (esum - se[p]) + 2. * se[p] - r;
                gcov[(ub + p) * K + vb + q] += inv_sig * uv;
                gcov[(vb + p) * K + ub + q] += inv_sig * vu;
            }
        }
    }
    __syncthreads();
    const size_t ob = SIDE == 0 ? (size_t)row * d : (size_t)n * d + (size_t)row * d;
    if (x < d) gmean[ob + x] += inv_sig * gm;
    for (int e = x; e < d * d; e += 64) {
        const int p = e / d, q = e % d;
        gcov[(ob + p) * K + ob + q] += inv_sig * OWN[p][q];
    }
}

} // namespace

// ---- the batch ----------------------------------------------------------------------------------------------
struct apmf_full_batch {
    apmf_ctx *ctx;
    const apmf_ratings *r;
    FRefs R;
    std::vector<void *> blocks;
    int *d_slot, *d_sel;
    double *d_lr;
    int32_t *d_ei, *d_ej;
    double *d_ev;
    ChJob *d_jobs;
    // host mirrors
    std::vector<int> slot, fvalid, hi, hj;
    std::vector<double> lr, old_kl, cur_ld, hv;
    long st_iters, st_trials, st_projected;
};

namespace {

int fb_alloc(apmf_full_batch *b, size_t bytes, void **out)
{
    APMF_TRY(apmf_malloc(b->ctx, bytes ? bytes : 8, out));
    b->blocks.push_back(*out);
    return APMF_OK;
}

int fb_upload_sel(apmf_full_batch *b, const std::vector<int> &v)
{
    return apmf_h2d(b->ctx, b->d_sel, v.data(), v.size() * sizeof(int));
}

// factor the matrices in F (and, with cert, the shifted ones in W) of the listed hypotheses
int fb_factor(apmf_full_batch *b, const std::vector<int> &list, bool cert)
{
    const FRefs &R = b->R;
    const size_t KK = (size_t)R.K * R.K;
    std::vector<ChJob> jobs;
    for (int h : list) {
        if (cert) jobs.push_back(ChJob{R.W + h * KK, R.info + 2 * h, R.linvW + (size_t)h * CH_NB * CH_NB, 0});
        jobs.push_back(ChJob{R.F + h * KK, R.info + 2 * h + 1, R.linvF + (size_t)h * R.nbk * CH_NB * CH_NB, 1});
    }
    APMF_TRY(apmf_h2d(b->ctx, b->d_jobs, jobs.data(), jobs.size() * sizeof(ChJob)));
    ChJobs J;
    memset(&J, 0, sizeof(J));
    J.list = b->d_jobs;
    return chol_lower_jobs(b->ctx, J, (int)jobs.size(), R.K);
}

// the KL of the current (which = 0) / trial (1) point of the listed hypotheses from kf_value's four sums; the log
// determinant comes from F's diagonal where the factorisation succeeded (ok[h]), from the general routine otherwise
int fb_values(apmf_full_batch *b, const std::vector<int> &list, int which, const std::vector<char> &ok, double sigma_sq,
              double sigma_u_sq, double sigma_v_sq, std::vector<double> &kl, std::vector<double> &ld)
{
    const FRefs &R = b->R;
    APMF_TRY(fb_upload_sel(b, list));
    hipLaunchKernelGGL(kf_value, dim3((unsigned)list.size()), dim3(256), 0, b->ctx->stream, R, b->d_sel, which,
                       b->r->ptr[0], b->r->idx[0], b->r->val[0]);
    APMF_CHECK_LAUNCH();
    std::vector<double> out((size_t)R.H * 4);
    APMF_TRY(apmf_d2h(b->ctx, out.data(), R.out, out.size() * sizeof(double)));
    const size_t KK = (size_t)R.K * R.K;
    for (size_t q = 0; q < list.size(); q++) {
        const int h = list[q];
        double logdet = out[(size_t)h * 4 + 3];
        if (!ok[q]) {
            const double *mat = R.cov[b->slot[h] ^ (which ? 1 : 0)] + h * KK;
            APMF_TRY(apmf_dense_inv_logdet(b->ctx, mat, R.K, nullptr, &logdet));
        }
        ld[h] = logdet;
        kl[h] = out[(size_t)h * 4] / (2. * sigma_sq) + out[(size_t)h * 4 + 1] / (2. * sigma_u_sq) +
                out[(size_t)h * 4 + 2] / (2. * sigma_v_sq) - logdet / 2.;
    }
    return APMF_OK;
}

int fb_push_state(apmf_full_batch *b)
{
    APMF_TRY(apmf_h2d(b->ctx, b->d_slot, b->slot.data(), b->slot.size() * sizeof(int)));
    return apmf_h2d(b->ctx, b->d_lr, b->lr.data(), b->lr.size() * sizeof(double));
}

} // namespace

extern "C" {

int apmf_full_batch_destroy(apmf_ctx *ctx, apmf_full_batch *b)
{
    if (!b) return APMF_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    for (void *p : b->blocks) (void)apmf_free(b->ctx, p);
    delete b;
    return APMF_OK;
}

int apmf_full_batch_create(apmf_ctx *ctx, const apmf_ratings *r, int d, int H, apmf_full_batch **out)
{
    APMF_REQUIRE(ctx && r && out, "NULL operand");
    *out = nullptr;
    APMF_REQUIRE(d >= 1 && d <= FB_D, "latent_d=%d unsupported by the full-covariance path (1..%d)", d, FB_D);
    APMF_REQUIRE(H >= 1, "empty batch");
    const size_t K = ((size_t)r->n + r->m) * d;
    APMF_REQUIRE(K <= 46000, "(n+m) d = %zu is too large for a dense covariance", K);
    apmf_full_batch *b = new (std::nothrow) apmf_full_batch();
    if (!b) { apmf_set_error("out of host memory"); return APMF_ENOMEM; }
    b->ctx = ctx;
    b->r = r;
    FRefs &R = b->R;
    memset(&R, 0, sizeof(R));
    R.H = H; R.n = r->n; R.m = r->m; R.d = d; R.K = (int)K; R.nbk = (int)((K + CH_NB - 1) / CH_NB);
    const size_t KK = K * K;
    int rc = APMF_OK;
    auto get = [&](size_t bytes, void **p) { if (rc == APMF_OK) rc = fb_alloc(b, bytes, p); };
    for (int s = 0; s < 2; s++) {
        get((size_t)H * K * 8, (void **)&R.mean[s]);
        get((size_t)H * KK * 8, (void **)&R.cov[s]);
    }
    get((size_t)H * K * 8, (void **)&R.gmean);
    get((size_t)H * KK * 8, (void **)&R.gcov);
    get((size_t)H * KK * 8, (void **)&R.F);
    get((size_t)H * KK * 8, (void **)&R.W);
    get((size_t)H * R.nbk * CH_NB * CH_NB * 8, (void **)&R.linvF);
    get((size_t)H * CH_NB * CH_NB * 8, (void **)&R.linvW);
    get((size_t)2 * H * sizeof(int), (void **)&R.info);
    get((size_t)H * 4 * 8, (void **)&R.out);
    get((size_t)H * sizeof(int), (void **)&b->d_slot);
    get((size_t)H * sizeof(int), (void **)&b->d_sel);
    get((size_t)H * 8, (void **)&b->d_lr);
    get((size_t)H * 4, (void **)&b->d_ei);
    get((size_t)H * 4, (void **)&b->d_ej);
    get((size_t)H * 8, (void **)&b->d_ev);
    get((size_t)2 * H * sizeof(ChJob), (void **)&b->d_jobs);
    if (rc != APMF_OK) { (void)apmf_full_batch_destroy(ctx, b); return rc; }
    R.slot = b->d_slot; R.lr = b->d_lr; R.ei = b->d_ei; R.ej = b->d_ej; R.ev = b->d_ev;
    b->slot.assign(H, 0);
    b->fvalid.assign(H, 0);
    b->hi.assign(H, -1);
    b->hj.assign(H, -1);
    b->hv.assign(H, 0.);
    b->lr.assign(H, 0.);
    b->old_kl.assign(H, NAN);
    b->cur_ld.assign(H, NAN);
    rc = apmf_h2d(ctx, b->d_ei, b->hi.data(), (size_t)H * 4);
    if (rc == APMF_OK) rc = apmf_h2d(ctx, b->d_ej, b->hj.data(), (size_t)H * 4);
    if (rc == APMF_OK) rc = apmf_h2d(ctx, b->d_ev, b->hv.data(), (size_t)H * 8);
    if (rc == APMF_OK) rc = fb_push_state(b);
    if (rc != APMF_OK) { (void)apmf_full_batch_destroy(ctx, b); return rc; }
    *out = b;
    return APMF_OK;
}

// every hypothesis starts from (d_mean (K), d_cov (K x K)); per_hypothesis != 0: from its own (H x K, H x K x K)
int apmf_full_batch_set_state(apmf_ctx *ctx, apmf_full_batch *b, const double *d_mean, const double *d_cov,
                              int per_hypothesis)
{
    APMF_REQUIRE(ctx && b && d_mean && d_cov, "NULL operand");
    const FRefs &R = b->R;
    const size_t K = (size_t)R.K, KK = K * K;
    for (int h = 0; h < R.H; h++) {
        APMF_TRY(apmf_d2d(ctx, R.mean[0] + h * K, d_mean + (per_hypothesis ? h * K : 0), K * 8));
        APMF_TRY(apmf_d2d(ctx, R.cov[0] + h * KK, d_cov + (per_hypothesis ? h * KK : 0), KK * 8));
        b->slot[h] = 0;
        b->fvalid[h] = 0;
        b->old_kl[h] = NAN;
    }
    return fb_push_state(b);
}

// one extra rating per hypothesis (h_i[h] < 0: none); NULL clears them all
int apmf_full_batch_set_extra(apmf_ctx *ctx, apmf_full_batch *b, const int32_t *h_i, const int32_t *h_j,
                              const double *h_v)
{
    APMF_REQUIRE(ctx && b, "NULL operand");
    const FRefs &R = b->R;
    for (int h = 0; h < R.H; h++) {
        b->hi[h] = h_i ? h_i[h] : -1;
        b->hj[h] = h_i ? h_j[h] : -1;
        b->hv[h] = h_i ? h_v[h] : 0.;
        APMF_REQUIRE(b->hi[h] < R.n && b->hj[h] < R.m && (b->hi[h] < 0) == (b->hj[h] < 0),
                     "hypothesis %d: entry (%d, %d) outside %d x %d", h, b->hi[h], b->hj[h], R.n, R.m);
        if (b->hi[h] < 0) b->hi[h] = b->hj[h] = -1;
        b->old_kl[h] = NAN;               // the KL depends on the ratings (the factor of the covariance does not)
    }
    APMF_TRY(apmf_h2d(ctx, b->d_ei, b->hi.data(), (size_t)R.H * 4));
    APMF_TRY(apmf_h2d(ctx, b->d_ej, b->hj.data(), (size_t)R.H * 4));
    return apmf_h2d(ctx, b->d_ev, b->hv.data(), (size_t)R.H * 8);
}

int apmf_full_batch_get_state(apmf_ctx *ctx, apmf_full_batch *b, int h, double *d_mean, double *d_cov)
{
    APMF_REQUIRE(ctx && b && h >= 0 && h < b->R.H, "bad hypothesis index");
    const FRefs &R = b->R;
    const size_t K = (size_t)R.K, KK = K * K;
    if (d_mean) APMF_TRY(apmf_d2d(ctx, d_mean, R.mean[b->slot[h]] + h * K, K * 8));
    if (d_cov) APMF_TRY(apmf_d2d(ctx, d_cov, R.cov[b->slot[h]] + h * KK, KK * 8));
    return APMF_OK;
}

int apmf_full_batch_fit_stats(const apmf_full_batch *b, int64_t *iterations, int64_t *trials, int64_t *projected)
{
    APMF_REQUIRE(b, "NULL batch");
    if (iterations) *iterations = b->st_iters;
    if (trials) *trials = b->st_trials;
    if (projected) *projected = b->st_projected;
    return APMF_OK;
}

// ActivePMF.fit_normal_kls (active_pmf.py:250-288) for every hypothesis.  h_lr / h_old_kl (H each): in = learning
// rate and KL to resume from (NaN: evaluate it), out = where the search stands; max_steps < 0: to convergence.
// h_kl_trace (H x trace_cap, nullable): the KL after every accepted step of this call.
int apmf_full_batch_fit(apmf_ctx *ctx, apmf_full_batch *b, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                        double min_eig, double *h_lr, double *h_old_kl, int64_t max_steps, double *h_kl_trace,
                        int64_t trace_cap, int64_t *h_steps, int *h_converged)
{
    APMF_REQUIRE(ctx && b && h_lr && h_old_kl, "NULL operand");
    APMF_REQUIRE(ctx == b->ctx, "batch belongs to another context");
    APMF_REQUIRE(sigma_sq != 0. && sigma_u_sq != 0. && sigma_v_sq != 0., "zero variance");
    APMF_REQUIRE(trace_cap == 0 || h_kl_trace, "h_kl_trace is NULL");
    FRefs &R = b->R;
    const int H = R.H;
    const size_t K = (size_t)R.K, KK = K * K;
    hipStream_t st = ctx->stream;
    const unsigned gKK = (unsigned)((KK + 255) / 256);
    std::vector<long long> steps(H, 0);
    std::vector<char> converged(H, 0), need_grad(H, 1);
    std::vector<double> kl(H, 0.), ld(H, 0.);
    b->st_iters = b->st_trials = b->st_projected = 0;
    for (int h = 0; h < H; h++) {
        b->lr[h] = h_lr[h];
        b->old_kl[h] = h_old_kl[h];
    }
    APMF_TRY(fb_push_state(b));
    // ---- starting points: the factor of the current covariance (the gradient's inverse comes from it) and the KL
    {
        std::vector<int> init;
        for (int h = 0; h < H; h++)
            if (!b->fvalid[h] || b->old_kl[h] != b->old_kl[h]) init.push_back(h);
        if (!init.empty()) {
            APMF_TRY(fb_upload_sel(b, init));
            hipLaunchKernelGGL(kf_copy_cur, dim3(gKK, (unsigned)init.size()), dim3(256), 0, st, R, b->d_sel);
            APMF_CHECK_LAUNCH();
            APMF_CHECK_HIP(hipMemsetAsync(R.info, 0, (size_t)2 * H * sizeof(int), st));
            APMF_TRY(fb_factor(b, init, false));
            std::vector<int> info((size_t)2 * H);
            APMF_TRY(apmf_d2h(ctx, info.data(), R.info, info.size() * sizeof(int)));
            std::vector<char> ok(init.size());
            for (size_t q = 0; q < init.size(); q++) {
                ok[q] = info[2 * init[q] + 1] == 0;
                b->fvalid[init[q]] = ok[q];
            }
            APMF_TRY(fb_values(b, init, 0, ok, sigma_sq, sigma_u_sq, sigma_v_sq, kl, ld));
            for (int h : init) {
                if (b->old_kl[h] != b->old_kl[h]) b->old_kl[h] = kl[h];
                b->cur_ld[h] = ld[h];
            }
        }
    }
    std::vector<int> active, grad, info((size_t)2 * H);
    for (;;) {
        active.clear();
        grad.clear();
        for (int h = 0; h < H; h++)
            if (!converged[h] && (max_steps < 0 || steps[h] < max_steps)) {
                active.push_back(h);
                if (need_grad[h]) grad.push_back(h);
            }
        if (active.empty()) break;
        b->st_iters++;
        b->st_trials += (long)active.size();
        // ---- gradient at the current point
        if (!grad.empty()) {
            std::vector<int> fast, slow;
            for (int h : grad) (b->fvalid[h] ? fast : slow).push_back(h);
            if (!fast.empty()) {
                APMF_TRY(fb_upload_sel(b, fast));
                const unsigned nf = (unsigned)fast.size();
                for (int k = 0; k + 1 < R.nbk; k++)
                    hipLaunchKernelGGL(kf_trinv_step, dim3((R.nbk - k - 1) * (k + 1), nf), dim3(256), 0, st, R, b->d_sel, k);
                hipLaunchKernelGGL(kf_lauum, dim3(R.nbk * (R.nbk + 1) / 2, nf), dim3(256), 0, st, R, b->d_sel);
                APMF_CHECK_LAUNCH();
            }
            for (int h : slow)      // no factor (a projected or indefinite point): the general inverse
                APMF_TRY(apmf_dense_inv_logdet(ctx, R.cov[b->slot[h]] + h * KK, R.K, R.gcov + h * KK, nullptr));
            APMF_TRY(fb_upload_sel(b, grad));
            const unsigned ng = (unsigned)grad.size();
            hipLaunchKernelGGL(kf_grad_init, dim3(gKK, ng), dim3(256), 0, st, R, b->d_sel, sigma_u_sq, sigma_v_sq);
            hipLaunchKernelGGL((kf_grad_rows<0>), dim3(R.n, ng), dim3(64), 0, st, R, b->d_sel, b->r->ptr[0], b->r->idx[0],
                               b->r->val[0], 1. / sigma_sq);
            hipLaunchKernelGGL((kf_grad_rows<1>), dim3(R.m, ng), dim3(64), 0, st, R, b->d_sel, b->r->ptr[1], b->r->idx[1],
                               b->r->val[1], 1. / sigma_sq);
            APMF_CHECK_LAUNCH();
            for (int h : grad) need_grad[h] = 0;
        }
        // ---- trial points, their two factorisations, their KL
        APMF_TRY(fb_upload_sel(b, active));
        const unsigned na = (unsigned)active.size();
        hipLaunchKernelGGL(kf_trial, dim3(gKK, na), dim3(256), 0, st, R, b->d_sel, min_eig);
        APMF_CHECK_LAUNCH();
        APMF_CHECK_HIP(hipMemsetAsync(R.info, 0, (size_t)2 * H * sizeof(int), st));
        APMF_TRY(fb_factor(b, active, true));
        APMF_TRY(apmf_d2h(ctx, info.data(), R.info, info.size() * sizeof(int)));
        std::vector<char> ok(active.size());
        std::vector<int> redo;
        for (size_t q = 0; q < active.size(); q++) {
            const int h = active[q];
            if (info[2 * h] != 0) {
                // not certified: project_psd's eigendecomposition decides (and rebuilds the matrix if it must)
                int projected = 0;
                APMF_TRY(apmf_dense_project_psd(ctx, R.cov[b->slot[h] ^ 1] + h * KK, R.K, min_eig, &projected));
                b->st_projected += projected;
                redo.push_back(h);
            }
            ok[q] = info[2 * h + 1] == 0;
        }
        if (!redo.empty()) {
            // the factor of the (possibly rebuilt) matrix, for its log det and a later gradient
            for (int h : redo)
                APMF_TRY(apmf_d2d(ctx, R.F + h * KK, R.cov[b->slot[h] ^ 1] + h * KK, KK * 8));
            APMF_CHECK_HIP(hipMemsetAsync(R.info, 0, (size_t)2 * H * sizeof(int), st));
            APMF_TRY(fb_factor(b, redo, false));
            std::vector<int> info2((size_t)2 * H);
            APMF_TRY(apmf_d2h(ctx, info2.data(), R.info, info2.size() * sizeof(int)));
            for (size_t q = 0; q < active.size(); q++)
                if (std::find(redo.begin(), redo.end(), active[q]) != redo.end()) ok[q] = info2[2 * active[q] + 1] == 0;
        }
        APMF_TRY(fb_values(b, active, 1, ok, sigma_sq, sigma_u_sq, sigma_v_sq, kl, ld));
        // ---- accept / reject (active_pmf.py:266-288)
        for (size_t q = 0; q < active.size(); q++) {
            const int h = active[q];
            const double new_kl = kl[h], old = b->old_kl[h];
            if (new_kl < old) {
                b->slot[h] ^= 1;
                b->fvalid[h] = ok[q];
                b->cur_ld[h] = ld[h];
                b->lr[h] *= 1.25;
                if (old - new_kl < .005) converged[h] = 1;
                if (h_kl_trace && steps[h] < trace_cap) h_kl_trace[(size_t)h * trace_cap + steps[h]] = new_kl;
                steps[h]++;
                b->old_kl[h] = new_kl;
                need_grad[h] = 1;
            } else {
                b->lr[h] *= .5;
                // F now holds the rejected trial's factor, not the current point's: the gradient is still in place
                // (the point did not move), but a resumed call must re-factor before it can form a new one
                b->fvalid[h] = 0;
                if (b->lr[h] < 1e-10) converged[h] = 1;
            }
        }
        APMF_TRY(fb_push_state(b));
    }
    for (int h = 0; h < H; h++) {
        h_lr[h] = b->lr[h];
        h_old_kl[h] = b->old_kl[h];
        if (h_steps) h_steps[h] = steps[h];
        if (h_converged) h_converged[h] = converged[h];
    }
    APMF_CHECK_HIP(hipStreamSynchronize(st));
    return APMF_OK;
}

// kl_divergence() of every current point (no step is taken)
int apmf_full_batch_kl(apmf_ctx *ctx, apmf_full_batch *b, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                       double *h_kl)
{
    APMF_REQUIRE(ctx && b && h_kl, "NULL operand");
    const int H = b->R.H;
    std::vector<double> lr(H, 0.), kl(H, NAN);
    for (int h = 0; h < H; h++) b->old_kl[h] = NAN;
    APMF_TRY(apmf_full_batch_fit(ctx, b, sigma_sq, sigma_u_sq, sigma_v_sq, 0., lr.data(), kl.data(), 0, nullptr, 0,
                                 nullptr, nullptr));
    for (int h = 0; h < H; h++) h_kl[h] = kl[h];
    return APMF_OK;
}

// log|det cov| of every current point (_approx_entropy, active_pmf.py:519-521) as the last fit / kl call left it
int apmf_full_batch_logdet(apmf_ctx *ctx, apmf_full_batch *b, double *h_logdet)
{
    APMF_REQUIRE(ctx && b && h_logdet, "NULL operand");
    for (int h = 0; h < b->R.H; h++) {
        APMF_REQUIRE(b->cur_ld[h] == b->cur_ld[h], "hypothesis %d: no fit or kl call has evaluated its current point", h);
        h_logdet[h] = b->cur_ld[h];
    }
    return APMF_OK;
}

// _total_variance() of every current point: the K2-full sweep with the fused sum, one launch per hypothesis
int apmf_full_batch_total_variance(apmf_ctx *ctx, apmf_full_batch *b, double *h_out)
{
    APMF_REQUIRE(ctx && b && h_out, "NULL operand");
    const FRefs &R = b->R;
    const size_t K = (size_t)R.K, KK = K * K;
    const int ld = round_up(R.m, APMF_GRID_ALIGN);
    double *d_sums = nullptr;
    APMF_TRY(apmf_malloc(ctx, (size_t)R.H * 8, (void **)&d_sums));
    int rc = APMF_OK;
    for (int h = 0; h < R.H && rc == APMF_OK; h++)
        rc = apmf_full_score_grid_f64(ctx, R.n, R.m, R.d, ld, R.mean[b->slot[h]] + h * K, R.cov[b->slot[h]] + h * KK, nullptr,
                                      nullptr, nullptr, d_sums + h, 0, nullptr, nullptr);
    if (rc == APMF_OK) rc = apmf_d2h(ctx, h_out, d_sums, (size_t)R.H * 8);
    (void)apmf_free(ctx, d_sums);
    return rc;
}

} // extern "C"
