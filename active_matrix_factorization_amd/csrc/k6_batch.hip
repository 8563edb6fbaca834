// k6_batch.hip - H matrix-normal variational fits advanced by ONE chain of launches
// (the batched lookahead driver; part of libapmf_dense.so).
//
// What it replaces.  The reference evaluates a lookahead criterion by running, for every
// hypothesis "entry (i, j) turns out to be v" (|pool| x |rating values| of them per query step,
// mn_active_pmf.py:660-672 mapped over the pool at :749-755):
//     deepcopy(model) -> add_rating(i, j, v) -> fit_normal() -> fn(model)
// fit_normal (mn_active_pmf.py:236-288) is projected gradient descent on the KL with a
// x1.25 / x0.5 learning-rate search: a latency-bound chain of small dependent kernels
// (Cholesky panels, K5 row kernels, d x d solves) that keeps a handful of the 256 CUs busy.
// Here the H fits are independent problems of the same shape, so every kernel of that chain
// gets a second grid dimension - the hypothesis - and the chain is walked once for all of them:
//   * the H states (mean, cov_useritems, cov_latents), their gradients, trial points and
//     Cholesky factors live in HBM for the whole fit (5 N^2 + 3 N d + 5 d^2 doubles each);
//     device-side POINTER TABLES say which buffer currently is "current" and which "trial", so
//     accepting a trial is a pointer swap made by the kernel that takes the decision;
//   * a hypothesis differs from the base model by ONE rating: the kernels walk the shared CSR
//     of the base ratings and take the extra (i, j, v) of their hypothesis as the last entry of
//     user row i / item row j - exactly where add_rating() + a rebuilt CSR would put it, so the
//     sums are bit-identical to the one-model path;
//   * the accept / reject decision, learning rate, KL, step counter and convergence flag of
//     every hypothesis are device state (k_decide); the host reads one small status record per
//     hypothesis per trial and compacts the list of still-active hypotheses (`sel`), so that
//     converged fits cost nothing;
//   * the rare exits of the fast path (a projection that really has to clip eigenvalues, a
//     factorisation that fails) are handled per hypothesis by the single-matrix routines of
//     dense.hip and re-enter the batch.
// A hypothesis' arithmetic never depends on which other hypotheses share its launches: H = 1
// (MNActivePMF.fit_normal itself runs through here) and H = 2048 give bit-identical fits.
//
// Inverse of the covariance for the entropy term of the gradient: from the Cholesky factor the
// trial evaluation left behind, by our own kernels (rocSOLVER's potri is one more library call
// chain per hypothesis and is not batched over pointer tables): W = L^-1 by block elimination
// (one launch per 64-column panel, every launch a set of independent 64^3 products), then
// S^-1 = W' W in one launch, with the -(d/2)(2 - delta) scaling of matrix_normal_exps_cy.pyx:477-480
// applied as the tiles are written.
#include <math.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

// every kernel launch of the batch engine is counted (apmf_mn_batch_launch_count): the chain of dependent
// launches per hypothesis-step is what bounds small batches, so it is a tracked number
static std::atomic<long long> g_k6_launches{0};
#define APMF_LAUNCH(...)                                              \
    do {                                                              \
        g_k6_launches.fetch_add(1, std::memory_order_relaxed);        \
        hipLaunchKernelGGL(__VA_ARGS__);                              \
    } while (0)

#include "chol.h"
#include "common.h"

extern "C" int apmf_dense_handle(apmf_ctx *ctx, void **out);
extern "C" int apmf_dense_lane_handle(apmf_ctx *ctx, int lane, hipStream_t stream, void **out);

extern "C" int apmf_dense_project_psd(apmf_ctx *ctx, double *d_mat, int N, double min_eig, int *h_projected);
extern "C" int apmf_dense_inv_logdet(apmf_ctx *ctx, const double *d_mat, int N, double *d_inv, double *h_logdet);
extern "C" int apmf_mn_score_grid_f64(apmf_ctx *ctx, int B, int n, int m, int d, int ld, const double *d_mean,
                                      const double *d_cross, const double *d_diag, const double *d_omega,
                                      const uint8_t *d_mask, double *d_var, double *d_pmean, double *d_sum,
                                      int want, double *d_best_val, int64_t *d_best_idx);
extern "C" int apmf_mn_score_grid_f32(apmf_ctx *ctx, int B, int n, int m, int d, int ld, const float *d_mean,
                                      const float *d_cross, const float *d_diag, const float *d_omega,
                                      const uint8_t *d_mask, float *d_var, float *d_pmean, double *d_sum,
                                      int want, float *d_best_val, int64_t *d_best_idx);

namespace {

// ---- pointer tables ------------------------------------------------------------------------
enum {
    T_M_CUR, T_M_CAND, T_M_GRAD,
    T_S_CUR, T_S_CAND, T_S_GRAD, T_S_F, T_S_W,
    T_O_CUR, T_O_CAND, T_O_GRAD, T_OI_CUR, T_OI_CAND,
    T_COUNT
};

enum { ST_ACTIVE = 0, ST_CONVERGED = 1, ST_PAUSED = 2, ST_SLOW = 3 };
enum { SLOW_CERT_S = 1, SLOW_FACT_S = 2, SLOW_CERT_O = 4, SLOW_INV_O = 8 };

struct HypState {
    double lr, old_kl, new_kl;
    double ld_s, ld_o;            // log-determinants of the point just evaluated
    double cur_ld_s, cur_ld_o;    // ... of the current point (entropy criterion)
    long long steps;              // accepted steps of this call
    long long max_steps;          // < 0: no limit
    int status, need_grad, slow, accepted;
    int cert_skipped, inv_ok;     // trial certificates settled by the eigenvalue bound; the bound's input is current
    int bnd_ok;                   // the structured bound's inputs (bndp) belong to the current gradient
    int cur_sym;                  // cov_useritems of the current point is EXACTLY symmetric (an accepted fast-path trial)
};

struct OmegaConsts {
    double T, D2, off2, sdd, t2;
    double pad[3];
};

struct Refs {
    double **tab;         // [T_COUNT][H]
    HypState *hs;         // [H]
    int *info;            // [2H]: 2h certificate factorisation (W), 2h + 1 factor of S (F)
    double *ores;         // [H][4]: cert_bad, inv_bad, logdet of cov_latents
    double *linvF;        // [H][nbk][64 * 64] inverses of F's diagonal blocks (kept for L^-1)
    double *linvW;        // [H][64 * 64]
    ChJob *jobs;          // [2H]
    // K5 per-row work arrays, [H] x ...
    double *xoff, *crow, *qrow, *sdiag, *dgv, *racc_u, *racc_v, *part, *out4;
    OmegaConsts *oc;
    double **ptrs;        // [H] scratch pointer array (batched eigendecomposition)
    double *evals, *evE;  // [H][N] eigenvalues / rocSOLVER's E
    int *evinfo;          // [H]
    double *trace;        // [H][trace_cap]
    double *invp;         // [H][nbk (nbk + 1) / 2] partial sums of |S^-1|_F^2 at the current point (kb_lauum)
    double *stepp;        // [H][nt32 (nt32 + 1) / 2] partial sums of |lr g_S|_F^2 of the trial (kb_trial_S)
    double *c2row;        // [H][n] per user row: sum of the squared cross entries of g_S's rating part (kb_kl_rows)
    double *bndp;         // [H][ceil(N / 256)][2] per kb_finish_rows block: max_t q_t, sum of c2row (cert_by_bound)
    int *skipword;        // one int, 1: the info word of a factorisation job that is not to run
    int *ei, *ej;         // extra rating of the hypothesis (ei < 0: none)
    double *ev;
    int H, n, m, d, N, nbk, trace_cap;
};

__device__ __forceinline__ double *tabp(const Refs &R, int which, int h) { return R.tab[(size_t)which * R.H + h]; }

// ---- small helpers shared with K5 ------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp6(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int DP>
__device__ __forceinline__ double gsum(double v)
{
    if (DP >= 2) v += dpp6<0xB1>(v);
    if (DP >= 4) v += dpp6<0x4E>(v);
    if (DP >= 8) v += dpp6<0x141>(v);
    if (DP >= 16) v += dpp6<0x140>(v);
    if (DP >= 32) v += __shfl_xor(v, 16, 64);
    if (DP >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

template <int DP>
__device__ __forceinline__ double xgsum(double v)
{
#pragma unroll
    for (int o = DP; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- state plumbing ----------------------------------------------------------------------------
// table[which][h] <- src (+ h * stride): broadcast (stride 0) or per-hypothesis initial states
__global__ __launch_bounds__(256) void kb_fill(const Refs R, int which, const double *__restrict__ src,
                                               size_t stride, size_t count)
{
    const int h = blockIdx.y;
    double *__restrict__ dst = tabp(R, which, h);
    const double *__restrict__ s = src + (size_t)h * stride;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) dst[i] = s[i];
}

// dst (+ y * stride) <- table[which][sel[y]]
__global__ __launch_bounds__(256) void kb_gather(const Refs R, const int *__restrict__ sel, int which,
                                                 double *__restrict__ dst, size_t stride, size_t count)
{
    const int h = sel ? sel[blockIdx.y] : blockIdx.y;
    const double *__restrict__ s = tabp(R, which, h);
    double *__restrict__ o = dst + (size_t)blockIdx.y * stride;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) o[i] = s[i];
}

// table[dst][h] <- table[src][h] for h in sel
__global__ __launch_bounds__(256) void kb_copy(const Refs R, const int *__restrict__ sel, int src, int dst,
                                               size_t count)
{
    const int h = sel[blockIdx.y];
    const double *__restrict__ s = tabp(R, src, h);
    double *__restrict__ o = tabp(R, dst, h);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) o[i] = s[i];
}

// ---- trial point ---------------------------------------------------------------------------------
// cov_useritems: cand = sym(cur - lr * grad) with NumPy's two roundings per entry, then
// (A + A') / 2 like project_psd's first line (mn_active_pmf.py:48).  The same pass drops the
// two matrices the projection test and the KL need into place: W = cand - (min_eig + delta) I (its
// Cholesky factorisation succeeding certifies "no eigenvalue below min_eig": nothing to clip)
// and F = cand (its factor gives log det and, if the trial is accepted, the inverse).
// 32 x 32 tiles on or above the diagonal; the mirrored tile goes through LDS so that every
// global access is a row segment.
// materialize: also write F and W (the factorisations that follow run in place); 0: they read cand on their first
// touch of every tile (ChJob::src, block-column route).  When the current point and the gradient are EXACTLY
// symmetric - the point an accepted fast-path trial (cur_sym), the gradient kb_lauum's (inv_ok) - the mirrored tile
// holds the same bits, (x + x) / 2 = x, and it is not read: same result, half the reads.
__global__ __launch_bounds__(256) void kb_trial_S(const Refs R, const int *__restrict__ sel, double min_eig,
                                                  int materialize, int allow_sym)
{
    __shared__ double A[32][33], B[32][33];
    const int h = sel[blockIdx.y];
    const int N = R.N;
    int tj, ti;
    tri_tile((int)blockIdx.x, tj, ti);       // ti <= tj
    const double lr = R.hs[h].lr;
    const bool sym = allow_sym && R.hs[h].cur_sym && R.hs[h].inv_ok;      // block-uniform
    const double *__restrict__ cur = tabp(R, T_S_CUR, h);
    const double *__restrict__ g = tabp(R, T_S_GRAD, h);
    double *__restrict__ cand = tabp(R, T_S_CAND, h);
    double *__restrict__ F = tabp(R, T_S_F, h);
    double *__restrict__ W = tabp(R, T_S_W, h);
    const int i0 = ti * 32, j0 = tj * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    double step2 = 0.;        // this block's share of |lr g|_F^2 >= |cur - cand|_F^2 (both tiles of the pair)
    if (sym) {
        double gv[4], cv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = ty + 8 * q;
            const size_t ea = (size_t)min(i0 + r, N - 1) * N + min(j0 + tx, N - 1);
            gv[q] = g[ea];
            cv[q] = cur[ea];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = ty + 8 * q;
            const bool oka = i0 + r < N && j0 + tx < N;
            const double sa = oka ? __dmul_rn(lr, gv[q]) : 0.;
            A[r][tx] = oka ? __dsub_rn(cv[q], sa) : 0.;
            step2 += ti != tj ? 2. * (sa * sa) : sa * sa;
        }
    } else {
        // all 16 loads of the thread are requested (at clamped addresses) before any is used
        double gv[4][2], cv[4][2];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = ty + 8 * q;
            const size_t ea = (size_t)min(i0 + r, N - 1) * N + min(j0 + tx, N - 1);
            const size_t eb = (size_t)min(j0 + r, N - 1) * N + min(i0 + tx, N - 1);
            gv[q][0] = g[ea];
            cv[q][0] = cur[ea];
            gv[q][1] = g[eb];
            cv[q][1] = cur[eb];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = ty + 8 * q;
            const bool oka = i0 + r < N && j0 + tx < N, okb = j0 + r < N && i0 + tx < N;
            const double sa = oka ? __dmul_rn(lr, gv[q][0]) : 0.;
            const double sb = okb ? __dmul_rn(lr, gv[q][1]) : 0.;
            A[r][tx] = oka ? __dsub_rn(cv[q][0], sa) : 0.;
            B[r][tx] = okb ? __dsub_rn(cv[q][1], sb) : 0.;
            step2 += sa * sa + (ti != tj ? sb * sb : 0.);
        }
    }
    {
        __shared__ double red[4];
        step2 = wave_sum(step2);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = step2;
        __syncthreads();
        if (threadIdx.x == 0) R.stepp[(size_t)h * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    if (ti != tj) {
        for (int r = ty; r < 32; r += 8) {
            // upper tile element (i, j), i < j, and the element (j0 + r, i0 + tx) of the lower tile
            const int i = i0 + r, j = j0 + tx;
            if (i < N && j < N) cand[(size_t)i * N + j] = sym ? A[r][tx] : (A[r][tx] + B[tx][r]) / 2;
            const int i2 = j0 + r, j2 = i0 + tx;
            if (i2 < N && j2 < N) {
                const double v = sym ? A[tx][r] : (A[tx][r] + B[r][tx]) / 2;
                cand[(size_t)i2 * N + j2] = v;
                if (materialize) {
                    F[(size_t)i2 * N + j2] = v;
                    W[(size_t)i2 * N + j2] = v;
                }
            }
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, j = i0 + tx;
            if (i >= N || j >= N) continue;
            // (upper + lower) / 2 off the diagonal; the diagonal itself is left alone
            const double v = r == tx ? A[r][tx] : (r < tx ? (A[r][tx] + A[tx][r]) / 2 : (A[tx][r] + A[r][tx]) / 2);
            cand[(size_t)i * N + j] = v;
            if (materialize && j <= i) {
                F[(size_t)i * N + j] = v;
                W[(size_t)i * N + j] = i == j ? v - (min_eig + 1e-10 * fmax(fabs(v), fabs(min_eig))) : v;
            }
        }
    }
}

__device__ __forceinline__ void make_jobs_one(const Refs &R, int y, int h, int mode, ChJob *__restrict__ jobs,
                                              int from_cand = 0, double min_eig = 0.);

// mean and cov_latents parts of the trial point (cov_latents symmetrised like project_psd does);
// jobs != nullptr: also the hypothesis' pair of factorisation jobs (certificate + factor)
// The certificate of a trial point - no eigenvalue of cand below min_eig - without its factorisation, when
// the step is short: l_min(cand) >= l_min(cur) - |cur - cand|_2 (Weyl) >= 1 / |cur^-1|_F - |lr g|_F, and the
// gradient at the current point has just computed cur^-1 (kb_lauum leaves the partial sums of its squared
// entries).  If that is >= min_eig the shifted factorisation is not run (its job gets an info word that
// says "skip", the certificate's own stays 0 = passed); either way the trial's arithmetic is the same.
__device__ __forceinline__ int cert_by_bound(const Refs &R, int h, int nstep, double min_eig)
{
    // one wave
    const int lane = threadIdx.x & 63;
    const HypState &s = R.hs[h];
    double a = 0.;
    int ok = 0;
    if (s.inv_ok) {
        const int ninv = R.nbk * (R.nbk + 1) / 2;
        double b = 0.;
        for (int i = lane; i < ninv; i += 64) a += R.invp[(size_t)h * ninv + i];
        for (int i = lane; i < nstep; i += 64) b += R.stepp[(size_t)h * nstep + i];
        a = wave_sum(a);
        b = wave_sum(b);
        ok = a > 0. && .99 / sqrt(a) - sqrt(b) >= min_eig * (1. + 1e-6);
    }
    // The structured bound (kb_finish_rows has the algebra): l_min(cand) >= l_min(cur + 2c cur^-1) - l_max(lr Rs + c D).
    // cur is positive definite (it has a Cholesky factor, or came out of the projection), so every eigenvalue of the
    // first bracket is l + 2c / l >= 2 sqrt(2c) - and >= L + 2c / L when all of cur's are >= L >= sqrt(2c), L the
    // Frobenius bound above.  Rigorous like the plain Weyl bound, and it does not lose the entropy term's size: it
    // settles most of the trials the plain one cannot.  1 % is kept to spare on either side.
    if (!ok && s.bnd_ok && s.lr > 0.) {
        const int nfr = (R.N + 255) / 256;
        double qmax = -INFINITY, c2 = 0.;
        for (int i = lane; i < nfr; i += 64) {
            qmax = fmax(qmax, R.bndp[((size_t)h * nfr + i) * 2]);
            c2 += R.bndp[((size_t)h * nfr + i) * 2 + 1];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) qmax = fmax(qmax, __shfl_xor(qmax, o, 64));
        c2 = wave_sum(c2);
        const double c = s.lr * R.d / 2.;
        double low = 2. * sqrt(2. * c);
        if (s.inv_ok && a > 0.) {
            const double L = .99 / sqrt(a);
            if (L >= 1.01 * sqrt(2. * c)) low = fmax(low, L + 2. * c / L);
        }
        const double up = s.lr * (fmax(qmax, 0.) + sqrt(c2));
        ok = c2 == c2 && qmax == qmax && .99 * low - 1.01 * up >= min_eig * (1. + 1e-6);
    }
    return ok;
}

// (bound_eig: the threshold the eigenvalue bounds are held to - infinity switches them off; from_cand / min_eig: see
// make_jobs_one)
__global__ __launch_bounds__(256) void kb_trial_small(const Refs R, const int *__restrict__ sel,
                                                      ChJob *__restrict__ jobs, int nstep, double bound_eig,
                                                      int from_cand, double min_eig)
{
    const int h = sel[blockIdx.y];
    if (jobs && blockIdx.x == 0 && threadIdx.x < 64) {
        const int skip = cert_by_bound(R, h, nstep, bound_eig);
        if (threadIdx.x == 0) {
            make_jobs_one(R, (int)blockIdx.y, h, 2, jobs, from_cand, min_eig);
            if (skip) {
                jobs[2 * blockIdx.y].info = R.skipword;
                jobs[2 * blockIdx.y].A = nullptr;        // its blocks leave before they request anything
                R.hs[h].cert_skipped++;
            }
        }
    }
    const double lr = R.hs[h].lr;
    const size_t nm = (size_t)R.N * R.d, dd = (size_t)R.d * R.d;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < nm) {
        tabp(R, T_M_CAND, h)[t] = __dsub_rn(tabp(R, T_M_CUR, h)[t], __dmul_rn(lr, tabp(R, T_M_GRAD, h)[t]));
    } else if (t < nm + dd) {
        const int e = (int)(t - nm), k = e / R.d, l = e % R.d;
        const double *cur = tabp(R, T_O_CUR, h), *g = tabp(R, T_O_GRAD, h);
        const double a = __dsub_rn(cur[k * R.d + l], __dmul_rn(lr, g[k * R.d + l]));
        const double b = __dsub_rn(cur[l * R.d + k], __dmul_rn(lr, g[l * R.d + k]));
        tabp(R, T_O_CAND, h)[e] = k == l ? a : (k < l ? (a + b) / 2 : (b + a) / 2);
    }
}

// ---- factorisation jobs --------------------------------------------------------------------------
// mode 2: certificate (W) + factor (F) per selected hypothesis; mode 1: factor only
// from_cand: F / W have not been written - the factorisations read the trial point (T_S_CAND) on their first touch
__device__ __forceinline__ void make_jobs_one(const Refs &R, int y, int h, int mode, ChJob *__restrict__ jobs,
                                              int from_cand, double min_eig)
{
    const size_t blk = (size_t)CH_NB * CH_NB;
    const double *src = from_cand ? tabp(R, T_S_CAND, h) : nullptr;
    ChJob f;
    f.A = tabp(R, T_S_F, h);
    f.info = R.info + 2 * h + 1;
    f.linv = R.linvF + (size_t)h * R.nbk * blk;
    f.linv_keep = 1;
    f.shifted = 0;
    f.src = src;
    f.eig = 0.;
    R.info[2 * h + 1] = 0;
    if (mode == 2) {
        ChJob w;
        w.A = tabp(R, T_S_W, h);
        w.info = R.info + 2 * h;
        w.linv = R.linvW + (size_t)h * blk;
        w.linv_keep = 0;
        w.shifted = src ? 1 : 0;
        w.src = src;
        w.eig = min_eig;
        R.info[2 * h] = 0;
        jobs[2 * y] = w;
        jobs[2 * y + 1] = f;
    } else {
        R.info[2 * h] = 0;
        jobs[y] = f;
    }
}

__global__ void kb_make_jobs(const Refs R, const int *__restrict__ sel, int nsel, int mode, ChJob *__restrict__ jobs)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y < nsel) make_jobs_one(R, y, sel[y], mode, jobs);
}

// hs[h].ld_s = 2 sum log F[i][i]   (one wave; the diagonal requested 8 entries per lane at a time)
__device__ __forceinline__ void logdet_body(const Refs &R, int h)
{
    const double *__restrict__ L = tabp(R, T_S_F, h);
    const int N = R.N, lane = threadIdx.x & 63;
    double s = 0;
    for (int base = 0; base < N; base += 64 * 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = min(base + q * 64 + lane, N - 1);
            v[q] = L[(size_t)i * N + i];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) s += base + q * 64 + lane < N ? log(v[q]) : 0.;
    }
    s = wave_sum(s);
    if (lane == 0) R.hs[h].ld_s = 2. * s;
}

__global__ __launch_bounds__(64) void kb_logdet(const Refs R, const int *__restrict__ sel)
{
    logdet_body(R, sel[blockIdx.x]);
}

// d x d covariance (d <= 64), one wave per hypothesis, in LDS: optionally the projection
// certificate (Cholesky of O - (min_eig + delta) I), then Cholesky of O itself -> log det and
// O^-1.  ores[h] = {certificate failed, factorisation failed, log det}.
__device__ __forceinline__ void small_spd_body(const Refs &R, int h, int src, int dst, int do_cert, double min_eig)
{
    extern __shared__ double sm[];
    const int N = R.d, P = N + 1;
    double *L = sm, *X = sm + (size_t)N * P;
    __shared__ int bad;
    const double *__restrict__ A = tabp(R, src, h);
    double *__restrict__ inv_out = tabp(R, dst, h);
    double *__restrict__ res = R.ores + (size_t)h * 4;
    const int t = threadIdx.x;
    double cert_bad = 0.;
    for (int pass = do_cert ? 0 : 1; pass < 2; pass++) {
        const bool shift = pass == 0;
        __syncthreads();
        if (t == 0) bad = 0;
        __syncthreads();
        for (int e = t; e < N * N; e += 64) {
            const int i = e / N, j = e % N;
            double v = A[e];
            if (A[(size_t)j * N + i] != v) bad = 1;
            if (shift && i == j) v -= min_eig + 1e-10 * fmax(fabs(v), fabs(min_eig));
            L[i * P + j] = v;
        }
        __syncthreads();
        // right-looking Cholesky, lane i owns row i
        for (int j = 0; j < N && !bad; j++) {
            __syncthreads();
            const double djj = L[j * P + j];
            if (!(djj > 0.)) {
                if (t == 0) bad = 1;
                __syncthreads();
                break;
            }
            const double rj = sqrt(djj);
            __syncthreads();
            if (t == j) L[j * P + j] = rj;
            if (t > j && t < N) L[t * P + j] /= rj;
            __syncthreads();
            if (t > j && t < N) {
                const double lij = L[t * P + j];
                for (int k = j + 1; k <= t; k++) L[t * P + k] -= lij * L[k * P + j];
            }
        }
        __syncthreads();
        if (shift) cert_bad = bad ? 1. : 0.;
    }
    if (bad) {
        if (t == 0) { res[0] = cert_bad; res[1] = 1.; res[2] = 0.; }
        return;
    }
    if (t == 0) {
        double s = 0;
        for (int i = 0; i < N; i++) s += log(L[i * P + i]);
        res[0] = cert_bad;
        res[1] = 0.;
        res[2] = 2. * s;
        R.hs[h].ld_o = 2. * s;
    }
    // X = L^-1 (lower), column t by forward substitution
    if (t < N) {
        for (int i = 0; i < N; i++) {
            double v = (i == t) ? 1. : 0.;
            for (int k = t; k < i; k++) v -= L[i * P + k] * X[k * P + t];
            X[i * P + t] = i < t ? 0. : v / L[i * P + i];
        }
    }
    __syncthreads();
    // O^-1 = X' X
    for (int e = t; e < N * N; e += 64) {
        const int i = e / N, j = e % N;
        double v = 0;
        for (int k = (i > j ? i : j); k < N; k++) v += X[k * P + i] * X[k * P + j];
        inv_out[e] = v;
    }
}

__global__ __launch_bounds__(64) void kb_small_spd(const Refs R, const int *__restrict__ sel, int src, int dst,
                                                   int do_cert, double min_eig)
{
    small_spd_body(R, sel[blockIdx.x], src, dst, do_cert, min_eig);
}

// ---- K5 kernels with a hypothesis dimension (see k5_klgrad.hip for the algebra) -------------------
// per-row constants of the point in tables (tm, ts, to): xoff, c, q, diag(S); Omega's scalars
__device__ __forceinline__ void rowconst_body(const Refs &R, int h, int tm, int ts, int to, int t)
{
    const int N = R.N, d = R.d;
    const double *__restrict__ M = tabp(R, tm, h);
    const double *__restrict__ S = tabp(R, ts, h);
    const double *__restrict__ O = tabp(R, to, h);
    if (t == 0) {
        double T = 0, D2 = 0, off2 = 0;
        for (int k = 0; k < d; k++) {
            T += O[k * d + k];
            D2 += O[k * d + k] * O[k * d + k];
            for (int l = k + 1; l < d; l++) off2 += O[k * d + l] * O[k * d + l];
        }
        OmegaConsts *oc = R.oc + h;
        oc->T = T; oc->D2 = D2; oc->off2 = off2; oc->sdd = (T * T - D2) / 2; oc->t2 = D2 + 2 * off2;
    }
    if (t < d) R.dgv[(size_t)h * 64 + t] = O[t * d + t];
    if (t >= N) return;
    const double *x = M + (size_t)t * d;
    double xo_dot = 0, x2dg = 0;
    for (int k = 0; k < d; k++) {
        double w = 0;
        for (int l = 0; l < d; l++)
            if (l != k) w += O[l < k ? l * d + k : k * d + l] * x[l];
        R.xoff[((size_t)h * N + t) * d + k] = w;
        xo_dot += x[k] * w;
        x2dg += x[k] * x[k] * O[k * d + k];
    }
    R.crow[(size_t)h * N + t] = (xo_dot + x2dg) / 2;
    R.qrow[(size_t)h * N + t] = xo_dot + x2dg;
    R.sdiag[(size_t)h * N + t] = S[(size_t)t * N + t];
}

__global__ __launch_bounds__(256) void kb_rowconst(const Refs R, const int *__restrict__ sel, int tm, int ts, int to)
{
    rowconst_body(R, sel[blockIdx.y], tm, ts, to, blockIdx.x * 256 + threadIdx.x);
}

// After the factorisations of a trial, in ONE launch of 64-thread blocks (blockIdx.y = hypothesis):
// block 0 the d x d part (certificate, factor, inverse of cov_latents), block 1 log det of the
// factor, blocks 2.. the per-row constants of the trial point.
__global__ __launch_bounds__(64) void kb_post(const Refs R, const int *__restrict__ sel, int tm, int ts, int to,
                                              int src_o, int dst_oi, int do_cert, double min_eig)
{
    const int h = sel[blockIdx.y];
    if (blockIdx.x == 0) small_spd_body(R, h, src_o, dst_oi, do_cert, min_eig);
    else if (blockIdx.x == 1) logdet_body(R, h);
    else rowconst_body(R, h, tm, ts, to, ((int)blockIdx.x - 2) * 64 + (int)threadIdx.x);
}

constexpr int UN6 = 2;

// One wave per row of the CURRENT point.  SIDE 0: user rows (also writes the per-rating cross
// entries of g_S and the sums cov_latents' gradient needs); SIDE 1: item rows.  The hypothesis'
// extra rating is entry `end` of its row.
template <int DP, int SIDE>
__device__ __forceinline__ void kl_rows_body(const Refs &R, const int *__restrict__ sel, int bx,
                                             const int64_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                             const double *__restrict__ val, double inv_sig)
{
    constexpr int G = 64 / DP;
    const int h = sel[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int row = bx * 4 + wave_id_uniform();
    const int rows = SIDE == 0 ? R.n : R.m;
    if (row >= rows) return;
    const int d = R.d, N = R.N, n = R.n;
    const double *__restrict__ M = tabp(R, T_M_CUR, h);
    const double *__restrict__ S = tabp(R, T_S_CUR, h);
    double *__restrict__ gS = tabp(R, T_S_GRAD, h);
    const double *__restrict__ xoff = R.xoff + (size_t)h * N * d;
    const double *__restrict__ crow = R.crow + (size_t)h * N;
    const double *__restrict__ sdiag = R.sdiag + (size_t)h * N;
    const OmegaConsts oc = R.oc[h];
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    const int self = SIDE == 0 ? row : n + row;
    const double xk = kok ? M[(size_t)self * d + k] : 0.;
    const double xok = kok ? xoff[(size_t)self * d + k] : 0.;
    const double dgk = kok ? R.dgv[(size_t)h * 64 + k] : 0.;
    const int64_t beg = ptr[row], end0 = ptr[row + 1];
    const int xi = R.ei[h], xj = R.ej[h];
    const bool has_extra = xi >= 0 && (SIDE == 0 ? xi == row : xj == row);
    const int xc = SIDE == 0 ? xj : xi;
    const double xv = has_extra ? R.ev[h] : 0.;
    const int64_t end = end0 + (has_extra ? 1 : 0);
    double gm = 0., wv = 0., gs_d = 0., cv = 0., s2 = 0., rs = 0., c2 = 0.;
    for (int64_t t0 = beg + g; t0 < end; t0 += (int64_t)G * UN6) {
#pragma unroll
        for (int u = 0; u < UN6; u++) {
            const int64_t t = t0 + (int64_t)u * G;
            const bool ok = t < end;
            const bool ex = ok && t == end0;     // only reachable when has_extra
            const int c = ok ? (ex ? xc : idx[t]) : 0;
            const double r = ok ? (ex ? xv : val[t]) : 0.;
            const int other = SIDE == 0 ? n + c : c;
            const double yk = (ok && kok) ? M[(size_t)other * d + k] : 0.;
            const double yok = (ok && kok) ? xoff[(size_t)other * d + k] : 0.;
            const double voth = ok ? sdiag[other] : 0.;
            const double coth = ok ? crow[other] : 0.;
            const double s = ok ? (SIDE == 0 ? S[(size_t)row * N + other] : S[(size_t)other * N + self]) : 0.;
            const double p = xk * yk;
            const double P = gsum<DP>(p);
            if (ok) {
                gm += yk * (P - p) + s * yok + voth * xok + yk * s * (oc.T - dgk) + 2. * yk * s * dgk
                      + xk * (yk * yk + voth * dgk) - r * yk;
                gs_d += coth + voth * (oc.off2 + oc.D2 / 2.);
                cv += voth;
            }
            if (SIDE == 0) {
                const double cross = gsum<DP>(xk * (yk * (oc.T + dgk) + yok));
                if (ok) {
                    wv += s * yk;
                    s2 += s * s;
                    rs += r * s;
                    if (k == 0) {
                        const double gij = inv_sig * (cross + 2. * s * (oc.sdd + oc.off2 + oc.D2) - r * oc.T);
                        gS[(size_t)row * N + other] += gij;
                        gS[(size_t)other * N + row] += gij;
                        c2 += gij * gij;
                    }
                }
            }
        }
    }
    gm = xgsum<DP>(gm);
    wv = xgsum<DP>(wv);
    gs_d = xgsum<DP>(gs_d); cv = xgsum<DP>(cv); s2 = xgsum<DP>(s2); rs = xgsum<DP>(rs);
    if (SIDE == 0) {
        c2 = xgsum<DP>(c2);          // (only the k == 0 lanes added: lane 0 ends with the row's total)
        if (lane == 0) R.c2row[(size_t)h * n + row] = c2;
    }
    const int stride = SIDE == 0 ? 2 * d + 4 : d + 2;
    double *out = (SIDE == 0 ? R.racc_u + (size_t)h * n * (2 * d + 4) : R.racc_v + (size_t)h * R.m * (d + 2)) +
                  (size_t)row * stride;
    if (g == 0 && kok) {
        out[k] = gm;
        if (SIDE == 0) out[d + k] = wv;
    }
    if (lane == 0) {
        if (SIDE == 0) {
            out[2 * d] = gs_d; out[2 * d + 1] = cv; out[2 * d + 2] = s2; out[2 * d + 3] = rs;
        } else {
            out[d] = gs_d; out[d + 1] = cv;
        }
    }
}

// user rows (blocks [0, nbu)) and item rows in one launch: neither reads what the other writes
template <int DP>
__global__ __launch_bounds__(256) void kb_kl_rows(const Refs R, const int *__restrict__ sel, int nbu,
                                                  const int64_t *__restrict__ ptr_u, const int32_t *__restrict__ idx_u,
                                                  const double *__restrict__ val_u, const int64_t *__restrict__ ptr_v,
                                                  const int32_t *__restrict__ idx_v, const double *__restrict__ val_v,
                                                  double inv_sig)
{
    if ((int)blockIdx.x < nbu) kl_rows_body<DP, 0>(R, sel, (int)blockIdx.x, ptr_u, idx_u, val_u, inv_sig);
    else kl_rows_body<DP, 1>(R, sel, (int)blockIdx.x - nbu, ptr_v, idx_v, val_v, inv_sig);
}

// g_mean rows and the diagonal of g_S from the per-row sums.  Also the inputs of the structured eigenvalue bound
// (cert_by_bound): with g_S = Rs - (d/2)(2 S^-1 - diag S^-1), Rs the sparse rating / prior part, the trial point is
// cand = (cur + 2c cur^-1) - (lr Rs + c diag cur^-1), c = lr d / 2, and the second bracket - sparse, dominated by its
// diagonal - has l_max <= lr (max_t q_t + |cross entries|_F), q_t = Rs_tt + (d/2) (S^-1)_tt.  Per block: max q_t and
// the users' share of the squared cross entries, fixed order.
__global__ __launch_bounds__(256) void kb_finish_rows(const Refs R, const int *__restrict__ sel, double inv_sig,
                                                      double sig_u, double sig_v)
{
    const int h = sel[blockIdx.y];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int d = R.d, n = R.n;
    double q = -INFINITY, c2 = 0.;
    if (t < R.N) {
        const double *__restrict__ M = tabp(R, T_M_CUR, h);
        double *__restrict__ gmean = tabp(R, T_M_GRAD, h);
        double *__restrict__ gS = tabp(R, T_S_GRAD, h);
        const double trO = R.oc[h].T;
        const bool user = t < n;
        const double *acc = user ? R.racc_u + ((size_t)h * n + t) * (2 * d + 4)
                                 : R.racc_v + ((size_t)h * R.m + (t - n)) * (d + 2);
        const double prior = user ? sig_u : sig_v;
        for (int k = 0; k < d; k++) gmean[(size_t)t * d + k] = inv_sig * acc[k] + M[(size_t)t * d + k] / prior;
        const double gs = user ? acc[2 * d] : acc[d];
        const double ent = gS[(size_t)t * R.N + t];            // -(d/2) (S^-1)_tt
        const double add = inv_sig * gs + trO / (2. * prior);
        gS[(size_t)t * R.N + t] = ent + add;
        q = add - ent;
        if (user) c2 = R.c2row[(size_t)h * n + t];
    }
    __shared__ double redq[4], redc[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q = fmax(q, __shfl_xor(q, o, 64));
    c2 = wave_sum(c2);
    if ((threadIdx.x & 63) == 0) { redq[threadIdx.x >> 6] = q; redc[threadIdx.x >> 6] = c2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *out = R.bndp + ((size_t)h * gridDim.x + blockIdx.x) * 2;
        out[0] = fmax(fmax(redq[0], redq[1]), fmax(redq[2], redq[3]));
        out[1] = (redc[0] + redc[1]) + (redc[2] + redc[3]);
        if (blockIdx.x == 0) R.hs[h].bnd_ok = 1;
    }
}

// cov_latents' gradient: one block per (k, l); fixed-order sums over the rows
__global__ __launch_bounds__(256) void kb_omega(const Refs R, const int *__restrict__ sel, double inv_sig,
                                                double sig_u, double sig_v)
{
    const int h = sel[blockIdx.y];
    const int d = R.d, n = R.n, m = R.m;
    const int k = blockIdx.x / d, l = blockIdx.x % d;
    const double *__restrict__ M = tabp(R, T_M_CUR, h);
    const double *__restrict__ O = tabp(R, T_O_CUR, h);
    const double *__restrict__ Oinv = tabp(R, T_OI_CUR, h);
    double *__restrict__ gO = tabp(R, T_O_GRAD, h);
    const double *__restrict__ racc_u = R.racc_u + (size_t)h * n * (2 * d + 4);
    const double *__restrict__ racc_v = R.racc_v + (size_t)h * m * (d + 2);
    const double *__restrict__ sdiag = R.sdiag + (size_t)h * R.N;
    const OmegaConsts oc = R.oc[h];
    __shared__ double sh[6][256];
    double g = 0, svv = 0, s2 = 0, rs = 0, aw = 0, tru = 0;
    const int su = 2 * d + 4, sv = d + 2;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double *a = M + (size_t)i * d, *acc = racc_u + (size_t)i * su;
        const double *w = acc + d;
        const double cvj = acc[2 * d + 1];
        g += w[k] * a[l] + a[k] * w[l] + cvj * a[k] * a[l];
        svv += sdiag[i] * cvj;
        s2 += acc[2 * d + 2];
        rs += acc[2 * d + 3];
        double dot = 0;
        for (int c = 0; c < d; c++) dot += a[c] * w[c];
        aw += dot;
        tru += sdiag[i];
    }
    double trv = 0;
    for (int j = threadIdx.x; j < m; j += 256) {
        const double *b = M + (size_t)(n + j) * d;
        const double cvi = racc_v[(size_t)j * sv + d + 1];
        g += cvi * b[k] * b[l];
        trv += sdiag[n + j];
    }
    double vals[6] = {g, svv, s2, rs, aw, tru};
    for (int c = 0; c < 6; c++) sh[c][threadIdx.x] = vals[c];
    __shared__ double sh_trv[256];
    sh_trv[threadIdx.x] = trv;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            for (int c = 0; c < 6; c++) sh[c][threadIdx.x] += sh[c][threadIdx.x + o];
            sh_trv[threadIdx.x] += sh_trv[threadIdx.x + o];
        }
        __syncthreads();
    }
    const double G = sh[0][0], SVV = sh[1][0], S2 = sh[2][0], RS = sh[3][0], AW = sh[4][0];
    const double TRU = sh[5][0], TRV = sh_trv[0];
    // diagonal entries also need X = sum_i M[i][k] w_i[k], added in row order by one thread: its
    // operands are fetched by the whole block, 256 rows at a time, and handed over through LDS (read
    // from global memory inside the loop this was n dependent round trips: 10 us at n = 94)
    double X = 0;
    if (k == l) {                    // block-uniform
        for (int base = 0; base < n; base += 256) {
            __syncthreads();
            const int i = base + (int)threadIdx.x;
            sh[0][threadIdx.x] = i < n ? M[(size_t)i * d + k] : 0.;
            sh[1][threadIdx.x] = i < n ? racc_u[(size_t)i * su + d + k] : 0.;
            __syncthreads();
            if (threadIdx.x == 0) {
                const int cnt = min(256, n - base);
                for (int q = 0; q < cnt; q++) X += sh[0][q] * sh[1][q];
            }
        }
    }
    if (threadIdx.x != 0) return;
    const double okl = O[k <= l ? k * d + l : l * d + k];
    double out;
    if (k != l) {
        out = inv_sig * (G + 2. * okl * (SVV + S2));
    } else {
        const double dgk = O[k * d + k];
        const double Q = G - 2. * X;
        const double diag_q = AW - X + S2 * (oc.T - dgk);
        const double diag_s = .5 * (4. * X + 4. * S2 * dgk + Q + 2. * SVV * dgk);
        out = inv_sig * (diag_q + diag_s - RS) + TRU / (2. * sig_u) + TRV / (2. * sig_v);
    }
    const double ik = Oinv[k * d + l], ikt = Oinv[l * d + k];
    out -= R.N / 2. * (ik + (k == l ? 0. : ikt));
    gO[k * d + l] = out;
}

// per-user partial of  sum_t [ E[(u.v)^2] - 2 r E[u.v] + r^2 ]  at the point in tables (tm, ts)
template <int DP>
__global__ __launch_bounds__(256) void kb_value_rows(const Refs R, const int *__restrict__ sel, int tm, int ts,
                                                     const int64_t *__restrict__ ptr,
                                                     const int32_t *__restrict__ idx,
                                                     const double *__restrict__ val)
{
    constexpr int G = 64 / DP;
    const int h = sel[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave_id_uniform();
    if (row >= R.n) return;
    const int d = R.d, N = R.N, n = R.n;
    const double *__restrict__ M = tabp(R, tm, h);
    const double *__restrict__ S = tabp(R, ts, h);
    const double *__restrict__ xoff = R.xoff + (size_t)h * N * d;
    const double *__restrict__ qrow = R.qrow + (size_t)h * N;
    const double *__restrict__ sdiag = R.sdiag + (size_t)h * N;
    const OmegaConsts oc = R.oc[h];
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    const double ak = kok ? M[(size_t)row * d + k] : 0.;
    const double oak = kok ? xoff[(size_t)row * d + k] + R.dgv[(size_t)h * 64 + k] * ak : 0.;   // (O' a)_k
    const double vi = sdiag[row], aOa = qrow[row];
    double bit = 0., un = 0.;
    if (g == 0) un = ak * ak;
    const int64_t beg = ptr[row], end0 = ptr[row + 1];
    const bool has_extra = R.ei[h] == row;
    const int64_t end = end0 + (has_extra ? 1 : 0);
    for (int64_t t = beg + g; t < end; t += G) {
        const bool ex = t == end0;
        const int j = n + (ex ? R.ej[h] : idx[t]);
        const double r = ex ? R.ev[h] : val[t];
        const double bk = kok ? M[(size_t)j * d + k] : 0.;
        const double ab = gsum<DP>(ak * bk);
        const double aOb = gsum<DP>(oak * bk);
        const double s = S[(size_t)row * N + j], vj = sdiag[j];
        const double pred = ab + s * oc.T;
        const double var = (vi * vj + s * s) * oc.t2 + vj * aOa + vi * qrow[j] + 2. * s * aOb;
        if (k == 0) bit += var + pred * pred - 2. * r * pred + r * r;
    }
    bit = wave_sum(bit);
    un = wave_sum(un);
    if (lane == 0) {
        double *part = R.part + ((size_t)h * n + row) * 3;
        part[0] = bit;
        part[1] = un;
        part[2] = vi;
    }
}

// ---- the decision of the learning-rate search, per hypothesis (mn_active_pmf.py:262-288) -----------
// mode 0: a trial point was evaluated - accept (swap the pointer tables) or reject;
// mode 1: the KL of the current point was evaluated at entry (old_kl unknown).
// forced: the flags were dealt with by the host (slow path), ld_s / ld_o are final.
__device__ __forceinline__ void decide_one(const Refs &R, int h, int mode, int forced, double sigma_sq,
                                           double sigma_u_sq)
{
    HypState &s = R.hs[h];
    if (!forced) {
        int slow = 0;
        if (mode == 0 && R.info[2 * h] != 0) slow |= SLOW_CERT_S;
        if (R.info[2 * h + 1] != 0) slow |= SLOW_FACT_S;
        if (mode == 0 && R.ores[(size_t)h * 4] != 0.) slow |= SLOW_CERT_O;
        if (R.ores[(size_t)h * 4 + 1] != 0.) slow |= SLOW_INV_O;
        s.slow = slow;
        if (slow) {
            s.status = ST_SLOW;
            return;
        }
    }
    const double *h4 = R.out4 + (size_t)h * 4;
    const double T = R.oc[h].T;
    // mn_kl_divergence, matrix_normal_exps_cy.pyx:159-213 (sic: the item trace term is dropped
    // and the item prior uses sigma_u_sq, :176, :191-197)
    double kl = -(s.ld_s * R.d + s.ld_o * R.N) / 2.;
    kl += (h4[1] + h4[2] * T) / (2. * sigma_u_sq);
    kl += (h4[3] + 0. * T) / (2. * sigma_u_sq);
    kl += h4[0] / (2. * sigma_sq);
    s.new_kl = kl;
    s.status = ST_ACTIVE;
    if (mode == 1) {
        if (s.old_kl != s.old_kl) s.old_kl = kl;
        s.cur_ld_s = s.ld_s;
        s.cur_ld_o = s.ld_o;
        s.accepted = 0;
        return;
    }
    if (kl < s.old_kl) {
        // accepted: the trial becomes the current point
        const int pairs[4][2] = {{T_M_CUR, T_M_CAND}, {T_S_CUR, T_S_CAND}, {T_O_CUR, T_O_CAND}, {T_OI_CUR, T_OI_CAND}};
        for (int p = 0; p < 4; p++) {
            double **a = R.tab + (size_t)pairs[p][0] * R.H + h, **b = R.tab + (size_t)pairs[p][1] * R.H + h;
            double *t = *a; *a = *b; *b = t;
        }
        s.lr *= 1.25;
        if (s.old_kl - kl < .005) s.status = ST_CONVERGED;
        if (s.steps < R.trace_cap) R.trace[(size_t)h * R.trace_cap + s.steps] = kl;
        s.steps++;
        s.old_kl = kl;
        s.cur_ld_s = s.ld_s;
        s.cur_ld_o = s.ld_o;
        s.need_grad = 1;
        s.inv_ok = 0;
        s.bnd_ok = 0;
        s.cur_sym = forced ? 0 : 1;      // a fast-path trial point is symmetric bit for bit (kb_trial_S)
        s.accepted = 1;
        if (s.status == ST_ACTIVE && s.max_steps >= 0 && s.steps >= s.max_steps) s.status = ST_PAUSED;
    } else {
        s.lr *= .5;
        if (s.lr < 1e-10) s.status = ST_CONVERGED;
        s.need_grad = 0;
        s.accepted = 0;
    }
}

// the four totals of a point's KL pieces, then the decision on it (one block per hypothesis):
// out4[h] = {sum bit, ||M_users||^2, tr(S_users), ||M_items||^2}
__global__ __launch_bounds__(256) void kb_value_finish(const Refs R, const int *__restrict__ sel, int tm, int mode,
                                                       int forced, double sigma_sq, double sigma_u_sq)
{
    const int h = sel[blockIdx.x];
    const double *__restrict__ part = R.part + (size_t)h * R.n * 3;
    const double *__restrict__ M = tabp(R, tm, h) + (size_t)R.n * R.d;
    __shared__ double sh[4][256];
    double a = 0, b = 0, c = 0, e = 0;
    for (int t = threadIdx.x; t < R.n; t += 256) {
        a += part[(size_t)t * 3]; b += part[(size_t)t * 3 + 1]; c += part[(size_t)t * 3 + 2];
    }
    const int md = R.m * R.d;
    for (int base = 0; base < md; base += 256 * 8) {       // 8 loads in flight per thread
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = M[min(base + q * 256 + (int)threadIdx.x, md - 1)];
#pragma unroll
        for (int q = 0; q < 8; q++) e += base + q * 256 + (int)threadIdx.x < md ? v[q] * v[q] : 0.;
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b; sh[2][threadIdx.x] = c; sh[3][threadIdx.x] = e;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int z = 0; z < 4; z++) sh[z][threadIdx.x] += sh[z][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int z = 0; z < 4; z++) R.out4[(size_t)h * 4 + z] = sh[z][0];
        __threadfence_block();
        decide_one(R, h, mode, forced, sigma_sq, sigma_u_sq);
    }
}

// ---- S^-1 from the factor ---------------------------------------------------------------------------
// W = L^-1 by block elimination.  With L = L_0 L_1 ... (L_k = identity except block column k),
// L^-1 = ... L_1^-1 L_0^-1; applying L_k^-1 multiplies block row k by Linv_kk and subtracts
// L_ik (row k) from the rows below.  Launch k (k = 0 .. nb-2) computes, for every tile (i > k,
// j <= k):   W_ij <- W_ij - L_ik W_kj   (W_ij = 0 when j = k, W_kk = Linv_kk),
// and the tiles of row k + 1 are multiplied by Linv_{k+1,k+1} on the way out, so that row k + 1
// is final when launch k + 1 reads it.  Diagonal tiles of W are never stored: they are Linv_kk.
// a 64 x 64 operand tile: a window of a row-major matrix (zero beyond its edge), or a dense Linv
struct TileSrc {
    const double *p;
    int ld, r0, c0, re, ce;
};
__device__ __forceinline__ TileSrc src_mat(const double *A, int N, int r0, int c0) { return {A, N, r0, c0, N, N}; }
__device__ __forceinline__ TileSrc src_dense(const double *B) { return {B, CH_NB, 0, 0, CH_NB, CH_NB}; }
// requested (all 16 elements of the thread in flight) ... and, later, placed in LDS
__device__ __forceinline__ void tile_request(TileRegs<64, 64> &t, const TileSrc &s)
{
    t.load(s.p, s.ld, s.r0, s.c0, s.re, s.ce);
}
__device__ __forceinline__ void tile_place(const TileRegs<64, 64> &t, const TileSrc &s, double (*T)[TP])
{
#pragma unroll
    for (int q = 0; q < t.PER; q++) {
        const int r = t.row(q), c = t.col(q);
        T[r][c] = (s.r0 + r < s.re && s.c0 + c < s.ce) ? t.v[q] : 0.;
    }
}

__global__ __launch_bounds__(256) void kb_trinv_step(const Refs R, const int *__restrict__ sel, int k)
{
    __shared__ double A[64][TP], B[64][TP];
    int bx, by;
    xcd_tile_job(bx, by);                 // the tiles of one hypothesis on one XCD: they share L_ik and W_kj
    const int h = sel[by];
    const int N = R.N, nb = R.nbk;
    const int i = k + 1 + bx / (k + 1), j = bx % (k + 1);
    const double *__restrict__ L = tabp(R, T_S_F, h);
    double *__restrict__ W = tabp(R, T_S_W, h);
    const double *__restrict__ linv = R.linvF + (size_t)h * nb * CH_NB * CH_NB;
    // everything the block reads is requested before any of it is used
    const TileSrc sa = src_mat(L, N, i * 64, k * 64);                                     // L_ik
    const TileSrc sb = j == k ? src_dense(linv + (size_t)k * CH_NB * CH_NB)               // W_kk = Linv_kk
                              : src_mat(W, N, k * 64, j * 64);                            // W_kj (final)
    const TileSrc sc = src_dense(linv + (size_t)(k + 1) * CH_NB * CH_NB);
    TileRegs<64, 64> ta, tb, tc;
    tile_request(ta, sa);
    tile_request(tb, sb);
    f64x4 old[2][2];
    TILE_FOR_EACH(u, v, r) {
        const int rr = min(i * 64 + tile_row(u, r), N - 1), cc = min(j * 64 + tile_col(v), N - 1);
        old[u][v][r] = gmem(W)[(size_t)rr * N + cc];
    }
    if (i == k + 1) tile_request(tc, sc);
    tile_place(ta, sa, A);
    tile_place(tb, sb, B);
    __syncthreads();
    f64x4 acc[2][2] = {};
    const int rmax = N - i * 64;                  // a short last block: accumulators outside the matrix are skipped
    tile_mfma<MM_NN>(acc, A, B, 64, rmax);
    __syncthreads();
    // out = (j < k ? W_ij : 0) - acc
    TILE_FOR_EACH(u, v, r) {
        const int rr = i * 64 + tile_row(u, r), cc = j * 64 + tile_col(v);
        acc[u][v][r] = ((j < k && rr < N && cc < N) ? old[u][v][r] : 0.) - acc[u][v][r];
    }
    if (i == k + 1) {
        // row k + 1 becomes final: multiply by Linv_{k+1,k+1} from the left
        TILE_FOR_EACH(u, v, r) B[tile_row(u, r)][tile_col(v)] = acc[u][v][r];
        tile_place(tc, sc, A);
        __syncthreads();
        TILE_FOR_EACH(u, v, r) acc[u][v][r] = 0.;
        // (inner dimension = the rows of block i: rows of B beyond them are zero)
        tile_mfma<MM_NN>(acc, A, B, min(64, (rmax + 3) & ~3), rmax);
    }
    TILE_FOR_EACH(u, v, r) {
        const int rr = i * 64 + tile_row(u, r), cc = j * 64 + tile_col(v);
        if (rr < N && cc < N) gmem(W)[(size_t)rr * N + cc] = acc[u][v][r];
    }
}

// g_S <- -(d/2) (2 - delta) S^-1,  S^-1 = W' W: tile (i >= j) = sum_{k >= i} W_ki' W_kj, both
// triangles written.  (The sparse rating terms are added on top by kb_kl_rows / kb_finish_rows.)
__global__ __launch_bounds__(256) void kb_lauum(const Refs R, const int *__restrict__ sel, double half_mult)
{
    __shared__ double A[64][TP], B[64][TP];
    int bx, by;
    xcd_tile_job(bx, by);                 // the tiles of one hypothesis on one XCD: every W tile is read by ~9 of them
    const int h = sel[by];
    const int N = R.N, nb = R.nbk;
    int ti, tj;
    tri_tile(bx, ti, tj);       // ti >= tj
    const double *__restrict__ W = tabp(R, T_S_W, h);
    double *__restrict__ G = tabp(R, T_S_GRAD, h);
    const double *__restrict__ linv = R.linvF + (size_t)h * nb * CH_NB * CH_NB;
    f64x4 acc[2][2] = {};
    auto src_a = [&](int k) { return k == ti ? src_dense(linv + (size_t)k * CH_NB * CH_NB) : src_mat(W, N, k * 64, ti * 64); };
    auto src_b = [&](int k) { return k == tj ? src_dense(linv + (size_t)k * CH_NB * CH_NB) : src_mat(W, N, k * 64, tj * 64); };
    // the operands of step k + 1 are requested while step k is multiplied
    TileRegs<64, 64> ta, tb;
    tile_request(ta, src_a(ti));
    tile_request(tb, src_b(ti));
    for (int k = ti; k < nb; k++) {
        __syncthreads();
        tile_place(ta, src_a(k), A);
        tile_place(tb, src_b(k), B);
        if (k + 1 < nb) {
            tile_request(ta, src_a(k + 1));
            tile_request(tb, src_b(k + 1));
        }
        __syncthreads();
        // rows of W beyond N are zero (a padded Linv: unit vectors whose outputs are dropped): the inner
        // dimension of a short last block stops there, and so do the kept rows / columns of the output
        tile_mfma<MM_TN>(acc, A, B, min(64, (N - k * 64 + 3) & ~3), N - ti * 64, N - tj * 64);
    }
    // the identity padding of a short last block must not leak into the sums: rows beyond N of a
    // padded Linv are unit vectors whose columns are beyond N too, and those outputs are dropped
    double inv2 = 0.;        // this tile's share of |S^-1|_F^2 (cert_by_bound)
    TILE_FOR_EACH(u, v, r) {
        const int rr = ti * 64 + tile_row(u, r), cc = tj * 64 + tile_col(v);
        if (rr < N && cc < N && (ti != tj || cc <= rr)) {
            const double val = -half_mult * acc[u][v][r] * (rr == cc ? 1. : 2.);
            gmem(G)[(size_t)rr * N + cc] = val;
            if (rr != cc) gmem(G)[(size_t)cc * N + rr] = val;
            inv2 += acc[u][v][r] * acc[u][v][r] * (rr == cc ? 1. : 2.);
        }
    }
    {
        __shared__ double red[4];
        inv2 = wave_sum(inv2);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = inv2;
        __syncthreads();
        if (threadIdx.x == 0) {
            R.invp[(size_t)h * gridDim.x + bx] = (red[0] + red[1]) + (red[2] + red[3]);
            if (bx == 0) R.hs[h].inv_ok = 1;
        }
    }
}

// slow hypotheses (no factor): g_S holds S^-1 from the LU route; apply the same scaling
__global__ __launch_bounds__(256) void kb_entropy_scale(double *__restrict__ g, int N, double half_mult)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)N * N) return;
    const int i = (int)(t / N), j = (int)(t % N);
    g[t] = -half_mult * g[t] * (i == j ? 1. : 2.);
}

// ---- projection onto the cone {eigenvalues >= min_eig} for the hypotheses whose certificate failed ----
// project_psd (mn_active_pmf.py:42-67): eigh, and - only if the smallest eigenvalue is below
// min_eig - rebuild V max(L, min_eig) V'.  The eigendecompositions of all such hypotheses of a
// search iteration are ONE rocsolver_dsyevd_batched call on copies (in W); the rebuild is ours.
__global__ void kb_ptrs(const Refs R, const int *__restrict__ sel, int nsel, int which)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y < nsel) R.ptrs[y] = tabp(R, which, sel[y]);
}

// cand <- V max(L, min_eig) V'  where W holds the eigenvectors as rocSOLVER leaves them (column-
// major: row c of the row-major buffer is eigenvector c) and evals (ascending) the spectrum of
// slot y; untouched when evals[0] >= min_eig.  Lower tiles computed, both triangles written (the
// reference symmetrises the rebuilt matrix; this one is symmetric by construction).
__global__ __launch_bounds__(256) void kb_rebuild(const Refs R, const int *__restrict__ sel, double min_eig)
{
    __shared__ double A[64][TP], B[64][TP];
    const int y = blockIdx.y, h = sel[y];
    const int N = R.N, nb = R.nbk;
    const double *__restrict__ w = R.evals + (size_t)y * N;
    if (!(w[0] < min_eig)) return;
    int ti, tj;
    tri_tile((int)blockIdx.x, ti, tj);       // ti >= tj
    const double *__restrict__ V = tabp(R, T_S_W, h);
    double *__restrict__ C = tabp(R, T_S_CAND, h);
    f64x4 acc[2][2] = {};
    for (int k = 0; k < nb; k++) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            const int ev = k * 64 + r;
            const double m = ev < N ? (w[ev] > min_eig ? w[ev] : min_eig) : 0.;
            const double vi = (ev < N && ti * 64 + c < N) ? V[(size_t)ev * N + ti * 64 + c] : 0.;
            const double vj = (ev < N && tj * 64 + c < N) ? V[(size_t)ev * N + tj * 64 + c] : 0.;
            A[r][c] = m * vi;
            B[r][c] = vj;
        }
        __syncthreads();
        tile_mfma<MM_TN>(acc, A, B, 64);
    }
    TILE_FOR_EACH(u, v, r) {
        const int rr = ti * 64 + tile_row(u, r), cc = tj * 64 + tile_col(v);
        if (rr < N && cc < N && (ti != tj || cc <= rr)) {
            C[(size_t)rr * N + cc] = acc[u][v][r];
            if (rr != cc) C[(size_t)cc * N + rr] = acc[u][v][r];
        }
    }
}

// ---- the same projection through the FEW eigenpairs below min_eig ------------------------------------------
// A trial matrix that fails the certificate has, in practice, one eigenvalue below min_eig (two or three now
// and then), well separated from the rest of the spectrum: V max(L, min_eig) V' is then
// A + sum_{l_k < min_eig} (min_eig - l_k) v_k v_k', and the few pairs (l_k, v_k) cost a fraction of the full
// eigendecomposition (12.5 ms of some 3000 dependent small kernels in rocSOLVER at 519 x 519):
//   bracket   Cholesky of A - s I for a ladder of shifts s below min_eig, all in ONE batched factorisation
//             (the jobs of a launch are independent): the largest shift that factors is a certified lower
//             bound of the smallest eigenvalue, within a factor 2 of its distance to min_eig;
//   iterate   inverse iteration x <- L^-T L^-1 x with the explicit triangular inverse of that factor (two
//             matrix-vector launches per step), until the residual |A x - l x| of the Rayleigh pair is at
//             rounding level;
//   deflate   A <- A + (t - l) v v' moves the pair to the top of the spectrum; the certificate of the deflated
//             matrix (one more Cholesky) says whether another eigenvalue is below min_eig - then once more.
// Every sum has a fixed order and a hypothesis' arithmetic does not depend on the others of the pass, so the
// result is the same alone or in any batch.  Anything unusual (no convergence, more than LP_KMAX pairs, a
// smallest eigenvalue far below the ladder) falls back to the full eigendecomposition.
constexpr int LP_NS = 14;      // jobs of a bracket: the certificate, 12 shifts, one below every eigenvalue
constexpr int LP_KMAX = 6;     // eigenpairs found one after the other
constexpr int LP_RS = 4;       // row splits of the transposed product
constexpr int LP_CAP = 48;     // hypotheses per chunk (lists travel as kernel arguments)

struct LpRefs {
    double *Wk;       // [cap][LP_NS][N * N] shifted copies, factored in place
    double *Linv;     // [cap][N * N]        strictly-lower blocks of the chosen factor's inverse
    double *linv;     // [cap * LP_NS][nbk][64 * 64] inverses of the factors' diagonal blocks
    double *zp;       // [cap][LP_RS][N]     partial sums of the iterate
    double *u;        // [cap][N]
    double *xh;       // [cap][N]            the normalised iterate
    double *wv;       // [cap][N]            A xh
    double *vecs;     // [cap][LP_KMAX][N]
    double *lam;      // [cap][LP_KMAX]
    double *out;      // [cap][2]            Rayleigh quotient, residual norm
    unsigned long long *norm;   // [cap][2]  bit patterns of |A|_inf and max |a_ii| (non-negative doubles order like integers)
    double *slb;      // [cap]               a lower bound of the trial matrix' smallest eigenvalue (-inf: none)
    ChJob *jobs;      // [cap * LP_NS]
    int *info;        // [cap * LP_NS]
    int N, nbk;
};

struct LpList {
    int n;
    int y[LP_CAP];    // slot in the chunk
    int h[LP_CAP];    // hypothesis
    int a[LP_CAP];    // per-kernel meaning: chosen job / pair index / number of pairs
};

__device__ __forceinline__ double block_sum_256(double v, double *sh)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// |A|_inf and max |a_ii| of the working copy (W) of each listed hypothesis: a wave per row
__global__ __launch_bounds__(256) void lp_norms(const Refs R, const LpRefs L, const LpList S, int use_bound)
{
    const int q = blockIdx.y, y = S.y[q], N = L.N;
    const double *__restrict__ A = tabp(R, T_S_W, S.h[q]);
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        // l_min(trial) >= 1 / |cur^-1|_F - |lr g|_F (cert_by_bound's quantities; the trial is this hypothesis' last):
        // rungs of the ladder below that are known to factor, and only the first of them is needed
        const int h = S.h[q];
        const int ninv = R.nbk * (R.nbk + 1) / 2, nt32 = (N + 31) / 32, nstep = nt32 * (nt32 + 1) / 2;
        double a = 0., b = 0.;
        for (int i = lane; i < ninv; i += 64) a += R.invp[(size_t)h * ninv + i];
        for (int i = lane; i < nstep; i += 64) b += R.stepp[(size_t)h * nstep + i];
        a = wave_sum(a);
        b = wave_sum(b);
        double lb = (use_bound && R.hs[h].inv_ok && a > 0.) ? .99 / sqrt(a) - 1.01 * sqrt(b) : -INFINITY;
        // ... or the structured bound of cert_by_bound (entropy term by AM-GM, rating part by its diagonal + Frobenius
        // norm), which here may well be negative: it is within a factor ~2 of the smallest eigenvalue where the plain
        // one is far below it, and prunes more rungs
        if (use_bound && R.hs[h].bnd_ok && R.hs[h].lr > 0.) {
            const int nfr = (N + 255) / 256;
            double qmax = -INFINITY, c2 = 0.;
            for (int i = lane; i < nfr; i += 64) {
                qmax = fmax(qmax, R.bndp[((size_t)h * nfr + i) * 2]);
                c2 += R.bndp[((size_t)h * nfr + i) * 2 + 1];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) qmax = fmax(qmax, __shfl_xor(qmax, o, 64));
            c2 = wave_sum(c2);
            const double lr = R.hs[h].lr, c = lr * R.d / 2.;
            const double lb1 = .99 * 2. * sqrt(2. * c) - 1.01 * lr * (fmax(qmax, 0.) + sqrt(c2));
            if (lb1 == lb1) lb = fmax(lb, lb1);
        }
        if (lane == 0) L.slb[y] = lb;
    }
    if (row >= N) return;
    double s = 0;
    for (int c0 = 0; c0 < N; c0 += 64 * 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = A[(size_t)row * N + min(c0 + k * 64 + lane, N - 1)];
#pragma unroll
        for (int k = 0; k < 8; k++) s += c0 + k * 64 + lane < N ? fabs(v[k]) : 0.;
    }
    s = wave_sum(s);
    if (lane == 0) {
        atomicMax(L.norm + 2 * y, (unsigned long long)__double_as_longlong(s));
        atomicMax(L.norm + 2 * y + 1, (unsigned long long)__double_as_longlong(fabs(A[(size_t)row * N + row])));
    }
}

// the shift of job t of a bracket
__device__ __forceinline__ double lp_shift_of(const LpRefs &L, int y, int t, double diag, double min_eig, int local)
{
    if (local) {
        // ladder below the current Rayleigh pair: an eigenvalue lies within the residual norm of the quotient
        const double rho = L.out[2 * y], res = L.out[2 * y + 1];
        return rho - (res * exp2((double)(t - 1)) * (1. + 1e-6) + 4e-16 * fabs(rho));
    }
    if (t == 0) return min_eig + 1e-10 * fmax(fabs(diag), fabs(min_eig));       // the certificate of the fast path
    const double ninf = __longlong_as_double((long long)L.norm[2 * y]);
    const double scale = __longlong_as_double((long long)L.norm[2 * y + 1]);
    if (t == LP_NS - 1) return -(2.02 * ninf + fabs(min_eig));                   // below every eigenvalue (deflated or not)
    return min_eig - scale * exp2((double)(t - (LP_NS - 1)));                    // scale * 2^-12 ... scale / 2
}

// jobs t0 .. t0 + nt - 1 of each listed hypothesis: lower triangle of W - shift I into the job's slot, the job record
__global__ __launch_bounds__(256) void lp_shift(const Refs R, const LpRefs L, const LpList S, int t0, int nt, double min_eig,
                                                int local)
{
    const int q = blockIdx.y / nt, t = t0 + blockIdx.y % nt, y = S.y[q], N = L.N;
    const int job = y * LP_NS + t;
    const double *__restrict__ A = tabp(R, T_S_W, S.h[q]);
    double *__restrict__ D = L.Wk + (size_t)job * N * N;
    // a rung whose predecessor already lies below the eigenvalue bound adds nothing: the predecessor factors
    // (ladder of the first kind only; the shifts of a ladder grow with t)
    const bool idle = !local && t >= 2 && lp_shift_of(L, y, t - 1, 0., min_eig, 0) < L.slb[y];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ChJob j;
        j.A = idle ? nullptr : D;             // (an idle rung's blocks leave before they request anything)
        j.info = L.info + job;
        j.linv = L.linv + (size_t)job * L.nbk * CH_NB * CH_NB;
        j.linv_keep = 1;
        j.shifted = 0;
        j.src = nullptr;
        j.eig = 0.;
        L.info[job] = idle ? -1 : 0;          // non-zero: the factorisation kernels pass over the job, the host over the rung
        L.jobs[q * nt + (t - t0)] = j;
    }
    if (idle) return;
    const size_t nn = (size_t)N * N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nn; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / N), j = (int)(e % N);
        if (j > i) continue;
        const double v = A[e];
        D[e] = i == j ? v - lp_shift_of(L, y, t, v, min_eig, local) : v;
    }
}

// start vector: the same fixed pattern for every hypothesis (any vector with a component along the wanted
// eigenvector does)
__global__ void lp_init(const LpRefs L, const LpList S)
{
    const int y = S.y[blockIdx.y], N = L.N;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const unsigned int x = (unsigned int)i * 2654435761u + 12345u;
    double *z = L.zp + (size_t)y * LP_RS * N;
    z[i] = (double)((x >> 8) & 0xffff) / 65536. - .5 + 1. / 131072.;
    for (int s = 1; s < LP_RS; s++) z[(size_t)s * N + i] = 0.;
}

// x = sum of the partial sums, normalised, into LDS (every block of a launch computes the same values)
__device__ __forceinline__ void lp_load_x(const LpRefs &L, int y, double *xs, double *sh)
{
    const int N = L.N;
    const double *__restrict__ z = L.zp + (size_t)y * LP_RS * N;
    double ss = 0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const double v = (z[i] + z[N + i]) + (z[2 * (size_t)N + i] + z[3 * (size_t)N + i]);
        xs[i] = v;
        ss += v * v;
    }
    const double inv = 1. / sqrt(block_sum_256(ss, sh));
    for (int i = threadIdx.x; i < N; i += 256) xs[i] *= inv;
    __syncthreads();
}

// u = L^-1 xh: 16 rows per block, 4 per wave; the strictly-lower blocks come from Linv, the diagonal block
// from the factor's kept inverses
__global__ __launch_bounds__(256) void lp_fwd(const LpRefs L, const LpList S)
{
    extern __shared__ double xs[];
    __shared__ double sh[4];
    const int q = blockIdx.y, y = S.y[q], N = L.N;
    lp_load_x(L, y, xs, sh);
    const double *__restrict__ Li = L.Linv + (size_t)y * N * N;
    const double *__restrict__ Dg = L.linv + (size_t)(y * LP_NS + S.a[q]) * L.nbk * CH_NB * CH_NB;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * 16 + w * 4;
    const int bi = (blockIdx.x * 16) / 64;           // the 16 rows of a block lie in one block row
    double acc[4] = {0., 0., 0., 0.};
    for (int cb = 0; cb < bi; cb++) {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = Li[(size_t)min(r0 + k, N - 1) * N + cb * 64 + lane];
        const double x = xs[cb * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] += v[k] * x;
    }
    {
        const double *__restrict__ D = Dg + (size_t)bi * CH_NB * CH_NB;
        const double x = bi * 64 + lane < N ? xs[bi * 64 + lane] : 0.;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rl = min(r0 + k, N - 1) - bi * 64;
            acc[k] += lane <= rl ? D[rl * CH_NB + lane] * x : 0.;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double s = wave_sum(acc[k]);
        if (lane == 0 && r0 + k < N) L.u[(size_t)y * N + r0 + k] = s;
    }
}

// z = L^-T u, as LP_RS partial sums: block (column block j, split s) takes the rows j*64 + s*4 + w + 16 t
__global__ __launch_bounds__(256) void lp_bwd(const LpRefs L, const LpList S)
{
    __shared__ double red[4][64];
    const int q = blockIdx.y, y = S.y[q], N = L.N;
    const int j = blockIdx.x / LP_RS, s = blockIdx.x % LP_RS;
    const double *__restrict__ Li = L.Linv + (size_t)y * N * N;
    const double *__restrict__ D = L.linv + ((size_t)(y * LP_NS + S.a[q]) * L.nbk + j) * CH_NB * CH_NB;
    const double *__restrict__ u = L.u + (size_t)y * N;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = j * 64 + lane, cc = min(c, N - 1);
    double acc = 0.;
    for (int r = j * 64 + s * 4 + w; r < N; r += 16 * 4) {
        double v[4], ur[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rr = min(r + 16 * k, N - 1);
            const int rl = rr - j * 64;
            const double *p = rl < 64 ? D + rl * CH_NB + lane : Li + (size_t)rr * N + cc;
            v[k] = *p;
            ur[k] = u[rr];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rr = r + 16 * k;
            acc += (rr < N && c <= rr) ? v[k] * ur[k] : 0.;
        }
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < N)
        L.zp[((size_t)y * LP_RS + s) * N + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// wv = W xh (the hypothesis' working matrix, full rows); block 0 also leaves xh
__global__ __launch_bounds__(256) void lp_resid(const Refs R, const LpRefs L, const LpList S)
{
    extern __shared__ double xs[];
    __shared__ double sh[4];
    const int q = blockIdx.y, y = S.y[q], N = L.N;
    lp_load_x(L, y, xs, sh);
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < N; i += 256) L.xh[(size_t)y * N + i] = xs[i];
    const double *__restrict__ A = tabp(R, T_S_W, S.h[q]);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * 16 + w * 4;
    double acc[4] = {0., 0., 0., 0.};
    for (int c0 = 0; c0 < N; c0 += 64) {
        double v[4];
        const int c = min(c0 + lane, N - 1);
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = A[(size_t)min(r0 + k, N - 1) * N + c];
        const double x = c0 + lane < N ? xs[c] : 0.;
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] += v[k] * x;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double s = wave_sum(acc[k]);
        if (lane == 0 && r0 + k < N) L.wv[(size_t)y * N + r0 + k] = s;
    }
}

// out = {l = xh' A xh, |A xh - l xh|}
__global__ __launch_bounds__(256) void lp_final(const LpRefs L, const LpList S)
{
    __shared__ double sh[4];
    const int y = S.y[blockIdx.x], N = L.N;
    const double *__restrict__ x = L.xh + (size_t)y * N, *__restrict__ w = L.wv + (size_t)y * N;
    double s = 0;
    for (int i = threadIdx.x; i < N; i += 256) s += x[i] * w[i];
    const double lam = block_sum_256(s, sh);
    double rr = 0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const double e = w[i] - lam * x[i];
        rr += e * e;
    }
    const double res = sqrt(block_sum_256(rr, sh));
    if (threadIdx.x == 0) {
        L.out[2 * y] = lam;
        L.out[2 * y + 1] = res;
    }
}

// keep the pair (slot a) and move it to the top of the working matrix' spectrum: W += (t - l) xh xh'
__global__ __launch_bounds__(256) void lp_commit(const Refs R, const LpRefs L, const LpList S)
{
    const int q = blockIdx.y, y = S.y[q], N = L.N, k = S.a[q];
    const double *__restrict__ x = L.xh + (size_t)y * N;
    const double lam = L.out[2 * y];
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < N; i += 256) L.vecs[((size_t)y * LP_KMAX + k) * N + i] = x[i];
        if (threadIdx.x == 0) L.lam[y * LP_KMAX + k] = lam;
    }
    const double coef = __longlong_as_double((long long)L.norm[2 * y]) - lam;
    double *__restrict__ A = tabp(R, T_S_W, S.h[q]);
    const size_t nn = (size_t)N * N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nn; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / N), j = (int)(e % N);
        A[e] += coef * (x[i] * x[j]);
    }
}

// cand += sum_k (min_eig - l_k) v_k v_k' over the a pairs of each listed hypothesis
__global__ __launch_bounds__(256) void lp_apply(const Refs R, const LpRefs L, const LpList S, double min_eig)
{
    const int q = blockIdx.y, y = S.y[q], N = L.N, nk = S.a[q];
    double *__restrict__ C = tabp(R, T_S_CAND, S.h[q]);
    const double *__restrict__ V = L.vecs + (size_t)y * LP_KMAX * N;
    const size_t nn = (size_t)N * N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nn; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / N), j = (int)(e % N);
        double add = 0.;
        for (int k = 0; k < nk; k++) add += (min_eig - L.lam[y * LP_KMAX + k]) * (V[(size_t)k * N + i] * V[(size_t)k * N + j]);
        C[e] += add;
    }
}

// ---- K2 inputs from the resident states ----------------------------------------------------------------
// contiguous (B, ...) arrays of the current points of the selected hypotheses: mean (B, N, d),
// cross (B, n, ld) = cov_useritems[:n, n:], diag (B, N), omega (B, d, d); T = double or float
template <typename T>
__global__ __launch_bounds__(256) void kb_extract(const Refs R, const int *__restrict__ sel, int ld,
                                                  T *__restrict__ mean, T *__restrict__ cross,
                                                  T *__restrict__ diag, T *__restrict__ omega)
{
    const int y = blockIdx.y;
    const int h = sel ? sel[y] : y;
    const int N = R.N, n = R.n, m = R.m, d = R.d;
    const double *__restrict__ M = tabp(R, T_M_CUR, h);
    const double *__restrict__ S = tabp(R, T_S_CUR, h);
    const double *__restrict__ O = tabp(R, T_O_CUR, h);
    const size_t n_cross = (size_t)n * ld, n_mean = (size_t)N * d, n_o = (size_t)d * d;
    const size_t total = n_cross + n_mean + N + n_o;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        if (t < n_cross) {
            const int i = (int)(t / ld), j = (int)(t % ld);
            cross[(size_t)y * n_cross + t] = j < m ? (T)S[(size_t)i * N + n + j] : (T)0;
        } else if (t < n_cross + n_mean) {
            const size_t e = t - n_cross;
            mean[(size_t)y * n_mean + e] = (T)M[e];
        } else if (t < n_cross + n_mean + N) {
            const size_t e = t - n_cross - n_mean;
            diag[(size_t)y * N + e] = (T)S[e * N + e];
        } else {
            const size_t e = t - n_cross - n_mean - N;
            omega[(size_t)y * n_o + e] = (T)O[e];
        }
    }
}

// arbitrary pool of candidates of ONE hypothesis, straight from its resident state (float64)
__global__ __launch_bounds__(256) void kb_pool(const Refs R, int h, const int32_t *__restrict__ pi,
                                               const int32_t *__restrict__ pj, int64_t P,
                                               double *__restrict__ pmean, double *__restrict__ var)
{
    extern __shared__ double sO[];  // d*d mirrored omega
    const int d = R.d, n = R.n, m = R.m, N = R.N;
    const double *__restrict__ M = tabp(R, T_M_CUR, h);
    const double *__restrict__ S = tabp(R, T_S_CUR, h);
    const double *__restrict__ O = tabp(R, T_O_CUR, h);
    for (int t = threadIdx.x; t < d * d; t += blockDim.x) {
        const int k = t / d, l = t % d;
        sO[t] = O[k <= l ? k * d + l : l * d + k];
    }
    __syncthreads();
    double t2 = 0, tr = 0;
    for (int k = 0; k < d; k++) {
        tr += sO[k * d + k];
        for (int l = 0; l < d; l++) t2 += sO[k * d + l] * sO[k * d + l];
    }
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < P; p += stride) {
        const int i = pi[p], j = pj[p];
        if (i < 0 || i >= n || j < 0 || j >= m) {
            if (pmean) pmean[p] = __builtin_nan("");
            if (var) var[p] = __builtin_nan("");
            continue;
        }
        const double *a = M + (size_t)i * d, *b = M + (size_t)(n + j) * d;
        const double sij = S[(size_t)i * N + n + j], sii = S[(size_t)i * N + i], sjj = S[(size_t)(n + j) * N + n + j];
        double aOa = 0, bOb = 0, aOb = 0, ab = 0;
        for (int k = 0; k < d; k++) {
            double oa = 0, ob = 0;
            for (int l = 0; l < d; l++) { oa += sO[k * d + l] * a[l]; ob += sO[k * d + l] * b[l]; }
            aOa += a[k] * oa; bOb += b[k] * ob; aOb += b[k] * oa; ab += a[k] * b[k];
        }
        if (var) var[p] = (sii * sjj + sij * sij) * t2 + sjj * aOa + sii * bOb + 2 * sij * aOb;
        if (pmean) pmean[p] = ab + sij * tr;
    }
}

int dp6_for(int d)
{
    int dp = 4;
    while (dp < d) dp <<= 1;
    return dp;
}

#define DISPATCH_DP6(dp, ...)                                                  \
    switch (dp) {                                                              \
    case 4: { constexpr int DP = 4; __VA_ARGS__; } break;                      \
    case 8: { constexpr int DP = 8; __VA_ARGS__; } break;                      \
    case 16: { constexpr int DP = 16; __VA_ARGS__; } break;                    \
    case 32: { constexpr int DP = 32; __VA_ARGS__; } break;                    \
    case 64: { constexpr int DP = 64; __VA_ARGS__; } break;                    \
    default: apmf_set_error("latent_d too large"); return APMF_EINVAL;         \
    }

} // namespace

// ---- the batch object -------------------------------------------------------------------------------------
constexpr int MAX_LANES = 4;
constexpr int SEL_SLOTS = 3 + 2 * (MAX_LANES + 1);   // 0 active, 1 gradient, 2 scratch, then two per eigen set

// What one eigen-projection pass works with: index 0 is the synchronous set (main stream, the
// context's rocBLAS handle), 1 .. nlanes the slow lanes - a stream, an event, a handle and a worker
// thread each, so that several passes can be in flight beside the fast path.
struct EigSet {
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    void *handle = nullptr;            // rocblas_handle of the lane (owned by the context)
    double **ptrs = nullptr;           // [H]
    double *evals = nullptr, *evE = nullptr;   // [H][N]
    int *evinfo = nullptr;             // [H]
    ChJob *jobs = nullptr;             // [H] job list of the re-factorisations
    int slot0 = 0;                     // selection-list slots slot0, slot0 + 1
    // projection through the lowest eigenpairs: workspace (low_ensure), chunk size
    LpRefs lp;
    int lp_cap = 0;
    double **lp_tab = nullptr;         // [T_COUNT][lp_cap * LP_NS] job table for the triangular inverse
    int *lp_sel = nullptr;             // [lp_cap] device selection list
    int *lp_host = nullptr;            // pinned: info words, selection, norms, Rayleigh pairs
    int n_full = 0;                    // matrices of the last pass that went through the full eigendecomposition
};

struct apmf_mn_batch {
    apmf_ctx *ctx;
    const apmf_ratings *r;
    Refs R;                       // device pointers
    void *blocks[128];            // every device allocation, for destroy
    int nblocks;
    HypState *hs_host;            // pinned mirror
    int *sel_host;                // pinned, SEL_SLOTS * H ints
    int *sel_dev;                 // SEL_SLOTS * H ints
    // the slow lanes: eigen-projections of parked hypotheses run on streams of their own, beside the
    // fast path of the others
    EigSet eig[MAX_LANES + 1];
    int nlanes;
    std::vector<char> fvalid;     // F holds the factor of the current cov_useritems
    std::vector<char> fvalid_cand;
    std::vector<char> oivalid;    // OI_CUR holds the inverse of the current cov_latents
    std::vector<double *> tab_host;
    // APMF_BATCH_STATS=1: counters of the last fit, printed to stderr when it returns
    long st_iters, st_trials, st_grads, st_slow[4], st_slow_grad;
    double st_slow_s, st_lane_s;
    long st_lane_passes, st_lane_hyps;
    long st_proj_low, st_proj_full;      // eigen-projections finished through the lowest eigenpairs / the full decomposition
    long st_cert_skipped;                // trial certificates settled by the eigenvalue bound instead of a factorisation
};

namespace {

int balloc(apmf_mn_batch *b, size_t bytes, void **out)
{
    APMF_TRY(apmf_malloc(b->ctx, bytes ? bytes : 8, out));
    b->blocks[b->nblocks++] = *out;
    return APMF_OK;
}

int upload_sel(apmf_mn_batch *b, int slot, const std::vector<int> &v, const int **d_out)
{
    int *hp = b->sel_host + (size_t)slot * b->R.H, *dp = b->sel_dev + (size_t)slot * b->R.H;
    // the pinned buffer is reused every iteration: the previous copy has completed (every
    // iteration ends with a synchronising read of the status records)
    memcpy(hp, v.data(), v.size() * sizeof(int));
    APMF_CHECK_HIP(hipMemcpyAsync(dp, hp, v.size() * sizeof(int), hipMemcpyHostToDevice, b->ctx->stream));
    *d_out = dp;
    return APMF_OK;
}

int read_status(apmf_mn_batch *b)
{
    APMF_CHECK_HIP(hipMemcpyAsync(b->hs_host, b->R.hs, (size_t)b->R.H * sizeof(HypState), hipMemcpyDeviceToHost,
                                  b->ctx->stream));
    APMF_CHECK_HIP(hipStreamSynchronize(b->ctx->stream));
    return APMF_OK;
}

int read_tables(apmf_mn_batch *b)
{
    b->tab_host.resize((size_t)T_COUNT * b->R.H);
    return apmf_d2h(b->ctx, b->tab_host.data(), b->R.tab, b->tab_host.size() * sizeof(double *));
}

inline dim3 g2(size_t count, int ny, int cap = 1024)
{
    size_t bx = (count + 255) / 256;
    if (bx > (size_t)cap) bx = cap;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)ny);
}

// KL pieces of the point in (tm, ts, to) for the selected hypotheses: row constants, per-user
// sums, the four totals
int launch_value(apmf_mn_batch *b, const int *sel, int nsel, int tm, int ts, int to, int mode, int forced,
                 double sigma_sq, double sigma_u_sq, bool with_rowconst = true)
{
    const Refs &R = b->R;
    hipStream_t st = b->ctx->stream;
    if (with_rowconst)
        APMF_LAUNCH(kb_rowconst, dim3((R.N + 255) / 256, nsel), dim3(256), 0, st, R, sel, tm, ts, to);
    DISPATCH_DP6(dp6_for(R.d), APMF_LAUNCH((kb_value_rows<DP>), dim3((R.n + 3) / 4, nsel), dim3(256), 0, st,
                                                 R, sel, tm, ts, b->r->ptr[0], b->r->idx[0], b->r->val[0]));
    APMF_LAUNCH(kb_value_finish, dim3(nsel), dim3(256), 0, st, R, sel, tm, mode, forced, sigma_sq, sigma_u_sq);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

size_t small_smem(int d) { return 2 * (size_t)d * (d + 1) * sizeof(double); }

// factor the matrices already placed in F (and W when mode == 2) of the selected hypotheses
int launch_factor(apmf_mn_batch *b, const int *sel, int nsel, int mode, ChJob *jobs = nullptr)
{
    const Refs &R = b->R;
    if (!jobs) jobs = R.jobs;
    APMF_LAUNCH(kb_make_jobs, dim3((nsel + 63) / 64), dim3(64), 0, b->ctx->stream, R, sel, nsel, mode, jobs);
    APMF_CHECK_LAUNCH();
    ChJobs J;
    memset(&J, 0, sizeof(J));
    J.list = jobs;
    APMF_TRY(chol_lower_jobs(b->ctx, J, mode == 2 ? 2 * nsel : nsel, R.N));
    APMF_LAUNCH(kb_logdet, dim3(nsel), dim3(64), 0, b->ctx->stream, R, sel);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

int write_ld(apmf_mn_batch *b, int h, const double *ld_s, const double *ld_o)
{
    char *base = (char *)(b->R.hs + h);
    if (ld_s) APMF_TRY(apmf_h2d(b->ctx, base + offsetof(HypState, ld_s), ld_s, sizeof(double)));
    if (ld_o) APMF_TRY(apmf_h2d(b->ctx, base + offsetof(HypState, ld_o), ld_o, sizeof(double)));
    return APMF_OK;
}

// Hypotheses left the fast path while their point (cand when trial, else cur) was evaluated:
// finish the projection / factorisation.  cov_useritems projections of all of them are batched;
// what is rarer still (a d x d projection, a factorisation that fails on a projected matrix) goes
// through the single-matrix routines of dense.hip, one hypothesis at a time.
// S part, enqueued on the CURRENT stream of the context (the aux stream when called for the lane) with
// no host synchronisation: eigen-projection of the hypotheses whose certificate failed, then the
// factorisation of every listed hypothesis' (projected) matrix.  slot0 / slot0 + 1: selection-list
// slots, jobs: job-list buffer - private to the caller's stream.
// Three parts, so that the middle one - the call into rocSOLVER, some 3000 small launches and 4 ms
// of HOST time per pass - can run on a thread of its own while the search loop keeps feeding the main
// stream:  pre (copies, pointer list, handle bound to the stream),  solve (the eigendecompositions),
// post (rebuild of the projected matrices, factorisation of every listed hypothesis' matrix).
struct SlowS {
    std::vector<int> proj, refac;
    void *handle = nullptr;
    // projection through the lowest eigenpairs (the default; APMF_BATCH_EIG=full: always the full decomposition)
    bool low = false;
    hipStream_t stream = nullptr;
    EigSet *set = nullptr;
    int *dsel = nullptr;               // the set's first selection-list slot
    double min_eig = 0.;
    std::vector<int> full;             // hypotheses that go through the full eigendecomposition
};

// the batch's device references with the eigen buffers of one set
Refs set_refs(const apmf_mn_batch *b, const EigSet &e)
{
    Refs R = b->R;
    R.ptrs = e.ptrs;
    R.evals = e.evals;
    R.evE = e.evE;
    R.evinfo = e.evinfo;
    return R;
}

// ---- host side of the projection through the few lowest eigenpairs (kernels lp_* above) -------------------
// workspace of a set, allocated by the search thread on first use
int low_ensure(apmf_mn_batch *b, EigSet &e)
{
    if (e.lp_cap) return APMF_OK;
    const Refs &R = b->R;
    const size_t nn = (size_t)R.N * R.N, blk = (size_t)CH_NB * CH_NB;
    int cap = (int)std::min<size_t>(LP_CAP, std::max<size_t>(1, (size_t)1536e6 / (LP_NS * nn * sizeof(double))));
    cap = std::min(cap, R.H);
    const size_t nj = (size_t)cap * LP_NS;
    LpRefs L;
    memset(&L, 0, sizeof(L));
    L.N = R.N;
    L.nbk = R.nbk;
    int rc = APMF_OK;
#define LA(ptr, bytes)                                                                          \
    do {                                                                                        \
        if (rc == APMF_OK && b->nblocks >= (int)(sizeof(b->blocks) / sizeof(b->blocks[0]))) rc = APMF_ENOMEM; \
        if (rc == APMF_OK) rc = balloc(b, (bytes), (void **)&(ptr));                            \
    } while (0)
    LA(L.Wk, nj * nn * sizeof(double));
    LA(L.Linv, (size_t)cap * nn * sizeof(double));
    LA(L.linv, nj * R.nbk * blk * sizeof(double));
    LA(L.zp, (size_t)cap * (LP_RS + 3 + LP_KMAX) * R.N * sizeof(double));
    LA(L.lam, (size_t)cap * (LP_KMAX + 2 + 2 + 1) * sizeof(double));
    LA(L.jobs, nj * sizeof(ChJob));
    LA(L.info, (nj + cap) * sizeof(int));
    LA(e.lp_tab, (size_t)T_COUNT * nj * sizeof(double *));
#undef LA
    if (rc == APMF_OK && hipHostMalloc((void **)&e.lp_host, (nj + cap) * sizeof(int) + 4 * (size_t)cap * sizeof(double),
                                       hipHostMallocDefault) != hipSuccess)
        rc = APMF_ENOMEM;
    if (rc != APMF_OK) {
        apmf_set_error("out of memory for the eigenpair workspace (N=%d)", R.N);
        return rc;
    }
    L.u = L.zp + (size_t)cap * LP_RS * R.N;
    L.xh = L.u + (size_t)cap * R.N;
    L.wv = L.xh + (size_t)cap * R.N;
    L.vecs = L.wv + (size_t)cap * R.N;
    L.out = L.lam + (size_t)cap * LP_KMAX;
    L.norm = (unsigned long long *)(L.out + 2 * (size_t)cap);
    L.slb = (double *)(L.norm + 2 * (size_t)cap);
    e.lp_sel = L.info + nj;
    // the table the triangular-inverse kernel reads: "hypothesis" = job; F = the job's factor, W = its slot's Linv
    std::vector<double *> tab((size_t)T_COUNT * nj, nullptr);
    for (size_t j = 0; j < nj; j++) {
        tab[(size_t)T_S_F * nj + j] = L.Wk + j * nn;
        tab[(size_t)T_S_W * nj + j] = L.Linv + (j / LP_NS) * nn;
    }
    APMF_TRY(apmf_h2d(b->ctx, e.lp_tab, tab.data(), tab.size() * sizeof(double *)));
    e.lp = L;
    e.lp_cap = cap;
    return APMF_OK;
}

// Projection of the listed hypotheses' trial matrices (cand; W holds a copy) through their eigenpairs below
// min_eig; every launch goes to `st`, and the call returns with the stream drained.  Hypotheses it cannot
// finish are appended to `fallback` with cand untouched.  Thread-safe with respect to the search thread: it
// only touches the set's own buffers and the listed hypotheses' cand / W.
int low_project(const Refs &R, const EigSet &e, hipStream_t st, const std::vector<int> &hyps, double min_eig, bool trial,
                std::vector<int> &fallback)
{
    const LpRefs &L = e.lp;
    const int N = R.N, cap = e.lp_cap, nj = cap * LP_NS;
    int *info_h = e.lp_host, *sel_h = info_h + nj;
    double *out_h = (double *)(sel_h + cap);            // (nj + cap is even: LP_NS is)
    unsigned long long *norm_h = (unsigned long long *)(out_h + 2 * (size_t)cap);
    static const bool debug = getenv("APMF_DEBUG_LOW") != nullptr;
    Refs R2 = R;                       // the triangular inverse of the chosen factors, through the job table
    R2.tab = e.lp_tab;
    R2.H = nj;
    R2.linvF = L.linv;
    const size_t xs_bytes = (size_t)N * sizeof(double);
    const dim3 g_rows((N + 15) / 16), g_mat((unsigned)std::min<size_t>(((size_t)N * N + 255) / 256, 1024));
    auto sync_read = [&](void *dst, const void *src, size_t bytes) -> int {
        APMF_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
        APMF_CHECK_HIP(hipStreamSynchronize(st));
        return APMF_OK;
    };
    for (size_t c0 = 0; c0 < hyps.size(); c0 += cap) {
        const int nc = (int)std::min<size_t>(cap, hyps.size() - c0);
        std::vector<int> todo(nc), npairs(nc, 0), rung(nc, 0);
        std::vector<char> failed(nc, 0);
        for (int y = 0; y < nc; y++) todo[y] = y;
        auto list_of = [&](const std::vector<int> &ys) {
            LpList S;
            memset(&S, 0, sizeof(S));
            S.n = (int)ys.size();
            for (int q = 0; q < S.n; q++) {
                S.y[q] = ys[q];
                S.h[q] = hyps[c0 + ys[q]];
            }
            return S;
        };
        APMF_CHECK_HIP(hipMemsetAsync(L.norm, 0, 2 * (size_t)cap * sizeof(unsigned long long), st));
        {
            const LpList S = list_of(todo);
            APMF_LAUNCH(lp_norms, dim3((N + 3) / 4, nc), dim3(256), 0, st, R, L, S, trial ? 1 : 0);
        }
        for (int round = 0; !todo.empty(); round++) {
            if (round > 0) {
                // has the deflated matrix anything left below min_eig?  its certificate, alone
                const LpList S = list_of(todo);
                APMF_LAUNCH(lp_shift, dim3(g_mat.x, S.n), dim3(256), 0, st, R, L, S, 0, 1, min_eig, 0);
                APMF_CHECK_LAUNCH();
                ChJobs J;
                memset(&J, 0, sizeof(J));
                J.list = L.jobs;
                APMF_TRY(chol_lower_jobs_on(st, J, S.n, N));
                APMF_TRY(sync_read(info_h, L.info, (size_t)nj * sizeof(int)));
                std::vector<int> rest;
                for (int y : todo)
                    if (info_h[y * LP_NS] != 0) rest.push_back(y);
                todo.swap(rest);
                if (todo.empty()) break;
            }
            if (round >= LP_KMAX) {
                for (int y : todo) failed[y] = 1;
                break;
            }
            // bracket: the ladder of shifts, one batched factorisation
            LpList S = list_of(todo);
            APMF_LAUNCH(lp_shift, dim3(g_mat.x, S.n * (LP_NS - 1)), dim3(256), 0, st, R, L, S, 1, LP_NS - 1, min_eig, 0);
            APMF_CHECK_LAUNCH();
            {
                ChJobs J;
                memset(&J, 0, sizeof(J));
                J.list = L.jobs;
                APMF_TRY(chol_lower_jobs_on(st, J, S.n * (LP_NS - 1), N));
            }
            if (round == 0)
                APMF_CHECK_HIP(hipMemcpyAsync(norm_h, L.norm, 2 * (size_t)cap * sizeof(unsigned long long),
                                              hipMemcpyDeviceToHost, st));
            APMF_TRY(sync_read(info_h, L.info, (size_t)nj * sizeof(int)));
            {
                std::vector<int> ok;
                for (int y : todo) {
                    int t = 1;
                    while (t < LP_NS && info_h[y * LP_NS + t] != 0) t++;
                    // (the last rung is below every eigenvalue; one that sits there converges too slowly to bother)
                    if (t >= LP_NS - 1) { failed[y] = 1; continue; }
                    sel_h[ok.size()] = y * LP_NS + t;
                    rung[y] = t;
                    ok.push_back(y);
                }
                todo.swap(ok);
                if (todo.empty()) break;
                S = list_of(todo);
                for (int q = 0; q < S.n; q++) S.a[q] = rung[S.y[q]];
            }
            APMF_CHECK_HIP(hipMemcpyAsync(e.lp_sel, sel_h, (size_t)S.n * sizeof(int), hipMemcpyHostToDevice, st));
            for (int k = 0; k + 1 < R.nbk; k++)
                APMF_LAUNCH(kb_trinv_step, dim3((R.nbk - k - 1) * (k + 1), S.n), dim3(256), 0, st, R2, e.lp_sel, k);
            APMF_LAUNCH(lp_init, dim3((N + 255) / 256, S.n), dim3(256), 0, st, L, S);
            APMF_CHECK_LAUNCH();
            // inverse iteration, the Rayleigh pair and its residual looked at every few steps
            std::vector<int> run = todo;
            std::vector<double> prev(nc, 0.);
            std::vector<char> conv(nc, 0);
            std::vector<int> nladder(nc, 0);
            for (int y : run) memcpy(&prev[y], norm_h + 2 * y, sizeof(double));      // |x| = 1: the first residual is below |A|
            for (int chk = 0; chk < 48 && !run.empty(); chk++) {
                LpList T = list_of(run);
                for (int q = 0; q < T.n; q++) T.a[q] = rung[T.y[q]];
                const int its = chk == 0 ? 12 : 6;
                for (int it = 0; it < its; it++) {
                    APMF_LAUNCH(lp_fwd, dim3(g_rows.x, T.n), dim3(256), xs_bytes, st, L, T);
                    APMF_LAUNCH(lp_bwd, dim3(R.nbk * LP_RS, T.n), dim3(256), 0, st, L, T);
                }
                APMF_LAUNCH(lp_resid, dim3(g_rows.x, T.n), dim3(256), xs_bytes, st, R, L, T);
                APMF_LAUNCH(lp_final, dim3(T.n), dim3(256), 0, st, L, T);
                APMF_CHECK_LAUNCH();
                APMF_TRY(sync_read(out_h, L.out, 2 * (size_t)cap * sizeof(double)));
                std::vector<int> more, closer;
                for (int y : run) {
                    const double res = out_h[2 * y + 1];
                    double ninf;
                    memcpy(&ninf, norm_h + 2 * y, sizeof(double));       // |A|_inf of the undeflated matrix
                    if (debug) fprintf(stderr, "[low] h %d round %d check %d: lambda %.17g res %.3e\n", hyps[c0 + y], round, chk,
                                       out_h[2 * y], res);
                    if (!(res == res)) { failed[y] = 1; continue; }
                    const double target = 1e-15 * ninf;
                    const bool good = res <= target || (res <= 1e-12 * ninf && res > .5 * prev[y]);
                    if (good) {
                        conv[y] = 1;
                    } else {
                        more.push_back(y);
                        // steps still needed at the rate just seen; a shift closer to the eigenvalue (a factorisation:
                        // some 40 steps' worth) pays when the next eigenvalue is not much further from the shift
                        const double gain = log(prev[y] / res) / its;
                        const double need = gain > 0 ? log(res / target) / gain : 1e9;
                        if (need > 40. && nladder[y] < 4) closer.push_back(y);
                    }
                    prev[y] = res;
                }
                run.swap(more);
                if (!closer.empty()) {
                    // ladder of shifts below the Rayleigh pair, one batched factorisation, the closest rung that factors
                    LpList C = list_of(closer);
                    APMF_LAUNCH(lp_shift, dim3(g_mat.x, C.n * (LP_NS - 2)), dim3(256), 0, st, R, L, C, 1, LP_NS - 2, min_eig, 1);
                    APMF_CHECK_LAUNCH();
                    ChJobs J;
                    memset(&J, 0, sizeof(J));
                    J.list = L.jobs;
                    APMF_TRY(chol_lower_jobs_on(st, J, C.n * (LP_NS - 2), N));
                    APMF_TRY(sync_read(info_h, L.info, (size_t)nj * sizeof(int)));
                    std::vector<int> ok;
                    for (int y : closer) {
                        nladder[y]++;
                        int t = 1;
                        while (t < LP_NS - 1 && info_h[y * LP_NS + t] != 0) t++;
                        if (t >= LP_NS - 1) {            // (the slot of the factor in use has been overwritten)
                            failed[y] = 1;
                            run.erase(std::remove(run.begin(), run.end(), y), run.end());
                            continue;
                        }
                        sel_h[ok.size()] = y * LP_NS + t;
                        rung[y] = t;
                        ok.push_back(y);
                    }
                    if (!ok.empty()) {
                        APMF_CHECK_HIP(hipMemcpyAsync(e.lp_sel, sel_h, ok.size() * sizeof(int), hipMemcpyHostToDevice, st));
                        for (int k = 0; k + 1 < R.nbk; k++)
                            APMF_LAUNCH(kb_trinv_step, dim3((R.nbk - k - 1) * (k + 1), (int)ok.size()), dim3(256), 0, st, R2,
                                               e.lp_sel, k);
                        APMF_CHECK_LAUNCH();
                        for (int y : ok) prev[y] = 1e300;        // the rate of the old shift says nothing about the new one
                    }
                }
            }
            for (int y : run) failed[y] = 1;
            // pairs below min_eig are kept and deflated; a smallest eigenvalue that is not below ends the hypothesis
            std::vector<int> keep;
            for (int y : todo) {
                if (!conv[y]) continue;
                if (out_h[2 * y] < min_eig) keep.push_back(y);
            }
            if (!keep.empty()) {
                LpList K = list_of(keep);
                for (int q = 0; q < K.n; q++) K.a[q] = npairs[keep[q]]++;
                APMF_LAUNCH(lp_commit, dim3(g_mat.x, K.n), dim3(256), 0, st, R, L, K);
                APMF_CHECK_LAUNCH();
            }
            todo.swap(keep);
        }
        std::vector<int> app;
        for (int y = 0; y < nc; y++) {
            if (failed[y]) fallback.push_back(hyps[c0 + y]);
            else if (npairs[y] > 0) app.push_back(y);
        }
        if (!app.empty()) {
            LpList A = list_of(app);
            for (int q = 0; q < A.n; q++) A.a[q] = npairs[app[q]];
            APMF_LAUNCH(lp_apply, dim3(g_mat.x, A.n), dim3(256), 0, st, R, L, A, min_eig);
            APMF_CHECK_LAUNCH();
        }
        APMF_CHECK_HIP(hipStreamSynchronize(st));     // the next chunk reuses the buffers the host reads
    }
    return APMF_OK;
}

// handle: bound to the current stream by the caller
int slow_S_pre(apmf_mn_batch *b, EigSet &e, void *handle, const std::vector<int> &list,
               const std::vector<int> &masks, SlowS &w, double min_eig)
{
    const Refs R = set_refs(b, e);
    hipStream_t st = b->ctx->stream;
    w.proj.clear();
    w.refac.clear();
    w.full.clear();
    w.handle = handle;
    w.stream = st;
    w.set = &e;
    w.min_eig = min_eig;
    w.dsel = b->sel_dev + (size_t)e.slot0 * R.H;
    const char *mode = getenv("APMF_BATCH_EIG");      // "full": always the full eigendecomposition (comparison runs)
    w.low = !(mode && !strcmp(mode, "full")) && (size_t)R.N * sizeof(double) <= 48 * 1024;    // (the iterate lives in LDS)
    e.n_full = 0;
    for (size_t q = 0; q < list.size(); q++) {
        if (masks[q] & SLOW_CERT_S) w.proj.push_back(list[q]);
        if (masks[q] & (SLOW_CERT_S | SLOW_FACT_S)) w.refac.push_back(list[q]);
    }
    if (w.proj.empty()) return APMF_OK;
    if (w.low) APMF_TRY(low_ensure(b, e));
    const size_t nn = (size_t)R.N * R.N;
    const int np = (int)w.proj.size();
    const int *dsel;
    APMF_TRY(upload_sel(b, e.slot0, w.proj, &dsel));
    APMF_LAUNCH(kb_copy, g2(nn, np, 256), dim3(256), 0, st, R, dsel, (int)T_S_CAND, (int)T_S_W, nn);
    if (!w.low) APMF_LAUNCH(kb_ptrs, dim3((np + 63) / 64), dim3(64), 0, st, R, dsel, np, (int)T_S_W);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

// thread-safe with respect to everything but the handle and the set
rocblas_status slow_S_solve(const Refs &R, SlowS &w)
{
    if (w.proj.empty()) return rocblas_status_success;
    hipStream_t st = w.stream;
    if (w.low) {
        // (the projected matrix is always a trial point: SLOW_CERT_S is only raised by a trial's certificate)
        if (low_project(R, *w.set, st, w.proj, w.min_eig, true, w.full) != APMF_OK) return rocblas_status_internal_error;
        if (w.full.empty()) return rocblas_status_success;
        // the leftovers: a fresh copy of their trial matrices, the pointer list
        const int nf = (int)w.full.size();
        const size_t nn = (size_t)R.N * R.N;
        if (hipMemcpyAsync(w.dsel, w.full.data(), (size_t)nf * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)      // (w.full is pageable: the copy is staged before the call returns anyway)
            return rocblas_status_internal_error;
        APMF_LAUNCH(kb_copy, g2(nn, nf, 256), dim3(256), 0, st, R, w.dsel, (int)T_S_CAND, (int)T_S_W, nn);
        APMF_LAUNCH(kb_ptrs, dim3((nf + 63) / 64), dim3(64), 0, st, R, w.dsel, nf, (int)T_S_W);
        if (hipGetLastError() != hipSuccess) return rocblas_status_internal_error;
    } else {
        w.full = w.proj;
    }
    // (replaying the call from a captured hipGraph was tried: same 12.5 ms per pass - the ~3000 small
    // kernels of a 519 x 519 syevd execute one after the other either way)
    return rocsolver_dsyevd_batched((rocblas_handle)w.handle, rocblas_evect_original, rocblas_fill_lower, R.N, R.ptrs,
                                    R.N, R.evals, R.N, R.evE, R.N, R.evinfo, (int)w.full.size());
}

int slow_S_post(apmf_mn_batch *b, EigSet &e, const SlowS &w, rocblas_status rs, bool trial, double min_eig)
{
    const Refs R = set_refs(b, e);
    hipStream_t st = b->ctx->stream;
    const int tS = trial ? T_S_CAND : T_S_CUR;
    const size_t nn = (size_t)R.N * R.N;
    if (rs != rocblas_status_success) {
        apmf_set_error("eigen-projection of a batch failed: rocblas_status %d", (int)rs);
        return APMF_EHIP;
    }
    e.n_full = (int)w.full.size();
    b->st_proj_full += (long)w.full.size();
    b->st_proj_low += (long)(w.proj.size() - w.full.size());
    if (!w.full.empty()) {
        const int *dsel = b->sel_dev + (size_t)e.slot0 * R.H;      // uploaded by slow_S_pre / slow_S_solve
        APMF_LAUNCH(kb_rebuild, dim3(R.nbk * (R.nbk + 1) / 2, (int)w.full.size()), dim3(256), 0, st, R, dsel,
                           min_eig);
        APMF_CHECK_LAUNCH();
    }
    if (!w.refac.empty()) {
        const int nr = (int)w.refac.size();
        const int *dsel;
        APMF_TRY(upload_sel(b, e.slot0 + 1, w.refac, &dsel));
        APMF_LAUNCH(kb_copy, g2(nn, nr, 256), dim3(256), 0, st, R, dsel, tS, (int)T_S_F, nn);
        APMF_CHECK_LAUNCH();
        APMF_TRY(launch_factor(b, dsel, nr, 1, e.jobs));
    }
    return APMF_OK;
}

// the synchronous form: set 0, the context's handle, the current stream
int slow_S_enqueue(apmf_mn_batch *b, const std::vector<int> &list, const std::vector<int> &masks, bool trial,
                   double min_eig)
{
    SlowS w;
    void *hv = nullptr;
    APMF_TRY(apmf_dense_handle(b->ctx, &hv));      // binds the handle to the current stream
    APMF_TRY(slow_S_pre(b, b->eig[0], hv, list, masks, w, min_eig));
    const rocblas_status rs = slow_S_solve(set_refs(b, b->eig[0]), w);
    return slow_S_post(b, b->eig[0], w, rs, trial, min_eig);
}

// Host side of the same, once the enqueued work has completed: convergence of the eigensolver, LU
// fallback for a factorisation that still fails, and the (rare) d x d parts through dense.hip.
int slow_finish(apmf_mn_batch *b, const EigSet &e, const std::vector<int> &list, const std::vector<int> &masks,
                bool trial, double min_eig)
{
    const Refs R = set_refs(b, e);
    const int H = R.H, N = R.N, d = R.d;
    const int tS = trial ? T_S_CAND : T_S_CUR, tO = trial ? T_O_CAND : T_O_CUR, tOI = trial ? T_OI_CAND : T_OI_CUR;
    const int np = e.n_full;
    bool any_refac = false, any_o = false;
    for (size_t q = 0; q < list.size(); q++) {
        if (masks[q] & (SLOW_CERT_S | SLOW_FACT_S)) any_refac = true;
        if (masks[q] & (SLOW_CERT_O | SLOW_INV_O)) any_o = true;
    }
    if (np) {
        std::vector<int> info(np);
        APMF_TRY(apmf_d2h(b->ctx, info.data(), R.evinfo, (size_t)np * sizeof(int)));
        for (int y = 0; y < np; y++)
            if (info[y] != 0) {
                apmf_set_error("rocsolver_dsyevd_batched did not converge (slot %d, info=%d)", y, info[y]);
                return APMF_EHIP;
            }
    }
    bool tables = false;
    if (any_refac) {
        std::vector<int> info(2 * (size_t)H);
        APMF_TRY(apmf_d2h(b->ctx, info.data(), R.info, info.size() * sizeof(int)));
        for (size_t q = 0; q < list.size(); q++) {
            if (!(masks[q] & (SLOW_CERT_S | SLOW_FACT_S))) continue;
            const int h = list[q];
            char &fv = trial ? b->fvalid_cand[h] : b->fvalid[h];
            fv = 1;
            if (info[2 * h + 1] != 0) {
                if (!tables) { APMF_TRY(read_tables(b)); tables = true; }
                double ld = 0.;
                APMF_TRY(apmf_dense_inv_logdet(b->ctx, b->tab_host[(size_t)tS * H + h], N, nullptr, &ld));   // LU
                APMF_TRY(write_ld(b, h, &ld, nullptr));
                fv = 0;
            }
        }
    }
    if (any_o)
        for (size_t q = 0; q < list.size(); q++) {
            if (!(masks[q] & (SLOW_CERT_O | SLOW_INV_O))) continue;
            const int h = list[q];
            if (!tables) { APMF_TRY(read_tables(b)); tables = true; }
            double *O = b->tab_host[(size_t)tO * H + h], *OI = b->tab_host[(size_t)tOI * H + h];
            if (masks[q] & SLOW_CERT_O) APMF_TRY(apmf_dense_project_psd(b->ctx, O, d, min_eig, nullptr));
            double ld = 0.;
            APMF_TRY(apmf_dense_inv_logdet(b->ctx, O, d, OI, &ld));
            APMF_TRY(write_ld(b, h, nullptr, &ld));
        }
    return APMF_OK;
}

int slow_fix_all(apmf_mn_batch *b, const std::vector<int> &slow, bool trial, double min_eig)
{
    std::vector<int> masks;
    for (int h : slow) masks.push_back(b->hs_host[h].slow);
    APMF_TRY(slow_S_enqueue(b, slow, masks, trial, min_eig));
    return slow_finish(b, b->eig[0], slow, masks, trial, min_eig);
}

// after a batch of evaluations: deal with the hypotheses that left the fast path, then decide them
int handle_slow(apmf_mn_batch *b, const std::vector<int> &evaluated, bool trial, double min_eig, double sigma_sq,
                double sigma_u_sq)
{
    std::vector<int> slow;
    for (int h : evaluated)
        if (b->hs_host[h].status == ST_SLOW) slow.push_back(h);
    if (slow.empty()) return APMF_OK;
    const auto t0 = std::chrono::steady_clock::now();
    if (!trial)
        for (int h : slow)
            for (int k = 0; k < 4; k++)
                if (b->hs_host[h].slow & (1 << k)) b->st_slow[k]++;
    APMF_TRY(slow_fix_all(b, slow, trial, min_eig));
    const int *dsel;
    APMF_TRY(upload_sel(b, 2, slow, &dsel));
    const int tm = trial ? T_M_CAND : T_M_CUR, ts = trial ? T_S_CAND : T_S_CUR, to = trial ? T_O_CAND : T_O_CUR;
    APMF_TRY(launch_value(b, dsel, (int)slow.size(), tm, ts, to, trial ? 0 : 1, 1, sigma_sq, sigma_u_sq));
    const int rc = read_status(b);
    b->st_slow_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// gradient of the KL at the current point of the selected hypotheses (matrixnormal_gradient)
int launch_gradient(apmf_mn_batch *b, const std::vector<int> &grad, double sigma_sq, double sigma_u_sq,
                    double sigma_v_sq)
{
    const Refs &R = b->R;
    hipStream_t st = b->ctx->stream;
    const int ng = (int)grad.size();
    std::vector<int> fast, slow;
    for (int h : grad) (b->fvalid[h] ? fast : slow).push_back(h);
    const int *dgrad, *dfast = nullptr;
    APMF_TRY(upload_sel(b, 1, grad, &dgrad));
    const double half_mult = R.d / 2.;
    b->st_slow_grad += (long)slow.size();
    if (!slow.empty()) {
        APMF_TRY(read_tables(b));
        for (int h : slow) {
            double *S = b->tab_host[(size_t)T_S_CUR * R.H + h], *G = b->tab_host[(size_t)T_S_GRAD * R.H + h];
            APMF_TRY(apmf_dense_inv_logdet(b->ctx, S, R.N, G, nullptr));
            APMF_LAUNCH(kb_entropy_scale, dim3((unsigned)(((size_t)R.N * R.N + 255) / 256)), dim3(256), 0, st, G,
                               R.N, half_mult);
        }
        APMF_CHECK_LAUNCH();
    }
    if (!fast.empty()) {
        if (slow.empty()) dfast = dgrad;
        else APMF_TRY(upload_sel(b, 2, fast, &dfast));
        const int nf = (int)fast.size();
        for (int k = 0; k + 1 < R.nbk; k++)
            APMF_LAUNCH(kb_trinv_step, dim3((R.nbk - k - 1) * (k + 1), nf), dim3(256), 0, st, R, dfast, k);
        APMF_LAUNCH(kb_lauum, dim3(R.nbk * (R.nbk + 1) / 2, nf), dim3(256), 0, st, R, dfast, half_mult);
        APMF_CHECK_LAUNCH();
    }
    APMF_LAUNCH(kb_rowconst, dim3((R.N + 255) / 256, ng), dim3(256), 0, st, R, dgrad, (int)T_M_CUR,
                       (int)T_S_CUR, (int)T_O_CUR);
    const double inv_sig = 1. / sigma_sq;
    const int dp = dp6_for(R.d);
    const int nbu = (R.n + 3) / 4, nbv = (R.m + 3) / 4;
    DISPATCH_DP6(dp, APMF_LAUNCH((kb_kl_rows<DP>), dim3(nbu + nbv, ng), dim3(256), 0, st, R, dgrad, nbu,
                                        b->r->ptr[0], b->r->idx[0], b->r->val[0], b->r->ptr[1], b->r->idx[1],
                                        b->r->val[1], inv_sig));
    APMF_LAUNCH(kb_finish_rows, dim3((R.N + 255) / 256, ng), dim3(256), 0, st, R, dgrad, inv_sig, sigma_u_sq,
                       sigma_v_sq);
    APMF_LAUNCH(kb_omega, dim3(R.d * R.d, ng), dim3(256), 0, st, R, dgrad, inv_sig, sigma_u_sq, sigma_v_sq);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

} // namespace

extern "C" {

int apmf_mn_batch_destroy(apmf_ctx *ctx, apmf_mn_batch *b)
{
    if (!b) return APMF_OK;
    (void)ctx;
    (void)hipStreamSynchronize(b->ctx->stream);
    for (int i = 0; i < b->nblocks; i++) (void)apmf_free(b->ctx, b->blocks[i]);
    for (int l = 1; l <= MAX_LANES; l++) {
        EigSet &e = b->eig[l];
        if (e.stream) (void)hipStreamSynchronize(e.stream);
        // the lane's handle outlives this batch (it is the context's): take it off the stream that goes away
        if (e.handle) (void)rocblas_set_stream((rocblas_handle)e.handle, b->ctx->stream);
        if (e.stream) (void)hipStreamDestroy(e.stream);
        if (e.ev) (void)hipEventDestroy(e.ev);
    }
    for (int l = 0; l <= MAX_LANES; l++)
        if (b->eig[l].lp_host) (void)hipHostFree(b->eig[l].lp_host);
    if (b->hs_host) (void)hipHostFree(b->hs_host);
    if (b->sel_host) (void)hipHostFree(b->sel_host);
    delete b;
    return APMF_OK;
}

int apmf_mn_batch_create(apmf_ctx *ctx, const apmf_ratings *r, int d, int H, int64_t trace_cap,
                         apmf_mn_batch **out)
{
    APMF_REQUIRE(ctx && r && out, "NULL operand");
    *out = nullptr;
    APMF_REQUIRE(d >= 1 && d <= APMF_MAX_D, "latent_d=%d unsupported (1..%d)", d, APMF_MAX_D);
    APMF_REQUIRE(H >= 1 && H <= 32000, "H=%d out of range (1..32000)", H);
    APMF_REQUIRE(trace_cap >= 0 && trace_cap <= (1 << 20), "bad trace_cap");
    apmf_mn_batch *b = new (std::nothrow) apmf_mn_batch();
    if (!b) { apmf_set_error("out of host memory"); return APMF_ENOMEM; }
    b->ctx = ctx;
    b->r = r;
    b->nblocks = 0;
    b->hs_host = nullptr;
    b->sel_host = nullptr;
    b->nlanes = 0;
    Refs &R = b->R;
    memset(&R, 0, sizeof(R));
    R.H = H; R.n = r->n; R.m = r->m; R.d = d; R.N = r->n + r->m;
    R.nbk = (R.N + CH_NB - 1) / CH_NB;
    R.trace_cap = (int)trace_cap;
    const size_t N = R.N, nn = N * N, nd = N * d, dd = (size_t)d * d, blk = (size_t)CH_NB * CH_NB;
    double *S_store = nullptr, *M_store = nullptr, *O_store = nullptr;
    int rc = APMF_OK;
#define BA(ptr, bytes)                                                      \
    if (rc == APMF_OK) rc = balloc(b, (bytes), (void **)&(ptr));
    BA(S_store, 5 * H * nn * sizeof(double));
    BA(M_store, 3 * H * nd * sizeof(double));
    BA(O_store, 5 * H * dd * sizeof(double));
    BA(R.tab, (size_t)T_COUNT * H * sizeof(double *));
    BA(R.hs, (size_t)H * sizeof(HypState));
    BA(R.info, (size_t)2 * H * sizeof(int));
    BA(R.ores, (size_t)4 * H * sizeof(double));
    BA(R.linvF, (size_t)H * R.nbk * blk * sizeof(double));
    BA(R.linvW, (size_t)H * blk * sizeof(double));
    BA(R.jobs, (size_t)2 * H * sizeof(ChJob));
    BA(R.xoff, (size_t)H * nd * sizeof(double));
    BA(R.crow, (size_t)H * N * sizeof(double));
    BA(R.qrow, (size_t)H * N * sizeof(double));
    BA(R.sdiag, (size_t)H * N * sizeof(double));
    BA(R.dgv, (size_t)H * 64 * sizeof(double));
    BA(R.racc_u, (size_t)H * R.n * (2 * d + 4) * sizeof(double));
    BA(R.racc_v, (size_t)H * R.m * (d + 2) * sizeof(double));
    BA(R.part, (size_t)H * R.n * 3 * sizeof(double));
    BA(R.out4, (size_t)H * 4 * sizeof(double));
    BA(R.oc, (size_t)H * sizeof(OmegaConsts));
    BA(R.invp, (size_t)H * (R.nbk * (R.nbk + 1) / 2) * sizeof(double));
    BA(R.stepp, (size_t)H * (((N + 31) / 32) * ((N + 31) / 32 + 1) / 2) * sizeof(double));
    BA(R.c2row, (size_t)H * R.n * sizeof(double));
    BA(R.bndp, (size_t)H * ((N + 255) / 256) * 2 * sizeof(double));
    BA(R.skipword, sizeof(int));
    // eigen sets: the synchronous one and one per slow lane.  One lane: a pass is a couple of ms of launches
    // (measured at 94 x 425: H = 10 - 1 lane 1.11 s, 2: 1.30 s; H = 250 - 1 lane 6.6 s, 2: 7.0 s; concurrent
    // launch-heavy host threads slow one another and the main stream down).  APMF_BATCH_LANES: up to MAX_LANES.
    b->nlanes = 1;
    if (const char *e = getenv("APMF_BATCH_LANES")) b->nlanes = std::max(1, std::min(MAX_LANES, atoi(e)));
    for (int l = 0; l <= b->nlanes; l++) {
        EigSet &e = b->eig[l];
        e.slot0 = 3 + 2 * l;
        BA(e.ptrs, (size_t)H * sizeof(double *));
        BA(e.evals, (size_t)H * N * sizeof(double));
        BA(e.evE, (size_t)H * N * sizeof(double));
        BA(e.evinfo, (size_t)H * sizeof(int));
        BA(e.jobs, (size_t)H * sizeof(ChJob));
    }
    R.ptrs = b->eig[0].ptrs;
    R.evals = b->eig[0].evals;
    R.evE = b->eig[0].evE;
    R.evinfo = b->eig[0].evinfo;
    BA(R.trace, (size_t)H * (trace_cap ? trace_cap : 1) * sizeof(double));
    BA(R.ei, (size_t)H * sizeof(int));
    BA(R.ej, (size_t)H * sizeof(int));
    BA(R.ev, (size_t)H * sizeof(double));
    BA(b->sel_dev, (size_t)SEL_SLOTS * H * sizeof(int));
#undef BA
    if (rc == APMF_OK && hipHostMalloc((void **)&b->hs_host, (size_t)H * sizeof(HypState), hipHostMallocDefault) != hipSuccess)
        rc = APMF_ENOMEM;
    if (rc == APMF_OK && hipHostMalloc((void **)&b->sel_host, (size_t)SEL_SLOTS * H * sizeof(int), hipHostMallocDefault) != hipSuccess)
        rc = APMF_ENOMEM;
    for (int l = 1; l <= b->nlanes && rc == APMF_OK; l++) {
        if (hipStreamCreateWithFlags(&b->eig[l].stream, hipStreamNonBlocking) != hipSuccess) rc = APMF_EHIP;
        if (rc == APMF_OK && hipEventCreateWithFlags(&b->eig[l].ev, hipEventDisableTiming) != hipSuccess) rc = APMF_EHIP;
    }
    if (rc != APMF_OK) {
        if (rc == APMF_ENOMEM) apmf_set_error("out of memory for a batch of %d hypotheses (N=%zu)", H, N);
        apmf_mn_batch_destroy(ctx, b);
        return rc;
    }
    // canonical pointer tables
    std::vector<double *> tab((size_t)T_COUNT * H);
    for (int h = 0; h < H; h++) {
        tab[(size_t)T_M_CUR * H + h] = M_store + ((size_t)0 * H + h) * nd;
        tab[(size_t)T_M_CAND * H + h] = M_store + ((size_t)1 * H + h) * nd;
        tab[(size_t)T_M_GRAD * H + h] = M_store + ((size_t)2 * H + h) * nd;
        tab[(size_t)T_S_CUR * H + h] = S_store + ((size_t)0 * H + h) * nn;
        tab[(size_t)T_S_CAND * H + h] = S_store + ((size_t)1 * H + h) * nn;
        tab[(size_t)T_S_GRAD * H + h] = S_store + ((size_t)2 * H + h) * nn;
        tab[(size_t)T_S_F * H + h] = S_store + ((size_t)3 * H + h) * nn;
        tab[(size_t)T_S_W * H + h] = S_store + ((size_t)4 * H + h) * nn;
        tab[(size_t)T_O_CUR * H + h] = O_store + ((size_t)0 * H + h) * dd;
        tab[(size_t)T_O_CAND * H + h] = O_store + ((size_t)1 * H + h) * dd;
        tab[(size_t)T_O_GRAD * H + h] = O_store + ((size_t)2 * H + h) * dd;
        tab[(size_t)T_OI_CUR * H + h] = O_store + ((size_t)3 * H + h) * dd;
        tab[(size_t)T_OI_CAND * H + h] = O_store + ((size_t)4 * H + h) * dd;
    }
    rc = apmf_h2d(ctx, R.tab, tab.data(), tab.size() * sizeof(double *));
    const int one = 1;
    if (rc == APMF_OK) rc = apmf_h2d(ctx, R.skipword, &one, sizeof(int));
    if (rc == APMF_OK) rc = apmf_memset(ctx, R.ei, 0xFF, (size_t)H * sizeof(int));     // -1: no extra rating
    if (rc == APMF_OK) rc = apmf_memset(ctx, R.ej, 0xFF, (size_t)H * sizeof(int));
    if (rc == APMF_OK) rc = apmf_memset(ctx, R.ev, 0, (size_t)H * sizeof(double));
    if (rc != APMF_OK) {
        apmf_mn_batch_destroy(ctx, b);
        return rc;
    }
    memset(b->hs_host, 0, (size_t)H * sizeof(HypState));
    if (small_smem(d) > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)kb_small_spd, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)small_smem(d));
        (void)hipFuncSetAttribute((const void *)kb_post, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)small_smem(d));
    }
    b->fvalid.assign(H, 0);
    b->fvalid_cand.assign(H, 0);
    b->oivalid.assign(H, 0);
    *out = b;
    return APMF_OK;
}

int apmf_mn_batch_fit_stats(const apmf_mn_batch *b, int64_t *iterations, int64_t *trials, int64_t *proj_low,
                            int64_t *proj_full, int64_t *cert_by_bound, int64_t *gradients)
{
    APMF_REQUIRE(b, "NULL batch");
    if (gradients) *gradients = b->st_grads;
    if (cert_by_bound) *cert_by_bound = b->st_cert_skipped;
    if (iterations) *iterations = b->st_iters;
    if (trials) *trials = b->st_trials;
    if (proj_low) *proj_low = b->st_proj_low;
    if (proj_full) *proj_full = b->st_proj_full;
    return APMF_OK;
}

int apmf_mn_batch_launch_count(int64_t *launches)
{
    APMF_REQUIRE(launches, "NULL operand");
    *launches = (int64_t)g_k6_launches.load(std::memory_order_relaxed);
    return APMF_OK;
}

int apmf_mn_batch_size(const apmf_mn_batch *b, int *H, int *n, int *m, int *d)
{
    APMF_REQUIRE(b, "NULL batch");
    if (H) *H = b->R.H;
    if (n) *n = b->R.n;
    if (m) *m = b->R.m;
    if (d) *d = b->R.d;
    return APMF_OK;
}

int apmf_mn_batch_set_state(apmf_ctx *ctx, apmf_mn_batch *b, const double *d_mean, const double *d_cov_useritems,
                            const double *d_cov_latents, int per_hypothesis)
{
    APMF_REQUIRE(ctx && b, "NULL operand");
    const Refs &R = b->R;
    const size_t nn = (size_t)R.N * R.N, nd = (size_t)R.N * R.d, dd = (size_t)R.d * R.d;
    hipStream_t st = b->ctx->stream;
    const int p = per_hypothesis;
    if (d_mean)
        APMF_LAUNCH(kb_fill, g2(nd, R.H, 64), dim3(256), 0, st, R, (int)T_M_CUR, d_mean, (p & 1) ? nd : 0, nd);
    if (d_cov_useritems) {
        APMF_LAUNCH(kb_fill, g2(nn, R.H, 256), dim3(256), 0, st, R, (int)T_S_CUR, d_cov_useritems,
                           (p & 2) ? nn : 0, nn);
        std::fill(b->fvalid.begin(), b->fvalid.end(), 0);
    }
    if (d_cov_latents) {
        APMF_LAUNCH(kb_fill, g2(dd, R.H, 4), dim3(256), 0, st, R, (int)T_O_CUR, d_cov_latents,
                           (p & 4) ? dd : 0, dd);
        std::fill(b->oivalid.begin(), b->oivalid.end(), 0);
    }
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

int apmf_mn_batch_set_extra(apmf_ctx *ctx, apmf_mn_batch *b, const int32_t *h_i, const int32_t *h_j,
                            const double *h_v)
{
    APMF_REQUIRE(ctx && b, "NULL operand");
    const Refs &R = b->R;
    if (!h_i) {
        APMF_TRY(apmf_memset(ctx, R.ei, 0xFF, (size_t)R.H * sizeof(int)));
        APMF_TRY(apmf_memset(ctx, R.ej, 0xFF, (size_t)R.H * sizeof(int)));
        return APMF_OK;
    }
    APMF_REQUIRE(h_j && h_v, "NULL operand");
    for (int h = 0; h < R.H; h++)
        APMF_REQUIRE(h_i[h] < R.n && h_j[h] < R.m && (h_i[h] >= 0) == (h_j[h] >= 0),
                     "extra rating %d: (%d, %d) out of range", h, h_i[h], h_j[h]);
    APMF_TRY(apmf_h2d(ctx, R.ei, h_i, (size_t)R.H * sizeof(int)));
    APMF_TRY(apmf_h2d(ctx, R.ej, h_j, (size_t)R.H * sizeof(int)));
    APMF_TRY(apmf_h2d(ctx, R.ev, h_v, (size_t)R.H * sizeof(double)));
    return APMF_OK;
}

int apmf_mn_batch_get_state(apmf_ctx *ctx, apmf_mn_batch *b, int h, double *d_mean, double *d_cov_useritems,
                            double *d_cov_latents)
{
    APMF_REQUIRE(ctx && b, "NULL operand");
    const Refs &R = b->R;
    APMF_REQUIRE(h >= 0 && h < R.H, "hypothesis %d out of range", h);
    APMF_TRY(read_tables(b));
    const size_t nn = (size_t)R.N * R.N, nd = (size_t)R.N * R.d, dd = (size_t)R.d * R.d;
    if (d_mean) APMF_TRY(apmf_d2d(ctx, d_mean, b->tab_host[(size_t)T_M_CUR * R.H + h], nd * sizeof(double)));
    if (d_cov_useritems)
        APMF_TRY(apmf_d2d(ctx, d_cov_useritems, b->tab_host[(size_t)T_S_CUR * R.H + h], nn * sizeof(double)));
    if (d_cov_latents) APMF_TRY(apmf_d2d(ctx, d_cov_latents, b->tab_host[(size_t)T_O_CUR * R.H + h], dd * sizeof(double)));
    return APMF_OK;
}

// fit_normal_kls() / fit_normal() for every hypothesis of the batch (mn_active_pmf.py:236-288).
// h_lr / h_old_kl (H each) carry the search state in and out (NaN old_kl: evaluate at entry);
// max_steps accepted steps per hypothesis at most (< 0: until convergence).
int apmf_mn_batch_fit(apmf_ctx *ctx, apmf_mn_batch *b, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                      double min_eig, double *h_lr, double *h_old_kl, int64_t max_steps, double *h_kl_trace,
                      int64_t *h_steps, int *h_converged)
{
    APMF_REQUIRE(ctx && b && h_lr && h_old_kl, "NULL operand");
    APMF_REQUIRE(ctx == b->ctx, "batch belongs to another context");
    APMF_REQUIRE(sigma_sq != 0. && sigma_u_sq != 0. && sigma_v_sq != 0., "zero variance");
    const Refs &R = b->R;
    const int H = R.H;
    hipStream_t st = ctx->stream;
    for (int h = 0; h < H; h++) {
        HypState &s = b->hs_host[h];      // cur_ld_s / cur_ld_o survive from the previous call
        s.lr = h_lr[h];
        s.old_kl = h_old_kl[h];
        s.new_kl = 0.;
        s.steps = 0;
        s.max_steps = max_steps;
        s.status = ST_ACTIVE;
        s.need_grad = 1;
        s.slow = 0;
        s.accepted = 0;
        s.cert_skipped = 0;
        s.inv_ok = 0;
        s.bnd_ok = 0;
        s.cur_sym = 0;
    }
    b->st_iters = b->st_trials = b->st_grads = b->st_slow_grad = 0;
    b->st_slow[0] = b->st_slow[1] = b->st_slow[2] = b->st_slow[3] = 0;
    b->st_slow_s = b->st_lane_s = 0.;
    b->st_lane_passes = b->st_lane_hyps = 0;
    b->st_proj_low = b->st_proj_full = 0;
    b->st_cert_skipped = 0;
    const auto t_fit = std::chrono::steady_clock::now();
    // a hypothesis whose KL is known AND whose factor / inverse are in place resumes directly
    std::vector<int> init;
    for (int h = 0; h < H; h++)
        if (!b->fvalid[h] || !b->oivalid[h] || b->hs_host[h].old_kl != b->hs_host[h].old_kl) init.push_back(h);
    APMF_CHECK_HIP(hipMemcpyAsync(R.hs, b->hs_host, (size_t)H * sizeof(HypState), hipMemcpyHostToDevice, st));
    if (!init.empty()) {
        const int *dsel;
        const int ni = (int)init.size();
        APMF_TRY(upload_sel(b, 0, init, &dsel));
        APMF_LAUNCH(kb_copy, g2((size_t)R.N * R.N, ni, 256), dim3(256), 0, st, R, dsel, (int)T_S_CUR,
                           (int)T_S_F, (size_t)R.N * R.N);
        APMF_TRY(launch_factor(b, dsel, ni, 1));
        APMF_LAUNCH(kb_small_spd, dim3(ni), dim3(64), small_smem(R.d), st, R, dsel, (int)T_O_CUR, (int)T_OI_CUR,
                           0, min_eig);
        APMF_TRY(launch_value(b, dsel, ni, T_M_CUR, T_S_CUR, T_O_CUR, 1, 0, sigma_sq, sigma_u_sq));
        APMF_TRY(read_status(b));
        for (int h : init) { b->fvalid[h] = 1; b->oivalid[h] = 1; }
        APMF_TRY(handle_slow(b, init, false, min_eig, sigma_sq, sigma_u_sq));
        for (int h : init) b->oivalid[h] = 1;
    } else {
        APMF_CHECK_HIP(hipStreamSynchronize(st));
    }
    if (max_steps == 0)
        for (int h = 0; h < H; h++) b->hs_host[h].status = ST_PAUSED;
    // Hypotheses whose trial left the fast path (a projection that really clips eigenvalues; the
    // trial matrix is then indefinite) are PARKED while the others keep iterating, and their
    // eigendecompositions run as ONE batched call on a second stream - the slow lane - beside the fast
    // path: when the lane is free it takes everything that is parked.  The hypotheses are
    // independent, so the order in which they advance changes nothing.  APMF_BATCH_ASYNC=0: the
    // parked hypotheses are dealt with on the main stream instead (debugging).
    const bool use_lane = !(getenv("APMF_BATCH_ASYNC") && getenv("APMF_BATCH_ASYNC")[0] == '0');
    const bool knob_cand = !(getenv("APMF_BATCH_FROM_CAND") && getenv("APMF_BATCH_FROM_CAND")[0] == '0');
    const bool knob_sym = !(getenv("APMF_BATCH_SYM") && getenv("APMF_BATCH_SYM")[0] == '0');
    // APMF_BATCH_CERT=factor: every trial's certificate by its factorisation, never by the eigenvalue bound
    // (comparison runs: the results are the same bits either way)
    const double bound_eig = (getenv("APMF_BATCH_CERT") && !strcmp(getenv("APMF_BATCH_CERT"), "factor")) ? INFINITY : min_eig;
    std::vector<int> active, grad, parked, decided, osync;
    std::vector<char> is_parked(H, 0);
    // One pass of a lane: busy from the moment it takes the parked hypotheses; `solving` while its worker
    // thread is inside rocSOLVER - lane_post() then joins the thread and enqueues the rest of the pass
    // behind the solver's kernels, after which the lane's event tells when the pass is over.
    struct LanePass {
        std::vector<int> list, masks;
        SlowS w;
        std::thread thr;
        std::atomic<int> enq{0};
        rocblas_status rs = rocblas_status_success;
        bool busy = false, solving = false;
        std::chrono::steady_clock::time_point t0;
    };
    const int NL = b->nlanes;
    LanePass pass[MAX_LANES];
    struct Joiner {              // an early return must not leave a joinable thread behind
        LanePass *p;
        ~Joiner() { for (int l = 0; l < MAX_LANES; l++) if (p[l].thr.joinable()) p[l].thr.join(); }
    } joiner{pass};
    auto any_busy = [&]() {
        for (int l = 0; l < NL; l++) if (pass[l].busy) return true;
        return false;
    };
    auto lane_post = [&](int l) -> int {
        LanePass &P = pass[l];
        if (!P.solving) return APMF_OK;
        if (P.thr.joinable()) P.thr.join();
        P.solving = false;
        EigSet &e = b->eig[l + 1];
        hipStream_t saved = ctx->stream;
        ctx->stream = e.stream;
        int rc = slow_S_post(b, e, P.w, P.rs, true, min_eig);
        if (rc == APMF_OK && hipEventRecord(e.ev, e.stream) != hipSuccess) rc = APMF_EHIP;
        ctx->stream = saved;
        return rc;
    };
    auto finish_decided = [&](const std::vector<int> &list) {
        // F holds the factor of the trial point: the current one's only if the trial was accepted
        for (int h : list) b->fvalid[h] = b->hs_host[h].accepted ? b->fvalid_cand[h] : 0;
    };
    for (;;) {
        active.clear();
        grad.clear();
        for (int h = 0; h < H; h++)
            if (b->hs_host[h].status == ST_ACTIVE && !is_parked[h]) {
                active.push_back(h);
                if (b->hs_host[h].need_grad) grad.push_back(h);
            }
        bool returned = false;
        for (int l = 0; l < NL; l++) {
            LanePass &P = pass[l];
            if (!P.busy) continue;
            const EigSet &e = b->eig[l + 1];
            // nothing else to do: wait for the first busy lane instead of spinning
            const bool idle = active.empty() && !returned;
            if (P.solving && (P.enq.load(std::memory_order_acquire) || idle)) APMF_TRY(lane_post(l));
            bool done = !P.solving && hipEventQuery(e.ev) == hipSuccess;
            (void)hipGetLastError();             // hipErrorNotReady is not an error
            if (!done && idle) {
                APMF_CHECK_HIP(hipEventSynchronize(e.ev));
                done = true;
            }
            if (!done) continue;
            const auto t0 = std::chrono::steady_clock::now();
            APMF_CHECK_HIP(hipStreamWaitEvent(st, e.ev, 0));
            APMF_TRY(slow_finish(b, e, P.list, P.masks, true, min_eig));
            const int *dsel;
            APMF_TRY(upload_sel(b, 2, P.list, &dsel));
            APMF_TRY(launch_value(b, dsel, (int)P.list.size(), T_M_CAND, T_S_CAND, T_O_CAND, 0, 1, sigma_sq,
                                  sigma_u_sq));
            APMF_TRY(read_status(b));
            finish_decided(P.list);
            for (int h : P.list) is_parked[h] = 0;
            b->st_lane_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - P.t0).count();
            b->st_lane_passes++;
            b->st_lane_hyps += (long)P.list.size();
            P.list.clear();
            P.busy = false;
            b->st_slow_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            returned = true;
        }
        if (returned) continue;              // rebuild the active list with the returned hypotheses
        if (active.empty() && parked.empty() && !any_busy()) break;
        decided.clear();
        osync.clear();
        if (!active.empty()) {
            b->st_iters++;
            b->st_trials += (long)active.size();
            b->st_grads += (long)grad.size();
            if (!grad.empty()) {
                APMF_TRY(launch_gradient(b, grad, sigma_sq, sigma_u_sq, sigma_v_sq));
            }
            const int na = (int)active.size();
            const int *dsel;
            APMF_TRY(upload_sel(b, 0, active, &dsel));
            const int nt32 = (R.N + 31) / 32;
            // many jobs: the factorisations go block column by block column and read the trial point where
            // kb_trial_S wrote it (ChJob::src) - no copies into F / W
            // (APMF_BATCH_FROM_CAND=0 / APMF_BATCH_SYM=0: the copies are made / the mirrored tiles are read
            // regardless - comparison runs, the results are the same bits)
            const int from_cand = (knob_cand && chol_route_by_column(2 * na)) ? 1 : 0;
            APMF_LAUNCH(kb_trial_S, dim3(nt32 * (nt32 + 1) / 2, na), dim3(256), 0, st, R, dsel, min_eig, !from_cand,
                               knob_sym ? 1 : 0);
            APMF_LAUNCH(kb_trial_small, dim3((unsigned)(((size_t)R.N * R.d + (size_t)R.d * R.d + 255) / 256), na),
                               dim3(256), 0, st, R, dsel, R.jobs, nt32 * (nt32 + 1) / 2, bound_eig, from_cand, min_eig);
            APMF_CHECK_LAUNCH();
            {
                ChJobs J;
                memset(&J, 0, sizeof(J));
                J.list = R.jobs;
                APMF_TRY(chol_lower_jobs(ctx, J, 2 * na, R.N));
            }
            APMF_LAUNCH(kb_post, dim3(2 + (R.N + 63) / 64, na), dim3(64), small_smem(R.d), st, R, dsel,
                               (int)T_M_CAND, (int)T_S_CAND, (int)T_O_CAND, (int)T_O_CAND, (int)T_OI_CAND, 1, min_eig);
            APMF_TRY(launch_value(b, dsel, na, T_M_CAND, T_S_CAND, T_O_CAND, 0, 0, sigma_sq, sigma_u_sq, false));
            APMF_TRY(read_status(b));
            for (int h : active) {
                b->fvalid_cand[h] = 1;
                if (b->hs_host[h].status != ST_SLOW) {
                    decided.push_back(h);
                    continue;
                }
                for (int k = 0; k < 4; k++)
                    if (b->hs_host[h].slow & (1 << k)) b->st_slow[k]++;
                if (b->hs_host[h].slow & (SLOW_CERT_S | SLOW_FACT_S)) {
                    parked.push_back(h);
                    is_parked[h] = 1;
                } else {
                    osync.push_back(h);          // only the d x d part left the fast path: rare, synchronous
                }
            }
        }
        if (!osync.empty()) {
            APMF_TRY(handle_slow(b, osync, true, min_eig, sigma_sq, sigma_u_sq));
            for (int h : osync) decided.push_back(h);
        }
        int free_lane = -1;
        for (int l = 0; l < NL && free_lane < 0; l++) if (!pass[l].busy) free_lane = l;
        if (!parked.empty() && (free_lane >= 0 || !use_lane)) {
            const auto t0 = std::chrono::steady_clock::now();
            if (use_lane) {
                LanePass &P = pass[free_lane];
                EigSet &e = b->eig[free_lane + 1];
                P.list.swap(parked);
                parked.clear();
                P.masks.clear();
                for (int h : P.list) P.masks.push_back(b->hs_host[h].slow);
                APMF_TRY(apmf_dense_lane_handle(ctx, free_lane, e.stream, &e.handle));     // the context's, kept
                // the main stream is idle here (read_status synchronised it): the lane may start at once
                hipStream_t saved = ctx->stream;
                ctx->stream = e.stream;
                int rc = slow_S_pre(b, e, e.handle, P.list, P.masks, P.w, min_eig);
                ctx->stream = saved;
                if (rc != APMF_OK) return rc;
                P.enq.store(0, std::memory_order_relaxed);
                P.solving = true;
                P.busy = true;
                P.t0 = std::chrono::steady_clock::now();
                if (P.w.proj.empty()) {
                    P.rs = rocblas_status_success;
                    P.enq.store(1, std::memory_order_release);
                    APMF_TRY(lane_post(free_lane));
                } else {
                    const int device = ctx->device;
                    const Refs RL = set_refs(b, e);
                    LanePass *pp = &P;
                    try {
                        P.thr = std::thread([pp, RL, device]() {
                            (void)hipSetDevice(device);
                            pp->rs = slow_S_solve(RL, pp->w);
                            pp->enq.store(1, std::memory_order_release);
                        });
                    } catch (...) {              // no thread to be had: issue the call from here
                        P.rs = slow_S_solve(RL, P.w);
                        P.enq.store(1, std::memory_order_release);
                    }
                }
            } else {
                APMF_TRY(handle_slow(b, parked, true, min_eig, sigma_sq, sigma_u_sq));
                for (int h : parked) {
                    decided.push_back(h);
                    is_parked[h] = 0;
                }
                parked.clear();
            }
            b->st_slow_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        finish_decided(decided);
    }
    for (int h = 0; h < H; h++) b->st_cert_skipped += b->hs_host[h].cert_skipped;
    if (const char *e = getenv("APMF_BATCH_STATS"))
        if (e[0] == '1')
            fprintf(stderr,
                    "[apmf batch] H=%d N=%d: %ld iterations, %ld trials, %ld gradients (%ld without factor); slow "
                    "exits: projection S %ld, factor S %ld, projection O %ld, inverse O %ld, %.3f s; lane: %ld passes, %ld "
                    "hypotheses, busy %.3f s; projections: %ld by eigenpairs, %ld full; %ld certificates by the eigenvalue "
                    "bound; total %.3f s\n",
                    H, R.N, b->st_iters, b->st_trials, b->st_grads, b->st_slow_grad, b->st_slow[0], b->st_slow[1],
                    b->st_slow[2], b->st_slow[3], b->st_slow_s, b->st_lane_passes, b->st_lane_hyps, b->st_lane_s,
                    b->st_proj_low, b->st_proj_full, b->st_cert_skipped,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_fit).count());
    for (int h = 0; h < H; h++) {
        const HypState &s = b->hs_host[h];
        h_lr[h] = s.lr;
        h_old_kl[h] = s.old_kl;
        if (h_steps) h_steps[h] = s.steps;
        if (h_converged) h_converged[h] = s.status == ST_CONVERGED ? 1 : 0;
    }
    if (h_kl_trace && R.trace_cap > 0)
        APMF_TRY(apmf_d2h(ctx, h_kl_trace, R.trace, (size_t)H * R.trace_cap * sizeof(double)));
    return APMF_OK;
}

// KL(approximation || model) of the current point of every hypothesis (no step is taken)
int apmf_mn_batch_kl(apmf_ctx *ctx, apmf_mn_batch *b, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                     double *h_kl)
{
    APMF_REQUIRE(ctx && b && h_kl, "NULL operand");
    const int H = b->R.H;
    std::vector<double> lr(H, 0.), kl(H, NAN);
    APMF_TRY(apmf_mn_batch_fit(ctx, b, sigma_sq, sigma_u_sq, sigma_v_sq, 0., lr.data(), kl.data(), 0, nullptr, nullptr,
                               nullptr));
    memcpy(h_kl, kl.data(), (size_t)H * sizeof(double));
    return APMF_OK;
}

// gradient of the KL at the current point of hypothesis h into caller buffers (matrixnormal_gradient)
int apmf_mn_batch_gradient(apmf_ctx *ctx, apmf_mn_batch *b, int h, double sigma_sq, double sigma_u_sq,
                           double sigma_v_sq, double *d_g_mean, double *d_g_cov_useritems, double *d_g_cov_latents)
{
    APMF_REQUIRE(ctx && b, "NULL operand");
    const Refs &R = b->R;
    APMF_REQUIRE(h >= 0 && h < R.H, "hypothesis %d out of range", h);
    // make sure the factor and the d x d inverse of the current point are in place
    std::vector<double> kl(R.H);
    APMF_TRY(apmf_mn_batch_kl(ctx, b, sigma_sq, sigma_u_sq, sigma_v_sq, kl.data()));
    std::vector<int> one(1, h);
    APMF_TRY(launch_gradient(b, one, sigma_sq, sigma_u_sq, sigma_v_sq));
    APMF_TRY(read_tables(b));
    const size_t nn = (size_t)R.N * R.N, nd = (size_t)R.N * R.d, dd = (size_t)R.d * R.d;
    if (d_g_mean) APMF_TRY(apmf_d2d(ctx, d_g_mean, b->tab_host[(size_t)T_M_GRAD * R.H + h], nd * sizeof(double)));
    if (d_g_cov_useritems)
        APMF_TRY(apmf_d2d(ctx, d_g_cov_useritems, b->tab_host[(size_t)T_S_GRAD * R.H + h], nn * sizeof(double)));
    if (d_g_cov_latents)
        APMF_TRY(apmf_d2d(ctx, d_g_cov_latents, b->tab_host[(size_t)T_O_GRAD * R.H + h], dd * sizeof(double)));
    return APMF_OK;
}

// _approx_entropy() of every hypothesis (mn_active_pmf.py:513-521):
// (d log|cov_useritems| + (n+m) log|cov_latents|) / 2, from the factorisations of the last fit
int apmf_mn_batch_entropy(apmf_ctx *ctx, apmf_mn_batch *b, double *h_out)
{
    APMF_REQUIRE(ctx && b && h_out, "NULL operand");
    const Refs &R = b->R;
    for (int h = 0; h < R.H; h++)
        APMF_REQUIRE(b->oivalid[h], "entropy needs a fit (or apmf_mn_batch_kl) of the current states first");
    APMF_TRY(read_status(b));
    for (int h = 0; h < R.H; h++)
        h_out[h] = 0.5 * (R.d * b->hs_host[h].cur_ld_s + R.N * b->hs_host[h].cur_ld_o);
    return APMF_OK;
}

// Candidate-grid sweep (K2) straight from the resident states: hypotheses [h0, h0 + B) are
// packed into K2's operand layout on the device and swept in one launch.  Outputs as in
// apmf_mn_score_grid_f64 / _f32 (use_f32: float32 sweep, d_var / d_pmean / d_best_val are float).
int apmf_mn_batch_score_grid(apmf_ctx *ctx, apmf_mn_batch *b, int h0, int B, int use_f32, int ld,
                             const uint8_t *d_mask, void *d_var, void *d_pmean, double *d_sum, int want,
                             void *d_best_val, int64_t *d_best_idx)
{
    APMF_REQUIRE(ctx && b, "NULL operand");
    const Refs &R = b->R;
    APMF_REQUIRE(h0 >= 0 && B >= 1 && h0 + B <= R.H, "hypotheses [%d, %d) out of range", h0, h0 + B);
    APMF_REQUIRE(ld >= R.m && ld % APMF_GRID_ALIGN == 0, "bad grid pitch %d", ld);
    const size_t isz = use_f32 ? 4 : 8;
    const size_t n_cross = (size_t)R.n * ld, n_mean = (size_t)R.N * R.d, n_o = (size_t)R.d * R.d;
    const size_t per = n_cross + n_mean + R.N + n_o;
    void *pack = nullptr;
    APMF_TRY(apmf_malloc(ctx, (size_t)B * per * isz + 256, &pack));
    std::vector<int> sel(B);
    for (int y = 0; y < B; y++) sel[y] = h0 + y;
    const int *dsel = nullptr;
    int rc = APMF_OK;
    // contiguous selection list: reuse slot 0 (no fit is in flight on this stream between calls)
    rc = upload_sel(b, 0, sel, &dsel);
    char *p = (char *)pack;
    void *cross = p;                      // first: 16-byte aligned
    void *mean = p + (size_t)B * n_cross * isz;
    void *diag = (char *)mean + (size_t)B * n_mean * isz;
    void *omega = (char *)diag + (size_t)B * R.N * isz;
    if (rc == APMF_OK) {
        if (use_f32) {
            APMF_LAUNCH((kb_extract<float>), g2(per, B, 512), dim3(256), 0, ctx->stream, R, dsel, ld,
                               (float *)mean, (float *)cross, (float *)diag, (float *)omega);
            rc = apmf_mn_score_grid_f32(ctx, B, R.n, R.m, R.d, ld, (const float *)mean, (const float *)cross,
                                        (const float *)diag, (const float *)omega, d_mask, (float *)d_var,
                                        (float *)d_pmean, d_sum, want, (float *)d_best_val, d_best_idx);
        } else {
            APMF_LAUNCH((kb_extract<double>), g2(per, B, 512), dim3(256), 0, ctx->stream, R, dsel, ld,
                               (double *)mean, (double *)cross, (double *)diag, (double *)omega);
            rc = apmf_mn_score_grid_f64(ctx, B, R.n, R.m, R.d, ld, (const double *)mean, (const double *)cross,
                                        (const double *)diag, (const double *)omega, d_mask, (double *)d_var,
                                        (double *)d_pmean, d_sum, want, (double *)d_best_val, d_best_idx);
        }
    }
    (void)apmf_free(ctx, pack);      // stream-ordered reuse
    return rc;
}

// approx_pred_mean_var over an arbitrary pool (mn_active_pmf.py:300-315, 731-762) for hypothesis h
int apmf_mn_batch_score_pool(apmf_ctx *ctx, apmf_mn_batch *b, int h, const int32_t *d_pi, const int32_t *d_pj,
                             int64_t P, double *d_pmean, double *d_var)
{
    APMF_REQUIRE(ctx && b, "NULL operand");
    const Refs &R = b->R;
    APMF_REQUIRE(h >= 0 && h < R.H, "hypothesis %d out of range", h);
    APMF_REQUIRE(P >= 0, "negative pool size");
    if (P == 0) return APMF_OK;
    APMF_REQUIRE(d_pi && d_pj, "NULL operand");
    const int blocks = (int)(cdiv64(P, 256) < 2048 ? cdiv64(P, 256) : 2048);
    APMF_LAUNCH(kb_pool, dim3(blocks), dim3(256), (size_t)R.d * R.d * sizeof(double), ctx->stream, R, h, d_pi,
                       d_pj, P, d_pmean, d_var);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

// _total_variance() of every hypothesis (mn_active_pmf.py:597-598): the sum of the predictive
// variance over the WHOLE n x m grid, one float64 sweep with B = H (chunked to bound the pack)
int apmf_mn_batch_total_variance(apmf_ctx *ctx, apmf_mn_batch *b, double *h_out)
{
    APMF_REQUIRE(ctx && b && h_out, "NULL operand");
    const Refs &R = b->R;
    const int ld = round_up(R.m, APMF_GRID_ALIGN);
    double *d_sum = nullptr;
    APMF_TRY(apmf_malloc(ctx, (size_t)R.H * sizeof(double), (void **)&d_sum));
    const size_t per = ((size_t)R.n * ld + (size_t)R.N * (R.d + 1) + (size_t)R.d * R.d) * sizeof(double);
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)R.H, ((size_t)4 << 30) / per));
    int rc = APMF_OK;
    for (int h0 = 0; h0 < R.H && rc == APMF_OK; h0 += chunk) {
        const int B = std::min(chunk, R.H - h0);
        rc = apmf_mn_batch_score_grid(ctx, b, h0, B, 0, ld, nullptr, nullptr, nullptr, d_sum + h0, 0, nullptr, nullptr);
    }
    if (rc == APMF_OK) rc = apmf_d2h(ctx, h_out, d_sum, (size_t)R.H * sizeof(double));
    (void)apmf_free(ctx, d_sum);
    return rc;
}

// fit_normal_kls() / fit_normal() of ONE model whose approximation lives in caller buffers
// (include/apmf.h): a batch of one, so that the product's own fit and the lookahead's H fits are
// the same code.  d_mean / d_cov_useritems / d_cov_latents are updated in place.
int apmf_mn_fit_normal(apmf_ctx *ctx, const apmf_ratings *r, int d, double *d_mean, double *d_cov_useritems,
                       double *d_cov_latents, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                       double min_eig, double *h_lr, double *h_old_kl, int64_t max_steps, double *h_kl_trace,
                       int64_t trace_cap, int64_t *h_steps, int *h_converged)
{
    APMF_REQUIRE(ctx && r && d_mean && d_cov_useritems && d_cov_latents && h_lr && h_old_kl && h_steps &&
                     h_converged, "NULL operand");
    APMF_REQUIRE(trace_cap == 0 || h_kl_trace, "h_kl_trace is NULL");
    apmf_mn_batch *b = nullptr;
    const int64_t cap = std::min<int64_t>(trace_cap, 1 << 20);
    APMF_TRY(apmf_mn_batch_create(ctx, r, d, 1, cap, &b));
    int rc = apmf_mn_batch_set_state(ctx, b, d_mean, d_cov_useritems, d_cov_latents, 0);
    if (rc == APMF_OK)
        rc = apmf_mn_batch_fit(ctx, b, sigma_sq, sigma_u_sq, sigma_v_sq, min_eig, h_lr, h_old_kl, max_steps,
                               h_kl_trace, h_steps, h_converged);
    if (rc == APMF_OK && *h_steps > 0) rc = apmf_mn_batch_get_state(ctx, b, 0, d_mean, d_cov_useritems, d_cov_latents);
    if (rc == APMF_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {
        apmf_set_error("stream synchronisation failed");
        rc = APMF_EHIP;
    }
    (void)apmf_mn_batch_destroy(ctx, b);
    return rc;
}

} // extern "C"
