// k1_pmf.hip - PMF MAP model over the observed ratings: gradient, log-likelihood and the
// fit_lls learning-rate search (pmf_cy.pyx:158-291), float64 like the reference.
//
// The reference walks the (N,3) ratings array once per call, one np.dot per rating, and
// scatters into grad_u[i] / grad_v[j] sequentially.  Here the ratings live on the device
// as CSR-by-user and CSR-by-item (stable: per-row order = storage order), so each output
// row is owned by exactly one wave - a gather, no atomics, a fixed summation tree, hence
// bit-reproducible run to run (the accept/reject control flow of fit_lls depends on it).
//
// Wave layout: the 64 lanes are split into G = 64/DP groups of DP lanes (DP = latent_d
// rounded up to a power of two).  Lane k of a group holds component k.  A group handles
// one rating at a time: it gathers the other side's row (8*d contiguous bytes), forms the
// prediction with a DP-lane butterfly reduce, and accumulates err * row into its
// per-lane accumulator; UN ratings per group are kept in flight.  Groups are combined by
// a final cross-group butterfly.
#include <algorithm>
#include <new>
#include <vector>

#include "common.h"


namespace {

#ifndef APMF_K1_UN
#define APMF_K1_UN 4
#endif
constexpr int UN = APMF_K1_UN;

// DPP lane exchange inside a 16-lane row (VALU rate; no LDS crossbar round trip like
// ds_bpermute).  CTRL: 0xB1 = quad_perm[1,0,3,2] (xor 1), 0x4E = quad_perm[2,3,0,1] (xor 2),
// 0x141 = row_half_mirror (i <-> 7-i), 0x140 = row_mirror (i <-> 15-i).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    const long long b = __double_as_longlong(v);
    // every lane has a valid source under these controls, so no "old" value has to be set up
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// all-reduce (sum) over the DP lanes of a group; every lane of the group gets the total
template <int DP>
__device__ __forceinline__ double group_sum(double v)
{
    if (DP >= 2) v += dpp_move<0xB1>(v);
    if (DP >= 4) v += dpp_move<0x4E>(v);
    if (DP >= 8) v += dpp_move<0x141>(v);
    if (DP >= 16) v += dpp_move<0x140>(v);
    if (DP >= 32) v += __shfl_xor(v, 16, 64);
    if (DP >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// fl(x + fl(lr * g)): the trial point with numpy's two roundings (`users + lr * grad_u`,
// pmf_cy.pyx:271-272) - never contracted into one fma, and the same bits wherever it is formed
__device__ __forceinline__ double trial_value(double x, double lr, double g)
{
#pragma clang fp contract(off)
    const double step = lr * g;
    return x + step;
}

template <int DP>
__device__ __forceinline__ double cross_group_sum(double v)
{
#pragma unroll
    for (int o = DP; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct PmfParams {
    int n, m, d;
    double sig;          // sigma_sq (divided by, like the reference: pmf_cy.pyx:220-221)
    double sig_u, sig_v; // prior variances
    double mean_add;     // mean_rating if subtract_mean else 0
    double inv_sig;      // 1 / sig, exact when sig_pow2
    int sig_pow2;        // sig is a normal power of two well inside the exponent range
};

// ---- fit_lls state kept on the device (apmf_pmf_fit_lls) -----------------------------------
// One iteration of the search is ONE launch (k_fit_iter), enqueued back to back without host
// round trips: it decides the previous launch's trial from that launch's partial sums, then
// evaluates the next trial.  The state is double-buffered (launch `it` reads st[it & 1], its
// block 0 writes st[(it + 1) & 1]), and so are the partial sums.
constexpr int FIT_GRID_MAX = 1024;

struct FitState {
    double lr, old_ll, min_lr, stop_thresh;
    long long steps, max_steps, max_trace;
    int cur;        // which factor buffer pair holds the current point (0: caller's arrays)
    int gsel;       // which gradient buffer pair holds the gradient at the current point
    int converged;
    int pending;    // a trial has been evaluated (its partials are in part[]) and not yet decided
};

// a row slot of the whole-model launches: rows in descending-nnz order
struct RowDesc {
    int32_t row_all;    // user row, or n + item row
    int32_t len;        // its number of ratings
    int64_t beg;        // first entry in the CSR of its side
};
static_assert(sizeof(RowDesc) == 16, "RowDesc is one 16-byte load");

// Batched re-fits (apmf_pmf_batch_fit_lls): hypothesis h is the SHARED ratings container plus ONE extra
// rating (i, j, v), which the row kernels take as the last entry of user row i / item row j - exactly where
// add_rating() + a rebuilt container puts it (the new rating is the last row of the ratings array and the
// CSR build is a stable counting sort) - so that a batched fit is bit-identical to the same fit run alone
// on a rebuilt container.  The row's extended length (len + 1) is what the caller passes as `end`.
struct RowExtra {
    int64_t at;     // position of the extra entry: end - 1 of the extended row; -1 = this row has none
    int32_t c;      // its index on the other side
    double r;       // its value
};

struct FitBatch {
    const int *sel;            // blockIdx.y -> hypothesis
    const int32_t *ei, *ej;    // the extra rating of every hypothesis
    const double *ev;
    const double *mean_add;    // per-hypothesis mean offset (subtract_mean), or NULL: PmfParams'
    size_t nu, nv;             // strides of the per-hypothesis factor / gradient arrays (n d, m d)
    int nrows;                 // n + m: stride of the per-hypothesis row descriptors
};

template <bool EX>
__device__ __forceinline__ RowExtra row_extra(const FitBatch &B, int h, bool is_user, int row, int64_t end)
{
    RowExtra x;
    x.at = -1; x.c = 0; x.r = 0.;
    if (EX) {
        const int i = B.ei[h], j = B.ej[h];
        if (is_user ? row == i : row == j) {
            x.at = end - 1;
            x.c = is_user ? j : i;
            x.r = B.ev[h];
        }
    }
    return x;
}

// entry `tc` of a row for the loads: the extra entry is not in memory - its neighbour is read instead
// (an address inside the array: the row before it, or entry 0) and replaced afterwards
template <bool EX>
__device__ __forceinline__ int64_t mem_entry(int64_t tc, const RowExtra &ex)
{
    if (!EX) return tc;
    const int64_t tm = tc == ex.at ? tc - 1 : tc;
    return tm < 0 ? 0 : tm;
}

// One pass over (a share of) one CSR row.  The row's own vector is xk (lane k of every
// group); for each rating the other side's row y is gathered and the prediction formed with a
// DP-lane reduce.
//   GRAD : returns this wave's share of  sum_t y_t * (r_t - pred_t) / sigma^2   (lane k)
//   !GRAD: returns this wave's share of  sum_t (r_t - pred_t)^2                 (lanes k == 0)
// WPR waves share a row: wave `sub` takes ratings sub*G + g, + WPR*G, ...
template <int DP, int WPR, bool GRAD, bool EX = false>
__device__ __forceinline__ double row_pass(const PmfParams &p, int lane, int sub, int64_t beg, int64_t end,
                                           const int32_t *__restrict__ idx,
                                           const double *__restrict__ val, double xk,
                                           const double *__restrict__ Y, const RowExtra ex = RowExtra{-1, 0, 0.})
{
    constexpr int G = 64 / DP;
    constexpr int STRIDE = WPR * G;
    const int d = p.d;
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    const int kc = kok ? k : 0;
    double acc = 0.;
    const int64_t span = end - (beg + sub * G);      // wave-uniform trip count
    if (span <= 0) return acc;
    // every load unconditional (entries past the end are read at the row's last entry, padding lanes
    // at k = 0) and masked afterwards: a predicated load is a branch around the instruction, and a trip
    // of those is a chain of waits instead of UN loads in flight
    const int64_t last = end - 1;
    for (int64_t t0 = beg + sub * G + g, done = 0; done < span; t0 += (int64_t)STRIDE * UN, done += (int64_t)STRIDE * UN) {
        int32_t c[UN];
        double r[UN], y[UN];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int64_t t = t0 + (int64_t)u * STRIDE;
            ok[u] = t < end;
            const int64_t tc = ok[u] ? t : last;
            const int64_t tm = mem_entry<EX>(tc, ex);
            c[u] = idx[tm];
            r[u] = val[tm];
            if (EX && tc == ex.at) { c[u] = ex.c; r[u] = ex.r; }
        }
#pragma unroll
        for (int u = 0; u < UN; u++) y[u] = Y[(size_t)c[u] * d + kc];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const double yv = (ok[u] && kok) ? y[u] : 0.;
            const double pred = group_sum<DP>(xk * yv) + p.mean_add;
            if (GRAD) {
                const double w = ok[u] ? (r[u] - pred) / p.sig : 0.;
                acc += yv * w;
            } else {
                const double e = ok[u] ? (r[u] - pred) : 0.;
                if (k == 0) acc += e * e;
            }
        }
    }
    return acc;
}

// gradient(), pmf_cy.pyx:204-223.  A block is 4 waves = 4/WPR output rows (users and items in
// one launch); WPR waves split a long row and are combined through LDS in a fixed order.
template <int DP, int WPR>
__device__ __forceinline__ void grad_block(const PmfParams &p, int block, int first_row, int nrows_launch,
                                           const int32_t *__restrict__ order,
                                           const int64_t *__restrict__ ptr_u,
                                           const int32_t *__restrict__ idx_u,
                                           const double *__restrict__ val_u,
                                           const int64_t *__restrict__ ptr_v,
                                           const int32_t *__restrict__ idx_v,
                                           const double *__restrict__ val_v, const double *U,
                                           const double *V, double *gU, double *gV, double (*sh)[64])
{
    constexpr int RPB = 4 / WPR;
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int slot = block * RPB + wave / WPR;
    const int sub = wave % WPR;
    const bool valid = slot < nrows_launch;
    const int d = p.d;
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    int row = 0;
    bool is_user = true;
    double xk = 0., acc = 0.;
    if (valid) {
        // whole model: rows in descending-nnz order (heaviest first); a shard: rows first_row..
        const int row_all = order ? order[slot] : first_row + slot;  // wave-uniform
        is_user = row_all < p.n;
        row = is_user ? row_all : row_all - p.n;
        const int64_t *ptr = is_user ? ptr_u : ptr_v;
        const double *X = is_user ? U : V;
        xk = kok ? X[(size_t)row * d + k] : 0.;
        acc = row_pass<DP, WPR, true>(p, lane, sub, ptr[row], ptr[row + 1], is_user ? idx_u : idx_v,
                                      is_user ? val_u : val_v, xk, is_user ? V : U);
        acc = cross_group_sum<DP>(acc);
    }
    if (WPR > 1) {
        sh[wave][lane] = acc;
        __syncthreads();
        if (valid && sub == 0)
#pragma unroll
            for (int w = 1; w < WPR; w++) acc += sh[wave + w][lane];
    }
    if (valid && sub == 0 && g == 0 && kok)
        (is_user ? gU : gV)[(size_t)row * d + k] = -xk / (is_user ? p.sig_u : p.sig_v) + acc;
}

template <int DP, int WPR>
__global__ __launch_bounds__(256) void k_grad(const PmfParams p, int first_row, int nrows_launch,
                                              const int32_t *__restrict__ order,
                                              const int64_t *__restrict__ ptr_u,
                                              const int32_t *__restrict__ idx_u,
                                              const double *__restrict__ val_u,
                                              const int64_t *__restrict__ ptr_v,
                                              const int32_t *__restrict__ idx_v,
                                              const double *__restrict__ val_v,
                                              const double *__restrict__ U, const double *__restrict__ V,
                                              double *__restrict__ gU, double *__restrict__ gV)
{
    __shared__ double sh[4][64];
    grad_block<DP, WPR>(p, blockIdx.x, first_row, nrows_launch, order, ptr_u, idx_u, val_u, ptr_v, idx_v,
                        val_v, U, V, gU, gV, sh);
}

// The three sums of log_likelihood() (pmf_cy.pyx:170-193) as per-wave partials: squared error
// over the ratings of users [u_lo, u_hi) (each rating once) and the squared norms of the user
// and item rows of the launched ranges.
template <int DP, int WPR>
__global__ __launch_bounds__(256) void k_sqerr(const PmfParams p, int u_lo, int u_hi, int v_lo, int v_hi,
                                               const int64_t *__restrict__ ptr_u,
                                               const int32_t *__restrict__ idx_u,
                                               const double *__restrict__ val_u,
                                               const double *__restrict__ U, const double *__restrict__ V,
                                               double *__restrict__ part /* (nslots * WPR, 3) */)
{
    constexpr int RPB = 4 / WPR;
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int slot = blockIdx.x * RPB + wave / WPR;
    const int sub = wave % WPR;
    const int nrows = max(u_hi - u_lo, v_hi - v_lo);
    if (slot >= nrows) return;
    const int d = p.d;
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    double sq = 0., un = 0., vn = 0.;
    if (u_lo + slot < u_hi) {
        const int urow = u_lo + slot;
        const double xk = kok ? U[(size_t)urow * d + k] : 0.;
        if (sub == 0 && g == 0 && kok) un = xk * xk;
        sq = row_pass<DP, WPR, false>(p, lane, sub, ptr_u[urow], ptr_u[urow + 1], idx_u, val_u, xk, V);
    }
    if (sub == 0 && v_lo + slot < v_hi && g == 0 && kok) {
        const double vk = V[(size_t)(v_lo + slot) * d + k];
        vn = vk * vk;
    }
    sq = wave_sum(sq);
    un = wave_sum(un);
    vn = wave_sum(vn);
    if (lane == 0) {
        const size_t e = ((size_t)slot * WPR + sub) * 3;
        part[e + 0] = sq;
        part[e + 1] = un;
        part[e + 2] = vn;
    }
}

// gradient() of hypothesis blockIdx.y at its own factors, the extra rating included (the starting gradient
// of a batched re-fit): the arithmetic of k_grad row by row
template <int DP, int WPR>
__global__ __launch_bounds__(256) void k_grad_batch(const PmfParams p_in, const FitBatch B,
                                                    const int64_t *__restrict__ ptr_u,
                                                    const int32_t *__restrict__ idx_u,
                                                    const double *__restrict__ val_u,
                                                    const int64_t *__restrict__ ptr_v,
                                                    const int32_t *__restrict__ idx_v,
                                                    const double *__restrict__ val_v,
                                                    const double *__restrict__ U, const double *__restrict__ V,
                                                    double *__restrict__ gU, double *__restrict__ gV)
{
    __shared__ double sh[4][64];
    constexpr int RPB = 4 / WPR;
    PmfParams p = p_in;
    const int hyp = B.sel[blockIdx.y];
    if (B.mean_add) p.mean_add = B.mean_add[hyp];
    U += hyp * B.nu; gU += hyp * B.nu;
    V += hyp * B.nv; gV += hyp * B.nv;
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int slot = blockIdx.x * RPB + wave / WPR;
    const int sub = wave % WPR;
    const bool valid = slot < B.nrows;
    const int d = p.d;
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    int row = 0;
    bool is_user = true;
    double xk = 0., acc = 0.;
    if (valid) {
        is_user = slot < p.n;
        row = is_user ? slot : slot - p.n;
        const int64_t *ptr = is_user ? ptr_u : ptr_v;
        const double *X = is_user ? U : V;
        xk = kok ? X[(size_t)row * d + k] : 0.;
        int64_t end = ptr[row + 1];
        const int xi = B.ei[hyp], xj = B.ej[hyp];
        if (is_user ? row == xi : row == xj) end += 1;
        const RowExtra ex = row_extra<true>(B, hyp, is_user, row, end);
        acc = row_pass<DP, WPR, true, true>(p, lane, sub, ptr[row], end, is_user ? idx_u : idx_v,
                                            is_user ? val_u : val_v, xk, is_user ? V : U, ex);
        acc = cross_group_sum<DP>(acc);
    }
    if (WPR > 1) {
        sh[wave][lane] = acc;
        __syncthreads();
        if (valid && sub == 0)
#pragma unroll
            for (int w = 1; w < WPR; w++) acc += sh[wave + w][lane];
    }
    if (valid && sub == 0 && g == 0 && kok)
        (is_user ? gU : gV)[(size_t)row * d + k] = -xk / (is_user ? p.sig_u : p.sig_v) + acc;
}

// the per-wave partials of k_sqerr for hypothesis blockIdx.y (whole model), the extra rating included
template <int DP, int WPR>
__global__ __launch_bounds__(256) void k_sqerr_batch(const PmfParams p_in, const FitBatch B,
                                                     const int64_t *__restrict__ ptr_u,
                                                     const int32_t *__restrict__ idx_u,
                                                     const double *__restrict__ val_u,
                                                     const double *__restrict__ U, const double *__restrict__ V,
                                                     double *__restrict__ part /* (H, nslots * WPR, 3) */, int nslots)
{
    constexpr int RPB = 4 / WPR;
    PmfParams p = p_in;
    const int hyp = B.sel[blockIdx.y];
    if (B.mean_add) p.mean_add = B.mean_add[hyp];
    U += hyp * B.nu;
    V += hyp * B.nv;
    part += (size_t)hyp * nslots * WPR * 3;
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int slot = blockIdx.x * RPB + wave / WPR;
    const int sub = wave % WPR;
    if (slot >= nslots) return;
    const int d = p.d;
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    double sq = 0., un = 0., vn = 0.;
    if (slot < p.n) {
        const int urow = slot;
        const double xk = kok ? U[(size_t)urow * d + k] : 0.;
        if (sub == 0 && g == 0 && kok) un = xk * xk;
        int64_t end = ptr_u[urow + 1];
        if (urow == B.ei[hyp]) end += 1;
        const RowExtra ex = row_extra<true>(B, hyp, true, urow, end);
        sq = row_pass<DP, WPR, false, true>(p, lane, sub, ptr_u[urow], end, idx_u, val_u, xk, V, ex);
    }
    if (sub == 0 && slot < p.m && g == 0 && kok) {
        const double vk = V[(size_t)slot * d + k];
        vn = vk * vk;
    }
    sq = wave_sum(sq);
    un = wave_sum(un);
    vn = wave_sum(vn);
    if (lane == 0) {
        const size_t e = ((size_t)slot * WPR + sub) * 3;
        part[e + 0] = sq;
        part[e + 1] = un;
        part[e + 2] = vn;
    }
}

// fixed-order combine of the per-wave partials (one 256-thread block) -> sums[3]
__device__ __forceinline__ void combine_parts(int nparts, const double *__restrict__ part, double (*sh)[256],
                                              double (&sums)[3])
{
    double a = 0, b = 0, c = 0;
    for (int t = threadIdx.x; t < nparts; t += 256) {
        a += part[(size_t)t * 3];
        b += part[(size_t)t * 3 + 1];
        c += part[(size_t)t * 3 + 2];
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b; sh[2][threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int q = 0; q < 3; q++) sh[q][threadIdx.x] += sh[q][threadIdx.x + o];
        __syncthreads();
    }
    sums[0] = sh[0][0]; sums[1] = sh[1][0]; sums[2] = sh[2][0];
}

// out[0] = LL, out[1] = sq_error, out[2] = ||U||^2, out[3] = ||V||^2
__global__ __launch_bounds__(256) void k_ll_finish(int nparts, const double *__restrict__ part, double sig,
                                                   double sig_u, double sig_v, double *__restrict__ out)
{
    __shared__ double sh[3][256];
    double s[3];
    combine_parts(nparts, part, sh, s);
    if (threadIdx.x == 0) {
        out[0] = -s[0] / (2. * sig) - s[1] / (2. * sig_u) - s[2] / (2. * sig_v);
        out[1] = s[0];
        out[2] = s[1];
        out[3] = s[2];
    }
}

// k_ll_finish for hypothesis blockIdx.x: out[h] = LL
__global__ __launch_bounds__(256) void k_ll_finish_batch(int nparts, const double *__restrict__ part, double sig,
                                                         double sig_u, double sig_v, const int *__restrict__ sel,
                                                         double *__restrict__ out)
{
    __shared__ double sh[3][256];
    double s[3];
    const int hyp = sel[blockIdx.x];
    combine_parts(nparts, part + (size_t)hyp * nparts * 3, sh, s);
    if (threadIdx.x == 0) out[hyp] = -s[0] / (2. * sig) - s[1] / (2. * sig_u) - s[2] / (2. * sig_v);
}

// ---- one search iteration in ONE launch ------------------------------------------------------
// Every row (users then items, heaviest first) is visited once at the trial point
// t = fl(cur + fl(lr * grad)), formed on the fly for the row itself and for every gathered row of
// the other side.  The one pass yields both  sum (r - pred)^2  (user rows: each rating once) and
// the gradient at t (speculative: it is the next step's gradient whenever the trial is accepted),
// and leaves three partial sums per block.  The accept / reject decision on that trial is taken at
// the START of the next launch, by every block for itself from the partials (a fixed-order sum, so
// all agree): no atomics, no fences, no second launch.  (A last-block-done ticket was measured
// first: 2625 device-scope atomics on one address serialise to 44 us per launch.)
//
// What bounds the launch is the chain of dependent loads, each a round trip of 0.7 us (L2) to
// 2.5 us (data another XCD wrote in the previous launch): state -> partials -> [decision] ->
// row -> ptr -> idx -> gather.  So everything that does not need the decision is requested
// BEFORE it is known: the row descriptor (one 16-byte record per slot instead of order -> ptr),
// the row's first entries, and the first trip's gathers from BOTH factor / gradient pairs - the
// decision then only selects among values already in registers.
template <int DP, int WPR, int UNT>
__device__ __forceinline__ void trip_compute(const PmfParams &p, bool kok, int k, double xk, double lr,
                                             const double (&y)[UNT], const double (&gy)[UNT],
                                             const double (&r)[UNT], const bool (&ok)[UNT], double &acc_g,
                                             double &acc_sq)
{
#pragma unroll
    for (int u = 0; u < UNT; u++) {
        const double yt = (ok[u] && kok) ? trial_value(y[u], lr, gy[u]) : 0.;
        const double pred = group_sum<DP>(xk * yt) + p.mean_add;
        const double e = ok[u] ? (r[u] - pred) : 0.;
        // (r - pred) / sigma^2 like the reference; a power-of-two sigma^2 (the default 1)
        // divides exactly by multiplying with its reciprocal
        acc_g += yt * (p.sig_pow2 ? e * p.inv_sig : e / p.sig);
        if (k == 0) acc_sq += e * e;
    }
}

// entries t0 + u*STRIDE of a row; past the end they are read at the last entry and masked (a
// predicated load is a branch around the instruction, and a trip of those is a chain of waits
// instead of UNT loads in flight)
template <int STRIDE, int UNT, bool EX = false>
__device__ __forceinline__ void load_entries(int64_t t0, int64_t end, const int32_t *__restrict__ idx,
                                             const double *__restrict__ val, int32_t (&c)[UNT],
                                             double (&r)[UNT], bool (&ok)[UNT], const RowExtra ex = RowExtra{-1, 0, 0.})
{
#pragma unroll
    for (int u = 0; u < UNT; u++) {
        const int64_t t = t0 + (int64_t)u * STRIDE;
        ok[u] = t < end;
        const int64_t tc = ok[u] ? t : end - 1;
        const int64_t tm = mem_entry<EX>(tc, ex);
        c[u] = idx[tm];
        r[u] = val[tm];
        if (EX && tc == ex.at) { c[u] = ex.c; r[u] = ex.r; }
    }
}

// The rest of a row whose first trip's entries (c, r, ok) and gathers (y, gy) are in registers.
template <int DP, int WPR, int UNT, bool EX = false>
__device__ __forceinline__ void row_pass_trial(const PmfParams &p, int lane, int sub, int64_t beg,
                                               int64_t end, const int32_t *__restrict__ idx,
                                               const double *__restrict__ val, double xk, const double *Y,
                                               const double *gY, double lr, int32_t (&c)[UNT],
                                               double (&r)[UNT], bool (&ok)[UNT], double (&y)[UNT],
                                               double (&gy)[UNT], double &acc_g, double &acc_sq,
                                               const RowExtra ex = RowExtra{-1, 0, 0.})
{
    constexpr int G = 64 / DP;
    constexpr int STRIDE = WPR * G;
    const int d = p.d;
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    const int kc = kok ? k : 0;
    // wave-uniform trip count (the groups of a wave differ by at most one rating): every lane
    // takes part in the DPP reductions of every trip
    const int64_t span = end - (beg + sub * G);
    int64_t t0 = beg + sub * G + g;
    for (int64_t done = 0; done < span; done += (int64_t)STRIDE * UNT) {
        // the next trip's entries are requested before this trip's gathers are consumed
        int32_t cn[UNT];
        double rn[UNT];
        bool okn[UNT];
        t0 += (int64_t)STRIDE * UNT;
        load_entries<STRIDE, UNT, EX>(t0, end, idx, val, cn, rn, okn, ex);
        trip_compute<DP, WPR, UNT>(p, kok, k, xk, lr, y, gy, r, ok, acc_g, acc_sq);
        if (done + (int64_t)STRIDE * UNT < span) {
#pragma unroll
            for (int u = 0; u < UNT; u++) {
                const size_t oy = (size_t)cn[u] * d + kc;
                y[u] = Y[oy];
                gy[u] = gY[oy];
            }
        }
#pragma unroll
        for (int u = 0; u < UNT; u++) {
            c[u] = cn[u];
            r[u] = rn[u];
            ok[u] = okn[u];
        }
    }
}

#ifdef APMF_K1_TIMING
__device__ unsigned long long g_k1_stamps[8 * 16384];
#define K1_STAMP(i)                                                                         \
    do {                                                                                    \
        if ((threadIdx.x & 63) == 0) {                                                      \
            const unsigned wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                      \
            if (wid < 16384) g_k1_stamps[wid * 8 + (i)] = wall_clock64();                   \
            if ((i) == 1 && wid < 16384) g_k1_stamps[wid * 8] = k1_t_start;                 \
        }                                                                                   \
    } while (0)
#define K1_START() const unsigned long long k1_t_start = wall_clock64()
#else
#define K1_STAMP(i)
#define K1_START()
#endif

// A block is RPB rows x WPR waves.
// EX (apmf_pmf_batch_fit_lls): gridDim.y hypotheses, each with its own state, partial sums, row descriptors
// (the order a rebuilt container would have), factor / gradient arrays and one extra rating (FitBatch).
template <int DP, int WPR, int RPB, int UNT, bool EX = false>
__global__ __launch_bounds__(64 * RPB * WPR) void k_fit_iter(const PmfParams p_in, const FitState *__restrict__ st_in,
                                                  FitState *__restrict__ st_out, int nparts_in,
                                                  const double *__restrict__ part_in,
                                                  double *__restrict__ part_out,
                                                  const RowDesc *__restrict__ desc,
                                                  const int32_t *__restrict__ idx_u,
                                                  const double *__restrict__ val_u,
                                                  const int32_t *__restrict__ idx_v,
                                                  const double *__restrict__ val_v, double *U0, double *V0,
                                                  double *U1, double *V1, double *gU0, double *gV0,
                                                  double *gU1, double *gV1, double sig,
                                                  double *__restrict__ trace, const FitBatch B = FitBatch{})
{
    PmfParams p = p_in;
    int hyp = 0;
    if (EX) {
        hyp = B.sel[blockIdx.y];
        if (B.mean_add) p.mean_add = B.mean_add[hyp];
        st_in += hyp;
        st_out += hyp;
        part_in += (size_t)hyp * 3 * FIT_GRID_MAX;
        part_out += (size_t)hyp * 3 * FIT_GRID_MAX;
        desc += (size_t)hyp * B.nrows;
        U0 += hyp * B.nu; U1 += hyp * B.nu; gU0 += hyp * B.nu; gU1 += hyp * B.nu;
        V0 += hyp * B.nv; V1 += hyp * B.nv; gV0 += hyp * B.nv; gV1 += hyp * B.nv;
    }
    constexpr int NW = RPB * WPR;
    __shared__ double sh_g[NW][64];
    __shared__ double sh_p[NW][3];
    constexpr int G = 64 / DP;
    constexpr int STRIDE = WPR * G;
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int sub = wave % WPR;
    const int d = p.d;
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    const int kc = kok ? k : 0;
    const int nrows = p.n + p.m;
    K1_START();

    // ---- (0) requested before the decision is known ----------------------------------------
    // partials of the previous launch (zeros before the first trial): the block's waves take every
    // NW-th group of 64 each, all loads of a chunk in flight together (a plain loop is one round trip
    // per 64 partials)
    double a0 = 0., a1 = 0., a2 = 0.;
    {
        constexpr int CH = 8;
        for (int base = wave * 64; base < nparts_in; base += NW * 64 * CH) {
            double v0[CH], v1[CH], v2[CH];
#pragma unroll
            for (int q = 0; q < CH; q++) {
                const int t = base + q * NW * 64 + lane;
                const int tc = t < nparts_in ? t : 0;
                v0[q] = part_in[tc];
                v1[q] = part_in[FIT_GRID_MAX + tc];
                v2[q] = part_in[2 * FIT_GRID_MAX + tc];
            }
#pragma unroll
            for (int q = 0; q < CH; q++) {
                const bool in = base + q * NW * 64 + lane < nparts_in;
                a0 += in ? v0[q] : 0.;
                a1 += in ? v1[q] : 0.;
                a2 += in ? v2[q] : 0.;
            }
        }
    }
    // this wave's first row: descriptor, first entries, own vector and first gathers from both
    // buffer pairs
    int slot = blockIdx.x * RPB + wave / WPR;
    bool valid = slot < nrows;
    RowDesc rd = desc[valid ? slot : 0];
    int row_all = __builtin_amdgcn_readfirstlane(rd.row_all);
    bool is_user = row_all < p.n;
    int row = is_user ? row_all : row_all - p.n;
    int64_t beg = ((int64_t)__builtin_amdgcn_readfirstlane((int)(rd.beg >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)rd.beg);
    int64_t end = beg + __builtin_amdgcn_readfirstlane(rd.len);
    bool has = valid && end - (beg + sub * G) > 0;      // wave-uniform
    RowExtra ex = row_extra<EX>(B, hyp, is_user, row, end);
    int32_t c[UNT];
    double r[UNT], y[UNT], gy[UNT], y1[UNT], gy1[UNT];
    bool ok[UNT];
#pragma unroll
    for (int u = 0; u < UNT; u++) { c[u] = 0; r[u] = 0.; ok[u] = false; y[u] = gy[u] = y1[u] = gy1[u] = 0.; }
    double x0 = 0., x1 = 0., gx0 = 0., gx1 = 0.;
    if (valid) {
        const size_t o = (size_t)row * d + kc;
        x0 = (is_user ? U0 : V0)[o];
        x1 = (is_user ? U1 : V1)[o];
        gx0 = (is_user ? gU0 : gV0)[o];
        gx1 = (is_user ? gU1 : gV1)[o];
    }
    if (has) {
        load_entries<STRIDE, UNT, EX>(beg + sub * G + g, end, is_user ? idx_u : idx_v, is_user ? val_u : val_v, c, r,
                                      ok, ex);
#pragma unroll
        for (int u = 0; u < UNT; u++) {
            const size_t oy = (size_t)c[u] * d + kc;
            y[u] = (is_user ? V0 : U0)[oy];
            y1[u] = (is_user ? V1 : U1)[oy];
            gy[u] = (is_user ? gV0 : gU0)[oy];
            gy1[u] = (is_user ? gV1 : gU1)[oy];
        }
    }

    // ---- (1) the decision on the trial of the previous launch ---------------------------------
    FitState s = *st_in;
    if (s.pending) {          // block-uniform
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        a2 = wave_sum(a2);
        if (lane == 0) { sh_p[wave][0] = a0; sh_p[wave][1] = a1; sh_p[wave][2] = a2; }
        __syncthreads();
        double b0 = sh_p[0][0], b1 = sh_p[0][1], b2 = sh_p[0][2];
#pragma unroll
        for (int w = 1; w < NW; w++) { b0 += sh_p[w][0]; b1 += sh_p[w][1]; b2 += sh_p[w][2]; }
        __syncthreads();      // sh_p is reused for the partials of this launch
        const double ll = -b0 / (2. * sig) - b1 / (2. * p.sig_u) - b2 / (2. * p.sig_v);
        const bool first = blockIdx.x == 0 && threadIdx.x == 0;
        if (ll > s.old_ll) {      // pmf_cy.pyx:275-285
            if (first && s.steps < s.max_trace) trace[s.steps] = ll;
            if (ll - s.old_ll < s.stop_thresh) s.converged = 1;
            s.cur ^= 1;
            s.gsel ^= 1;
            s.lr = s.lr * 1.25;
            s.steps += 1;
            s.old_ll = ll;
        } else {                  // pmf_cy.pyx:286-291
            s.lr = s.lr * .5;
            if (s.lr < s.min_lr) s.converged = 1;
        }
    }
    const bool done = s.converged || (s.max_steps >= 0 && s.steps >= s.max_steps);
    s.pending = done ? 0 : 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) *st_out = s;
    if (done) return;
    K1_STAMP(1);

    // ---- (2) the next trial: LL partials and the gradient at the trial point -------------------
    const double lr = s.lr;
    const double *cU = s.cur ? U1 : U0, *cV = s.cur ? V1 : V0;       // current point
    double *tU = s.cur ? U0 : U1, *tV = s.cur ? V0 : V1;             // trial point
    const double *gU = s.gsel ? gU1 : gU0, *gV = s.gsel ? gV1 : gV0; // gradient at the current point
    double *nU = s.gsel ? gU0 : gU1, *nV = s.gsel ? gV0 : gV1;       // gradient at the trial point
    double xc = s.cur ? x1 : x0, gxc = s.gsel ? gx1 : gx0;
#pragma unroll
    for (int u = 0; u < UNT; u++) {
        if (s.cur) y[u] = y1[u];
        if (s.gsel) gy[u] = gy1[u];
    }
    double sq = 0., un = 0., vn = 0.;
    for (int base = blockIdx.x * RPB;;) {   // block-uniform trip count
        double xk = 0., acc = 0., sq_row = 0.;
        if (valid) {
            const size_t o = (size_t)row * d + k;
            if (kok) xk = trial_value(xc, lr, gxc);
            if (sub == 0 && g == 0 && kok) {
                (is_user ? tU : tV)[o] = xk;
                if (is_user) un += xk * xk;
                else vn += xk * xk;
            }
            K1_STAMP(2);
            if (has)
                row_pass_trial<DP, WPR, UNT, EX>(p, lane, sub, beg, end, is_user ? idx_u : idx_v,
                                                 is_user ? val_u : val_v, xk, is_user ? cV : cU,
                                                 is_user ? gV : gU, lr, c, r, ok, y, gy, acc, sq_row, ex);
            K1_STAMP(3);
            acc = cross_group_sum<DP>(acc);
            if (is_user) sq += sq_row;
        }
        if (WPR > 1) {
            sh_g[wave][lane] = acc;
            __syncthreads();
            if (valid && sub == 0)
#pragma unroll
                for (int w = 1; w < WPR; w++) acc += sh_g[wave + w][lane];
        }
        if (valid && sub == 0 && g == 0 && kok)
            (is_user ? nU : nV)[(size_t)row * d + k] = -xk / (is_user ? p.sig_u : p.sig_v) + acc;
        base += gridDim.x * RPB;
        if (base >= nrows) break;
        if (WPR > 1) __syncthreads();      // this trip's reads of sh_g are over
        // the block's next row, the decision being known
        slot = base + wave / WPR;
        valid = slot < nrows;
        rd = desc[valid ? slot : 0];
        row_all = __builtin_amdgcn_readfirstlane(rd.row_all);
        is_user = row_all < p.n;
        row = is_user ? row_all : row_all - p.n;
        beg = ((int64_t)__builtin_amdgcn_readfirstlane((int)(rd.beg >> 32)) << 32) |
              (uint32_t)__builtin_amdgcn_readfirstlane((int)rd.beg);
        end = beg + __builtin_amdgcn_readfirstlane(rd.len);
        has = valid && end - (beg + sub * G) > 0;
        ex = row_extra<EX>(B, hyp, is_user, row, end);
        if (valid) {
            const size_t o = (size_t)row * d + kc;
            xc = (is_user ? cU : cV)[o];
            gxc = (is_user ? gU : gV)[o];
        }
        if (has) {
            load_entries<STRIDE, UNT, EX>(beg + sub * G + g, end, is_user ? idx_u : idx_v, is_user ? val_u : val_v, c,
                                          r, ok, ex);
#pragma unroll
            for (int u = 0; u < UNT; u++) {
                const size_t oy = (size_t)c[u] * d + kc;
                y[u] = (is_user ? cV : cU)[oy];
                gy[u] = (is_user ? gV : gU)[oy];
            }
        }
    }
    sq = wave_sum(sq);
    un = wave_sum(un);
    vn = wave_sum(vn);
    if (lane == 0) {
        sh_p[wave][0] = sq;
        sh_p[wave][1] = un;
        sh_p[wave][2] = vn;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int q = threadIdx.x;
        double tot = sh_p[0][q];
#pragma unroll
        for (int w = 1; w < NW; w++) tot += sh_p[w][q];
        part_out[(size_t)q * FIT_GRID_MAX + blockIdx.x] = tot;
    }
    K1_STAMP(4);
}

// minibatch momentum update, pmf_cy.pyx:338-344:
//   inc = momentum*inc + (prior + data)*(lr/n);  param += inc
__global__ __launch_bounds__(256) void k_momentum(size_t nu, size_t nv, double momentum, double scale,
                                                  double sig_u, double sig_v, double *__restrict__ U,
                                                  double *__restrict__ V, double *__restrict__ uinc,
                                                  double *__restrict__ vinc, double *__restrict__ gaccU,
                                                  double *__restrict__ gaccV)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < nu + nv; t += stride) {
        const bool isu = t < nu;
        const size_t e = isu ? t : t - nu;
        double *X = isu ? U : V, *inc = isu ? uinc : vinc, *ga = isu ? gaccU : gaccV;
        const double grad = -X[e] / (isu ? sig_u : sig_v) + ga[e];
        const double ni = __dadd_rn(__dmul_rn(inc[e], momentum), __dmul_rn(grad, scale));
        inc[e] = ni;
        X[e] = __dadd_rn(X[e], ni);
        ga[e] = 0.;
    }
}

// data term of the gradient for the rows present in one minibatch (CSR over the batch)
template <int DP>
__global__ __launch_bounds__(256) void k_pmf_grad_rows(const PmfParams p, int nrows_present,
                                                       const int32_t *__restrict__ rows /* >= n: item */,
                                                       const int64_t *__restrict__ ptr,
                                                       const int32_t *__restrict__ idx,
                                                       const double *__restrict__ val,
                                                       const double *__restrict__ U,
                                                       const double *__restrict__ V,
                                                       double *__restrict__ gaccU,
                                                       double *__restrict__ gaccV)
{
    constexpr int G = 64 / DP;
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + wave_id_uniform();
    if (slot >= nrows_present) return;
    const int row_all = rows[slot];
    const bool is_user = row_all < p.n;
    const int row = is_user ? row_all : row_all - p.n;
    const double *X = is_user ? U : V, *Y = is_user ? V : U;
    double *gout = is_user ? gaccU : gaccV;
    const int d = p.d;
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    const double xk = kok ? X[(size_t)row * d + k] : 0.;
    const int64_t beg = ptr[slot], end = ptr[slot + 1];
    double acc = 0.;
    for (int64_t t0 = beg + g; t0 < end; t0 += (int64_t)G * UN) {
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int64_t t = t0 + (int64_t)u * G;
            const bool ok = t < end;
            const int32_t c = ok ? idx[t] : 0;
            const double r = ok ? val[t] : 0.;
            const double y = (ok && kok) ? Y[(size_t)c * d + k] : 0.;
            const double pred = group_sum<DP>(xk * y) + p.mean_add;
            const double w = ok ? (r - pred) / p.sig : 0.;
            acc += y * w;
        }
    }
    acc = cross_group_sum<DP>(acc);
    if (g == 0 && kok) gout[(size_t)row * d + k] = acc;
}

// predicted_matrix(): out[i, j] = U_i . V_j + add
__global__ __launch_bounds__(256) void k_predict(int n, int m, int d, int ld, double add,
                                                 const double *__restrict__ U, const double *__restrict__ V,
                                                 double *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * 16;
    if (j >= ld) return;
    for (int i = i0; i < min(i0 + 16, n); i++) {
        double s = 0.;
        if (j < m)
            for (int k = 0; k < d; k++) s += U[(size_t)i * d + k] * V[(size_t)j * d + k];
        out[(size_t)i * ld + j] = j < m ? s + add : 0.;
    }
}

int dp_for(int d)
{
    int dp = 4;
    while (dp < d) dp <<= 1;
    return dp;
}

#define DISPATCH_DP(dp, ...)                                                   \
    switch (dp) {                                                              \
    case 4: { constexpr int DP = 4; __VA_ARGS__; } break;                      \
    case 8: { constexpr int DP = 8; __VA_ARGS__; } break;                      \
    case 16: { constexpr int DP = 16; __VA_ARGS__; } break;                    \
    case 32: { constexpr int DP = 32; __VA_ARGS__; } break;                    \
    case 64: { constexpr int DP = 64; __VA_ARGS__; } break;                    \
    default: apmf_set_error("latent_d too large"); return APMF_EINVAL;         \
    }

PmfParams make_params(const apmf_ratings *r, int d, double sigma_sq, double sigma_u_sq,
                      double sigma_v_sq, int subtract_mean, double mean_rating)
{
    PmfParams p;
    p.n = r->n; p.m = r->m; p.d = d;
    p.sig = sigma_sq;
    p.sig_u = sigma_u_sq; p.sig_v = sigma_v_sq;
    p.mean_add = subtract_mean ? mean_rating : 0.;
    int ex = 0;
    const double mant = frexp(sigma_sq, &ex);
    p.sig_pow2 = (mant == 0.5 && ex > -500 && ex < 500) ? 1 : 0;
    p.inv_sig = 1. / sigma_sq;
    return p;
}

int check_common(apmf_ctx *ctx, const apmf_ratings *r, int d, double sigma_sq, double sigma_u_sq,
                 double sigma_v_sq)
{
    APMF_REQUIRE(ctx && r, "ctx/ratings is NULL");
    APMF_REQUIRE(d >= 1 && d <= APMF_MAX_D, "latent_d=%d unsupported (1..%d)", d, APMF_MAX_D);
    APMF_REQUIRE(sigma_sq != 0. && sigma_u_sq != 0. && sigma_v_sq != 0., "zero variance");
    return APMF_OK;
}

// waves per row: long rows (MovieLens-like: ~100 ratings per user) are split over 4 waves
int wpr_for(const apmf_ratings *r)
{
    const int64_t rows = std::max(1, std::min(r->n, r->m));
    return (r->N / rows) >= 48 ? 4 : 1;
}

#define DISPATCH_DP_WPR(dp, wpr, ...)                                           \
    if ((wpr) == 4) { constexpr int WPR = 4; DISPATCH_DP(dp, __VA_ARGS__); }        \
    else { constexpr int WPR = 1; DISPATCH_DP(dp, __VA_ARGS__); }

int launch_grad(apmf_ctx *ctx, const apmf_ratings *r, const PmfParams &p, const double *U,
                const double *V, double *gU, double *gV, int first_row = 0, int nrows = -1)
{
    const bool whole = nrows < 0;
    const int rows = whole ? r->n + r->m : nrows;
    if (rows == 0) return APMF_OK;
    const int wpr = wpr_for(r);
    const int blocks = (rows + (4 / wpr) - 1) / (4 / wpr);
    ProfScope prof(ctx);
    DISPATCH_DP_WPR(dp_for(p.d), wpr,
                    hipLaunchKernelGGL((k_grad<DP, WPR>), dim3(blocks), dim3(256), 0, ctx->stream, p,
                                       first_row, rows, whole ? r->order : (const int32_t *)nullptr,
                                       r->ptr[0], r->idx[0], r->val[0], r->ptr[1], r->idx[1], r->val[1],
                                       U, V, gU, gV));
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

int ll_parts(const apmf_ratings *r, int u_lo, int u_hi, int v_lo, int v_hi)
{
    return std::max(std::max(u_hi - u_lo, v_hi - v_lo), 1) * wpr_for(r);
}

// LL of (U, V) -> d_out4 = {ll, sq_error, ||U||^2, ||V||^2} (device), no sync
int launch_ll(apmf_ctx *ctx, const apmf_ratings *r, const PmfParams &p, double sigma_sq,
              const double *U, const double *V, double *d_part, double *d_out4, int u_lo = 0,
              int u_hi = -1, int v_lo = 0, int v_hi = -1)
{
    if (u_hi < 0) u_hi = r->n;
    if (v_hi < 0) v_hi = r->m;
    const int wpr = wpr_for(r);
    const int nslots = std::max(std::max(u_hi - u_lo, v_hi - v_lo), 1);
    const int blocks = (nslots + (4 / wpr) - 1) / (4 / wpr);
    APMF_CHECK_HIP(hipMemsetAsync(d_part, 0, (size_t)nslots * wpr * 24, ctx->stream));
    DISPATCH_DP_WPR(dp_for(p.d), wpr,
                    hipLaunchKernelGGL((k_sqerr<DP, WPR>), dim3(blocks), dim3(256), 0, ctx->stream, p, u_lo,
                                       u_hi, v_lo, v_hi, r->ptr[0], r->idx[0], r->val[0], U, V, d_part));
    APMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ll_finish, dim3(1), dim3(256), 0, ctx->stream, nslots * wpr, d_part, sigma_sq,
                       p.sig_u, p.sig_v, d_out4);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

} // namespace

// ---- ratings container straight from a dense matrix + "known" mask ----------
// The input contract of the reference's drivers is a dense `_real` matrix plus the list `_ratings`
// that choose_training.py's make_ratings() writes in row-major order of the known mask
// (choose_training.py:13-17, 215-259).  For such data the container can be built on the device from
// the dense arrays: per-row / per-column counts, prefix sums (host: n + m integers), then an ordered
// compaction - one wave per row walks its columns 64 at a time and places the known entries by
// ballot prefix (ascending j), one wave per column likewise (ascending i).  That is exactly the
// stable order apmf_ratings_create gives the row-major list, so every sum downstream is
// bit-identical.
namespace {

__global__ __launch_bounds__(64) void k_dense_count(int n, int m, const uint8_t *__restrict__ known, int ldk,
                                                    int *__restrict__ cnt)      // cnt[n + m], zeroed
{
    const int i = blockIdx.x, lane = threadIdx.x;
    int c = 0;
    for (int j0 = 0; j0 < m; j0 += 64) {
        const int j = j0 + lane;
        const bool k = j < m && known[(size_t)i * ldk + j];
        if (k) atomicAdd(&cnt[n + j], 1);
        c += __popcll(__ballot(k));
    }
    if (lane == 0) cnt[i] = c;
}

// side 0: wave per user row; side 1: wave per item column
template <int SIDE>
__global__ __launch_bounds__(64) void k_dense_fill(int n, int m, const double *__restrict__ real, int ldr,
                                                   const uint8_t *__restrict__ known, int ldk,
                                                   const int64_t *__restrict__ ptr, int32_t *__restrict__ idx,
                                                   double *__restrict__ val)
{
    const int row = blockIdx.x, lane = threadIdx.x;
    const int len = SIDE == 0 ? m : n;
    int64_t at = ptr[row];
    for (int t0 = 0; t0 < len; t0 += 64) {
        const int t = t0 + lane;
        const size_t ek = SIDE == 0 ? (size_t)row * ldk + t : (size_t)t * ldk + row;
        const size_t er = SIDE == 0 ? (size_t)row * ldr + t : (size_t)t * ldr + row;
        const bool k = t < len && known[ek];
        const unsigned long long b = __ballot(k);
        if (k) {
            const int64_t o = at + __popcll(b & ((1ull << lane) - 1ull));
            idx[o] = t;
            val[o] = real[er];
        }
        at += __popcll(b);
    }
}

// the row slots of the whole-model launches (k_fit_iter): descending-nnz order, one record each
int upload_desc(apmf_ctx *ctx, apmf_ratings *r, const std::vector<int32_t> &order,
                const std::vector<int64_t> &ptr0, const std::vector<int64_t> &ptr1)
{
    std::vector<RowDesc> desc(order.size());
    for (size_t t = 0; t < order.size(); t++) {
        const int ra = order[t];
        const std::vector<int64_t> &ptr = ra < r->n ? ptr0 : ptr1;
        const int row = ra < r->n ? ra : ra - r->n;
        desc[t].row_all = ra;
        desc[t].beg = ptr[row];
        desc[t].len = (int32_t)(ptr[row + 1] - ptr[row]);
        r->max_len = std::max(r->max_len, desc[t].len);
    }
    if (desc.empty()) return APMF_OK;
    r->h_desc = malloc(desc.size() * sizeof(RowDesc));
    if (!r->h_desc) { apmf_set_error("out of host memory"); return APMF_ENOMEM; }
    memcpy(r->h_desc, desc.data(), desc.size() * sizeof(RowDesc));
    return apmf_h2d(ctx, r->desc, desc.data(), desc.size() * sizeof(RowDesc));
}

} // namespace

extern "C" {

int apmf_ratings_create(apmf_ctx *ctx, const double *h_ratings, int64_t N, int n, int m,
                        apmf_ratings **out)
{
    APMF_REQUIRE(ctx && out, "ctx/out is NULL");
    *out = nullptr;
    APMF_REQUIRE(N >= 0 && n >= 1 && m >= 1, "bad shape N=%lld n=%d m=%d", (long long)N, n, m);
    APMF_REQUIRE(N == 0 || h_ratings, "h_ratings is NULL");
    // host CSR build: stable counting sort by user and by item
    std::vector<int64_t> ptr[2];
    std::vector<int32_t> idx[2];
    std::vector<double> val[2];
    ptr[0].assign((size_t)n + 1, 0);
    ptr[1].assign((size_t)m + 1, 0);
    for (int64_t t = 0; t < N; t++) {
        const double fi = h_ratings[3 * t], fj = h_ratings[3 * t + 1];
        APMF_REQUIRE(fi >= 0 && fi < n && fj >= 0 && fj < m && fi == (double)(int64_t)fi &&
                         fj == (double)(int64_t)fj,
                     "rating %lld has index (%g, %g) outside %d x %d", (long long)t, fi, fj, n, m);
        ptr[0][(size_t)fi + 1]++;
        ptr[1][(size_t)fj + 1]++;
    }
    for (int s = 0; s < 2; s++) {
        for (size_t k = 1; k < ptr[s].size(); k++) ptr[s][k] += ptr[s][k - 1];
        idx[s].resize((size_t)N);
        val[s].resize((size_t)N);
    }
    {
        std::vector<int64_t> cur0(ptr[0].begin(), ptr[0].end() - 1), cur1(ptr[1].begin(), ptr[1].end() - 1);
        for (int64_t t = 0; t < N; t++) {
            const int32_t i = (int32_t)h_ratings[3 * t], j = (int32_t)h_ratings[3 * t + 1];
            const double rv = h_ratings[3 * t + 2];
            int64_t a = cur0[i]++, b = cur1[j]++;
            idx[0][a] = j; val[0][a] = rv;
            idx[1][b] = i; val[1][b] = rv;
        }
    }
    std::vector<int32_t> order((size_t)n + m);
    for (int t = 0; t < n + m; t++) order[t] = t;
    auto nnz = [&](int t) { return t < n ? ptr[0][t + 1] - ptr[0][t] : ptr[1][t - n + 1] - ptr[1][t - n]; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nnz(a) > nnz(b); });

    apmf_ratings *r = new (std::nothrow) apmf_ratings();
    if (!r) { apmf_set_error("out of host memory"); return APMF_ENOMEM; }
    memset(r, 0, sizeof(*r));
    r->N = N; r->n = n; r->m = m;
    for (int t = 0; t < n; t++) if (nnz(t) == 0) r->has_empty[0] = 1;
    for (int t = n; t < n + m; t++) if (nnz(t) == 0) r->has_empty[1] = 1;

    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_ptr0 = al((size_t)(n + 1) * 8), b_ptr1 = al((size_t)(m + 1) * 8);
    const size_t b_idx = al((size_t)N * 4 + 4), b_val = al((size_t)N * 8 + 8), b_ord = al((size_t)(n + m) * 4);
    const size_t b_desc = al((size_t)(n + m) * sizeof(RowDesc));
    const size_t total = b_ptr0 + b_ptr1 + 2 * b_idx + 2 * b_val + b_ord + b_desc;
    APMF_CHECK_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(&r->block, total);
    if (e != hipSuccess) {
        delete r;
        apmf_set_error("hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
        return APMF_ENOMEM;
    }
    char *base = (char *)r->block;
    r->ptr[0] = (int64_t *)base; base += b_ptr0;
    r->ptr[1] = (int64_t *)base; base += b_ptr1;
    r->idx[0] = (int32_t *)base; base += b_idx;
    r->idx[1] = (int32_t *)base; base += b_idx;
    r->val[0] = (double *)base; base += b_val;
    r->val[1] = (double *)base; base += b_val;
    r->order = (int32_t *)base; base += b_ord;
    r->desc = base;
    int rc = APMF_OK;
    for (int s = 0; s < 2 && rc == APMF_OK; s++) {
        rc = apmf_h2d(ctx, r->ptr[s], ptr[s].data(), ptr[s].size() * 8);
        if (rc == APMF_OK && N) rc = apmf_h2d(ctx, r->idx[s], idx[s].data(), (size_t)N * 4);
        if (rc == APMF_OK && N) rc = apmf_h2d(ctx, r->val[s], val[s].data(), (size_t)N * 8);
    }
    if (rc == APMF_OK) rc = apmf_h2d(ctx, r->order, order.data(), order.size() * 4);
    if (rc == APMF_OK) rc = upload_desc(ctx, r, order, ptr[0], ptr[1]);
    if (rc != APMF_OK) { (void)hipFree(r->block); free(r->h_desc); delete r; return rc; }
    *out = r;
    return APMF_OK;
}

int apmf_ratings_create_dense(apmf_ctx *ctx, const double *d_real, int ld_real, const uint8_t *d_known,
                              int ld_known, int n, int m, apmf_ratings **out)
{
    APMF_REQUIRE(ctx && out && d_real && d_known, "NULL operand");
    *out = nullptr;
    APMF_REQUIRE(n >= 1 && m >= 1 && ld_real >= m && ld_known >= m, "bad shape n=%d m=%d", n, m);
    int *d_cnt = nullptr;
    APMF_TRY(apmf_malloc(ctx, (size_t)(n + m) * sizeof(int), (void **)&d_cnt));
    int rc = apmf_memset(ctx, d_cnt, 0, (size_t)(n + m) * sizeof(int));
    std::vector<int> cnt((size_t)n + m);
    if (rc == APMF_OK) {
        hipLaunchKernelGGL(k_dense_count, dim3(n), dim3(64), 0, ctx->stream, n, m, d_known, ld_known, d_cnt);
        rc = apmf_d2h(ctx, cnt.data(), d_cnt, cnt.size() * sizeof(int));
    }
    (void)apmf_free(ctx, d_cnt);
    if (rc != APMF_OK) return rc;
    std::vector<int64_t> ptr0((size_t)n + 1, 0), ptr1((size_t)m + 1, 0);
    for (int i = 0; i < n; i++) ptr0[i + 1] = ptr0[i] + cnt[i];
    for (int j = 0; j < m; j++) ptr1[j + 1] = ptr1[j] + cnt[n + j];
    const int64_t N = ptr0[n];
    std::vector<int32_t> order((size_t)n + m);
    for (int t = 0; t < n + m; t++) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });

    apmf_ratings *r = new (std::nothrow) apmf_ratings();
    if (!r) { apmf_set_error("out of host memory"); return APMF_ENOMEM; }
    memset(r, 0, sizeof(*r));
    r->N = N; r->n = n; r->m = m;
    for (int t = 0; t < n; t++) if (cnt[t] == 0) r->has_empty[0] = 1;
    for (int t = n; t < n + m; t++) if (cnt[t] == 0) r->has_empty[1] = 1;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_ptr0 = al((size_t)(n + 1) * 8), b_ptr1 = al((size_t)(m + 1) * 8);
    const size_t b_idx = al((size_t)N * 4 + 4), b_val = al((size_t)N * 8 + 8), b_ord = al((size_t)(n + m) * 4);
    const size_t b_desc = al((size_t)(n + m) * sizeof(RowDesc));
    const size_t total = b_ptr0 + b_ptr1 + 2 * b_idx + 2 * b_val + b_ord + b_desc;
    APMF_CHECK_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(&r->block, total);
    if (e != hipSuccess) {
        delete r;
        apmf_set_error("hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
        return APMF_ENOMEM;
    }
    char *base = (char *)r->block;
    r->ptr[0] = (int64_t *)base; base += b_ptr0;
    r->ptr[1] = (int64_t *)base; base += b_ptr1;
    r->idx[0] = (int32_t *)base; base += b_idx;
    r->idx[1] = (int32_t *)base; base += b_idx;
    r->val[0] = (double *)base; base += b_val;
    r->val[1] = (double *)base; base += b_val;
    r->order = (int32_t *)base; base += b_ord;
    r->desc = base;
    rc = apmf_h2d(ctx, r->ptr[0], ptr0.data(), ptr0.size() * 8);
    if (rc == APMF_OK) rc = apmf_h2d(ctx, r->ptr[1], ptr1.data(), ptr1.size() * 8);
    if (rc == APMF_OK) rc = apmf_h2d(ctx, r->order, order.data(), order.size() * 4);
    if (rc == APMF_OK) rc = upload_desc(ctx, r, order, ptr0, ptr1);
    if (rc == APMF_OK && N > 0) {
        hipLaunchKernelGGL((k_dense_fill<0>), dim3(n), dim3(64), 0, ctx->stream, n, m, d_real, ld_real, d_known,
                           ld_known, r->ptr[0], r->idx[0], r->val[0]);
        hipLaunchKernelGGL((k_dense_fill<1>), dim3(m), dim3(64), 0, ctx->stream, n, m, d_real, ld_real, d_known,
                           ld_known, r->ptr[1], r->idx[1], r->val[1]);
        if (hipGetLastError() != hipSuccess) { apmf_set_error("kernel launch failed"); rc = APMF_EHIP; }
        if (rc == APMF_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { apmf_set_error("sync failed"); rc = APMF_EHIP; }
    }
    if (rc != APMF_OK) { (void)hipFree(r->block); free(r->h_desc); delete r; return rc; }
    *out = r;
    return APMF_OK;
}

int apmf_ratings_destroy(apmf_ctx *ctx, apmf_ratings *r)
{
    if (!r) return APMF_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (r->block) (void)hipFree(r->block);
    free(r->h_desc);
    delete r;
    return APMF_OK;
}

int apmf_ratings_count(const apmf_ratings *r, int64_t *N, int *n, int *m)
{
    APMF_REQUIRE(r, "ratings is NULL");
    if (N) *N = r->N;
    if (n) *n = r->n;
    if (m) *m = r->m;
    return APMF_OK;
}

int apmf_pmf_gradient(apmf_ctx *ctx, const apmf_ratings *r, const double *d_U, const double *d_V,
                      int d, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                      int subtract_mean, double mean_rating, double *d_gU, double *d_gV)
{
    APMF_TRY(check_common(ctx, r, d, sigma_sq, sigma_u_sq, sigma_v_sq));
    APMF_REQUIRE(d_U && d_V && d_gU && d_gV, "NULL operand");
    const PmfParams p = make_params(r, d, sigma_sq, sigma_u_sq, sigma_v_sq, subtract_mean, mean_rating);
    return launch_grad(ctx, r, p, d_U, d_V, d_gU, d_gV);
}

int apmf_pmf_log_likelihood(apmf_ctx *ctx, const apmf_ratings *r, const double *d_U,
                            const double *d_V, int d, double sigma_sq, double sigma_u_sq,
                            double sigma_v_sq, int subtract_mean, double mean_rating,
                            double *h_ll, double *h_sq_error)
{
    APMF_TRY(check_common(ctx, r, d, sigma_sq, sigma_u_sq, sigma_v_sq));
    APMF_REQUIRE(d_U && d_V, "NULL operand");
    const PmfParams p = make_params(r, d, sigma_sq, sigma_u_sq, sigma_v_sq, subtract_mean, mean_rating);
    const int nparts = ll_parts(r, 0, r->n, 0, r->m);
    void *w;
    APMF_TRY(apmf_ws(ctx, 1, (size_t)nparts * 24 + 128, &w));
    double *d_part = (double *)w;
    double *d_out2 = d_part + (size_t)nparts * 3;
    APMF_TRY(launch_ll(ctx, r, p, sigma_sq, d_U, d_V, d_part, d_out2));
    double h[2];
    APMF_TRY(apmf_d2h(ctx, h, d_out2, sizeof(h)));
    if (h_ll) *h_ll = h[0];
    if (h_sq_error) *h_sq_error = h[1];
    return APMF_OK;
}

int apmf_pmf_gradient_rows(apmf_ctx *ctx, const apmf_ratings *r, int side, int row_lo, int row_hi,
                           const double *d_U, const double *d_V, int d, double sigma_sq,
                           double sigma_u_sq, double sigma_v_sq, int subtract_mean,
                           double mean_rating, double *d_g)
{
    APMF_TRY(check_common(ctx, r, d, sigma_sq, sigma_u_sq, sigma_v_sq));
    APMF_REQUIRE(d_U && d_V && d_g, "NULL operand");
    APMF_REQUIRE(side == 0 || side == 1, "side must be 0 (users) or 1 (items)");
    const int rows = side == 0 ? r->n : r->m;
    APMF_REQUIRE(0 <= row_lo && row_lo <= row_hi && row_hi <= rows, "row range [%d,%d) outside [0,%d)",
                 row_lo, row_hi, rows);
    const PmfParams p = make_params(r, d, sigma_sq, sigma_u_sq, sigma_v_sq, subtract_mean, mean_rating);
    return launch_grad(ctx, r, p, d_U, d_V, side == 0 ? d_g : (double *)nullptr,
                       side == 1 ? d_g : (double *)nullptr, side == 0 ? row_lo : r->n + row_lo,
                       row_hi - row_lo);
}

int apmf_pmf_ll_terms(apmf_ctx *ctx, const apmf_ratings *r, const double *d_U, const double *d_V, int d,
                      int subtract_mean, double mean_rating, int user_lo, int user_hi, int item_lo,
                      int item_hi, double *h_terms3)
{
    APMF_TRY(check_common(ctx, r, d, 1., 1., 1.));
    APMF_REQUIRE(d_U && d_V && h_terms3, "NULL operand");
    APMF_REQUIRE(0 <= user_lo && user_lo <= user_hi && user_hi <= r->n && 0 <= item_lo &&
                     item_lo <= item_hi && item_hi <= r->m, "bad row ranges");
    const PmfParams p = make_params(r, d, 1., 1., 1., subtract_mean, mean_rating);
    const int nparts = ll_parts(r, user_lo, user_hi, item_lo, item_hi);
    void *w;
    APMF_TRY(apmf_ws(ctx, 1, (size_t)nparts * 24 + 128, &w));
    double *d_part = (double *)w;
    double *d_out = d_part + (size_t)nparts * 3;
    APMF_TRY(launch_ll(ctx, r, p, 1., d_U, d_V, d_part, d_out, user_lo, user_hi, item_lo, item_hi));
    double h[4];
    APMF_TRY(apmf_d2h(ctx, h, d_out, sizeof(h)));
    h_terms3[0] = h[1];
    h_terms3[1] = h[2];
    h_terms3[2] = h[3];
    return APMF_OK;
}

int apmf_pmf_ll_terms_dev(apmf_ctx *ctx, const apmf_ratings *r, const double *d_U, const double *d_V, int d,
                          int subtract_mean, double mean_rating, int user_lo, int user_hi, int item_lo,
                          int item_hi, double *d_terms3)
{
    APMF_TRY(check_common(ctx, r, d, 1., 1., 1.));
    APMF_REQUIRE(d_U && d_V && d_terms3, "NULL operand");
    APMF_REQUIRE(0 <= user_lo && user_lo <= user_hi && user_hi <= r->n && 0 <= item_lo &&
                     item_lo <= item_hi && item_hi <= r->m, "bad row ranges");
    const PmfParams p = make_params(r, d, 1., 1., 1., subtract_mean, mean_rating);
    const int nparts = ll_parts(r, user_lo, user_hi, item_lo, item_hi);
    void *w;
    APMF_TRY(apmf_ws(ctx, 1, (size_t)nparts * 24 + 128, &w));
    double *d_part = (double *)w;
    double *d_out = d_part + (size_t)nparts * 3;
    APMF_TRY(launch_ll(ctx, r, p, 1., d_U, d_V, d_part, d_out, user_lo, user_hi, item_lo, item_hi));
    APMF_CHECK_HIP(hipMemcpyAsync(d_terms3, d_out + 1, 3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return APMF_OK;
}

#ifdef APMF_K1_TIMING
int apmf_k1_stamps(unsigned long long *h_out)
{
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_k1_stamps), sizeof(unsigned long long) * 8 * 16384) == hipSuccess ? 0 : 1;
}
#endif

int apmf_pmf_fit_lls(apmf_ctx *ctx, const apmf_ratings *r, double *d_U, double *d_V, int d,
                     double sigma_sq, double sigma_u_sq, double sigma_v_sq, int subtract_mean,
                     double mean_rating, double *h_lr, double *h_old_ll,
                     double min_learning_rate, double stop_thresh, int64_t max_steps,
                     double *h_ll_trace, int64_t max_trace, int64_t *h_steps, int *h_converged)
{
    APMF_TRY(check_common(ctx, r, d, sigma_sq, sigma_u_sq, sigma_v_sq));
    APMF_REQUIRE(d_U && d_V && h_lr && h_steps, "NULL operand");
    APMF_REQUIRE(max_trace == 0 || h_ll_trace, "h_ll_trace is NULL");
    const PmfParams p = make_params(r, d, sigma_sq, sigma_u_sq, sigma_v_sq, subtract_mean, mean_rating);
    const size_t nu = (size_t)r->n * d, nv = (size_t)r->m * d;
    const int nslots = std::max(r->n, r->m);
    const int wpr = wpr_for(r);
    const int nparts = nslots * wpr;
    const int64_t trace_cap = max_trace < 0 ? 0 : max_trace;
    // workspace 0: two gradient pairs | second factor pair | trace ; workspace 1: partials,
    // outputs, the two states
    void *w0, *w1;
    APMF_TRY(apmf_ws(ctx, 0, (3 * (nu + nv) + (size_t)trace_cap) * 8 + 64, &w0));
    const size_t npart_alloc = std::max<size_t>(nparts, 2 * (size_t)FIT_GRID_MAX);
    APMF_TRY(apmf_ws(ctx, 1, npart_alloc * 24 + 128 + 2 * sizeof(FitState) + 64, &w1));
    double *gU0 = (double *)w0, *gV0 = gU0 + nu, *gU1 = gV0 + nv, *gV1 = gU1 + nu, *tU = gV1 + nv,
           *tV = tU + nu, *d_trace = tV + nv;
    double *d_part = (double *)w1, *d_out2 = d_part + npart_alloc * 3;
    FitState *d_state = (FitState *)(d_out2 + 8);   // d_state[0], d_state[1]
    void *pin;
    APMF_TRY(apmf_pinned(ctx, 256, &pin));
    FitState *h_state = (FitState *)pin;

    double old_ll;
    if (h_old_ll && *h_old_ll == *h_old_ll) old_ll = *h_old_ll;
    else {
        // LL of the starting point (fit_lls computes it once before its loop, pmf_cy.pyx:261)
        APMF_TRY(launch_ll(ctx, r, p, sigma_sq, d_U, d_V, d_part, d_out2));
        APMF_CHECK_HIP(hipMemcpyAsync(pin, d_out2, 16, hipMemcpyDeviceToHost, ctx->stream));
        APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        old_ll = ((volatile double *)pin)[0];
    }
    memset(h_state, 0, sizeof(FitState));
    h_state->lr = *h_lr;
    h_state->old_ll = old_ll;
    h_state->min_lr = min_learning_rate;
    h_state->stop_thresh = stop_thresh;
    h_state->max_steps = max_steps;
    h_state->max_trace = trace_cap;
    APMF_CHECK_HIP(hipMemcpyAsync(d_state, h_state, sizeof(FitState), hipMemcpyHostToDevice, ctx->stream));
    APMF_CHECK_HIP(hipMemsetAsync(d_part, 0, 2 * (size_t)FIT_GRID_MAX * 24, ctx->stream));
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // the pinned block is reused for read-backs

    if (max_steps != 0) APMF_TRY(launch_grad(ctx, r, p, d_U, d_V, gU0, gV0));   // gradient at the start
    // a fixed grid (the partial sums, hence the LL bits, must not depend on the device): two
    // blocks per CU of a 256-CU part, each striding over the row slots
    // Block shape: a wave per row (4 rows per block) unless some row is long - the launch lasts as
    // long as its heaviest row, about 0.4 us per 8 ratings of one wave - then 4 waves per row.
    const int fit_wpr = getenv("APMF_K1_WPR") ? atoi(getenv("APMF_K1_WPR")) : 0;
    const int fwpr = (fit_wpr == 1 || fit_wpr == 4) ? fit_wpr : (r->max_len > 192 ? 4 : 1);
    const int frpb = 4 / fwpr;
    // a fixed grid cap (the partial sums, hence the LL bits, must not depend on the device)
    const int g_blocks = std::min((r->n + r->m + frpb - 1) / frpb, FIT_GRID_MAX);
    const int dp = dp_for(d);
    const RowDesc *desc = (const RowDesc *)r->desc;
    long long it = 0;
    // Chunks of <= 16 launches, the state read back after each.  A launch past the end of the search passes the
    // state through untouched, so the NEXT chunk can be enqueued before the read-back of the current one has
    // arrived (the stream never drains: the read-back round trip cost 2 us per launch): done whenever plenty of
    // steps are left; a call for a step or two behaves as before.
    hipEvent_t ev[2] = {nullptr, nullptr};
    auto cleanup = [&]() {
        for (int q = 0; q < 2; q++)
            if (ev[q]) (void)hipEventDestroy(ev[q]);
    };
    auto enqueue_chunk = [&](int chunk, int slot) -> int {
        for (int c = 0; c < chunk; c++, it++) {
            FitState *s_in = d_state + (it & 1), *s_out = d_state + ((it + 1) & 1);
            double *p_in = d_part + (size_t)(it & 1) * FIT_GRID_MAX * 3, *p_out = d_part + (size_t)((it + 1) & 1) * FIT_GRID_MAX * 3;
            ProfScope prof(ctx);
#define FIT_ITER(WPR_, RPB_)                                                                             \
    DISPATCH_DP(dp, hipLaunchKernelGGL((k_fit_iter<DP, WPR_, RPB_, 2>), dim3(g_blocks), dim3(256), 0,    \
                                       ctx->stream, p, s_in, s_out, g_blocks, p_in, p_out, desc, r->idx[0], \
                                       r->val[0], r->idx[1], r->val[1], d_U, d_V, tU, tV, gU0, gV0, gU1, gV1, \
                                       sigma_sq, d_trace))
            if (fwpr == 4) { FIT_ITER(4, 1); } else { FIT_ITER(1, 4); }
#undef FIT_ITER
        }
        APMF_CHECK_LAUNCH();
        APMF_CHECK_HIP(hipMemcpyAsync(h_state + slot, d_state + (it & 1), sizeof(FitState), hipMemcpyDeviceToHost,
                                      ctx->stream));
        if (!ev[slot]) APMF_CHECK_HIP(hipEventCreateWithFlags(&ev[slot], hipEventDisableTiming));
        APMF_CHECK_HIP(hipEventRecord(ev[slot], ctx->stream));
        return APMF_OK;
    };
    FitState last = *h_state;          // the state as the host last saw it
    if (max_steps != 0) {
        int rc = APMF_OK;
        int slot = 0;                  // read-back slot of the oldest chunk in flight
        bool ahead = false;            // a second chunk is in flight behind it
        auto chunk_for = [&](const FitState &st) {
            const long long remaining = max_steps < 0 ? 1 << 20 : max_steps - st.steps;
            return (int)std::min<long long>(std::max<long long>(remaining + 1, 2), 16);
        };
        rc = enqueue_chunk(chunk_for(last), slot);
        while (rc == APMF_OK) {
            // plenty left: keep the stream fed while the read-back travels
            if (!ahead && (max_steps < 0 || max_steps - last.steps > 48)) {
                rc = enqueue_chunk(16, slot ^ 1);
                if (rc != APMF_OK) break;
                ahead = true;
            }
            if (hipEventSynchronize(ev[slot]) != hipSuccess) { rc = APMF_EHIP; apmf_set_error("hipEventSynchronize failed"); break; }
            last = h_state[slot];
            if (last.converged || (max_steps >= 0 && last.steps >= max_steps)) break;
            if (ahead) {
                slot ^= 1;             // the chunk behind becomes the oldest
                ahead = false;
            } else {
                rc = enqueue_chunk(chunk_for(last), slot);
            }
        }
        if (rc != APMF_OK) {
            (void)hipStreamSynchronize(ctx->stream);
            cleanup();
            return rc;
        }
    }
    cleanup();
    const FitState fin = last;         // (not through the pinned block: a read-back of the chunk behind may still land there)
    if (fin.cur) {
        APMF_CHECK_HIP(hipMemcpyAsync(d_U, tU, nu * 8, hipMemcpyDeviceToDevice, ctx->stream));
        APMF_CHECK_HIP(hipMemcpyAsync(d_V, tV, nv * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    const int64_t ntr = std::min<int64_t>(fin.steps, trace_cap);
    if (ntr > 0)
        APMF_CHECK_HIP(hipMemcpyAsync(h_ll_trace, d_trace, (size_t)ntr * 8, hipMemcpyDeviceToHost, ctx->stream));
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_lr = fin.lr;
    if (h_old_ll) *h_old_ll = fin.old_ll;
    *h_steps = fin.steps;
    if (h_converged) *h_converged = fin.converged ? 1 : 0;
    return APMF_OK;
}

namespace {

// per hypothesis: the factors of the accepted point live in the second buffer pair -> copy them back
__global__ __launch_bounds__(256) void k_fit_batch_collect(const int *__restrict__ cur, size_t nu, size_t nv,
                                                           double *__restrict__ U, const double *__restrict__ tU,
                                                           double *__restrict__ V, const double *__restrict__ tV)
{
    const int h = blockIdx.y;
    if (!cur[h]) return;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nu + nv; e += (size_t)gridDim.x * 256) {
        if (e < nu) U[h * nu + e] = tU[h * nu + e];
        else V[h * nv + (e - nu)] = tV[h * nv + (e - nu)];
    }
}

struct DevBlocks {            // device blocks of one call, released together
    apmf_ctx *ctx;
    std::vector<void *> p;
    explicit DevBlocks(apmf_ctx *c) : ctx(c) {}
    ~DevBlocks() { for (void *q : p) (void)apmf_free(ctx, q); }
    int get(size_t bytes, void **out)
    {
        int rc = apmf_malloc(ctx, bytes ? bytes : 8, out);
        if (rc == APMF_OK) p.push_back(*out);
        return rc;
    }
};

} // namespace

// H fit_lls searches in one chain of launches: hypothesis h = the shared container + the rating
// (h_ei[h], h_ej[h], h_ev[h]) appended where add_rating() would put it, started from its own factors
// d_U[h], d_V[h] (updated in place).  The lookahead with refit_lookahead re-fits the MAP estimate of every
// hypothesis first (mn_active_pmf.py:662-665: apmf.do_fit() on the deep copy).  Every hypothesis runs the
// arithmetic of apmf_pmf_fit_lls on a container rebuilt with its rating - same row order, same partial
// sums, same decisions: bit-identical - with its own accept / reject state on the device.
int apmf_pmf_batch_fit_lls(apmf_ctx *ctx, const apmf_ratings *r, int d, int H, const int32_t *h_ei,
                           const int32_t *h_ej, const double *h_ev, double *d_U, double *d_V, double sigma_sq,
                           double sigma_u_sq, double sigma_v_sq, int subtract_mean, const double *h_mean_rating,
                           double learning_rate, double min_learning_rate, double stop_thresh, int64_t max_steps,
                           double *h_ll, int64_t *h_steps, int *h_converged)
{
    APMF_TRY(check_common(ctx, r, d, sigma_sq, sigma_u_sq, sigma_v_sq));
    APMF_REQUIRE(H >= 1 && h_ei && h_ej && h_ev && d_U && d_V, "NULL operand / empty batch");
    APMF_REQUIRE(!subtract_mean || h_mean_rating, "subtract_mean needs the mean rating of every hypothesis");
    APMF_REQUIRE(r->N >= 1 && r->h_desc, "the ratings container is empty");
    const int n = r->n, m = r->m, R = n + m;
    for (int h = 0; h < H; h++)
        APMF_REQUIRE(h_ei[h] >= 0 && h_ei[h] < n && h_ej[h] >= 0 && h_ej[h] < m, "hypothesis %d: entry (%d, %d) outside %d x %d",
                     h, h_ei[h], h_ej[h], n, m);
    const PmfParams p = make_params(r, d, sigma_sq, sigma_u_sq, sigma_v_sq, 0, 0.);
    const size_t nu = (size_t)n * d, nv = (size_t)m * d;
    // what a container rebuilt with one more rating would choose (wpr_for, fit block shape)
    const int64_t rows_min = std::max(1, std::min(n, m));
    const int wpr = ((r->N + 1) / rows_min) >= 48 ? 4 : 1;
    const int nslots = std::max(n, m);
    const int nparts_ll = nslots * wpr;
    const RowDesc *base = (const RowDesc *)r->h_desc;
    std::vector<int32_t> len_of((size_t)R);
    std::vector<int64_t> beg_of((size_t)R);
    for (int t = 0; t < R; t++) { len_of[base[t].row_all] = base[t].len; beg_of[base[t].row_all] = base[t].beg; }
    // per-hypothesis row descriptors: the stable descending-nnz order of the rebuilt container
    std::vector<RowDesc> desc((size_t)H * R);
    std::vector<int> fwpr_of((size_t)H);
    const int fit_wpr = getenv("APMF_K1_WPR") ? atoi(getenv("APMF_K1_WPR")) : 0;
    auto before = [](const RowDesc &a, const RowDesc &b) { return a.len != b.len ? a.len > b.len : a.row_all < b.row_all; };
    for (int h = 0; h < H; h++) {
        const int a = h_ei[h], b = n + h_ej[h];
        RowDesc ins[2] = {{a, len_of[a] + 1, beg_of[a]}, {b, len_of[b] + 1, beg_of[b]}};
        if (before(ins[1], ins[0])) std::swap(ins[0], ins[1]);
        RowDesc *out = desc.data() + (size_t)h * R;
        int q = 0, w = 0;
        for (int t = 0; t < R; t++) {
            const RowDesc &e = base[t];
            if (e.row_all == a || e.row_all == b) continue;
            while (q < 2 && before(ins[q], e)) out[w++] = ins[q++];
            out[w++] = e;
        }
        while (q < 2) out[w++] = ins[q++];
        const int max_len = std::max(r->max_len, std::max(len_of[a], len_of[b]) + 1);
        fwpr_of[h] = (fit_wpr == 1 || fit_wpr == 4) ? fit_wpr : (max_len > 192 ? 4 : 1);
    }
    DevBlocks dev(ctx);
    RowDesc *d_desc;
    int32_t *d_ei, *d_ej;
    double *d_ev, *d_mean = nullptr, *d_work, *d_part, *d_part_ll, *d_ll0;
    FitState *d_state;
    int *d_sel, *d_cur;
    APMF_TRY(dev.get(desc.size() * sizeof(RowDesc), (void **)&d_desc));
    APMF_TRY(dev.get((size_t)H * 4, (void **)&d_ei));
    APMF_TRY(dev.get((size_t)H * 4, (void **)&d_ej));
    APMF_TRY(dev.get((size_t)H * 8, (void **)&d_ev));
    if (subtract_mean) APMF_TRY(dev.get((size_t)H * 8, (void **)&d_mean));
    APMF_TRY(dev.get((size_t)H * 6 * (nu + nv) / 2 * 8 + 64, (void **)&d_work));       // gU0 gV0 gU1 gV1 tU tV
    APMF_TRY(dev.get((size_t)2 * H * 3 * FIT_GRID_MAX * 8, (void **)&d_part));
    APMF_TRY(dev.get((size_t)H * nparts_ll * 24, (void **)&d_part_ll));
    APMF_TRY(dev.get((size_t)H * 8, (void **)&d_ll0));
    APMF_TRY(dev.get((size_t)2 * H * sizeof(FitState), (void **)&d_state));
    APMF_TRY(dev.get((size_t)H * 4, (void **)&d_sel));
    APMF_TRY(dev.get((size_t)H * 4, (void **)&d_cur));
    double *gU0 = d_work, *gV0 = gU0 + H * nu, *gU1 = gV0 + H * nv, *gV1 = gU1 + H * nu, *tU = gV1 + H * nv,
           *tV = tU + H * nu;
    APMF_TRY(apmf_h2d(ctx, d_desc, desc.data(), desc.size() * sizeof(RowDesc)));
    APMF_TRY(apmf_h2d(ctx, d_ei, h_ei, (size_t)H * 4));
    APMF_TRY(apmf_h2d(ctx, d_ej, h_ej, (size_t)H * 4));
    APMF_TRY(apmf_h2d(ctx, d_ev, h_ev, (size_t)H * 8));
    if (subtract_mean) APMF_TRY(apmf_h2d(ctx, d_mean, h_mean_rating, (size_t)H * 8));
    FitBatch B;
    B.sel = d_sel; B.ei = d_ei; B.ej = d_ej; B.ev = d_ev; B.mean_add = d_mean; B.nu = nu; B.nv = nv; B.nrows = R;
    std::vector<int> sel((size_t)H);
    for (int h = 0; h < H; h++) sel[h] = h;
    APMF_TRY(apmf_h2d(ctx, d_sel, sel.data(), (size_t)H * 4));
    const int dp = dp_for(d);
    // LL of every starting point (launch_ll's arithmetic) and the gradient there (launch_grad's)
    APMF_CHECK_HIP(hipMemsetAsync(d_part_ll, 0, (size_t)H * nparts_ll * 24, ctx->stream));
    {
        const int blocks = (nslots + (4 / wpr) - 1) / (4 / wpr);
        DISPATCH_DP_WPR(dp, wpr,
                        hipLaunchKernelGGL((k_sqerr_batch<DP, WPR>), dim3(blocks, H), dim3(256), 0, ctx->stream, p, B,
                                           r->ptr[0], r->idx[0], r->val[0], d_U, d_V, d_part_ll, nslots));
        hipLaunchKernelGGL(k_ll_finish_batch, dim3(H), dim3(256), 0, ctx->stream, nparts_ll, d_part_ll, sigma_sq,
                           p.sig_u, p.sig_v, d_sel, d_ll0);
        APMF_CHECK_LAUNCH();
    }
    std::vector<double> ll0((size_t)H);
    APMF_TRY(apmf_d2h(ctx, ll0.data(), d_ll0, (size_t)H * 8));
    std::vector<FitState> st((size_t)H), fin((size_t)H);
    for (int h = 0; h < H; h++) {
        memset(&st[h], 0, sizeof(FitState));
        st[h].lr = learning_rate;
        st[h].old_ll = ll0[h];
        st[h].min_lr = min_learning_rate;
        st[h].stop_thresh = stop_thresh;
        st[h].max_steps = max_steps;
        st[h].max_trace = 0;
        fin[h] = st[h];
    }
    APMF_TRY(apmf_h2d(ctx, d_state, st.data(), (size_t)H * sizeof(FitState)));
    APMF_CHECK_HIP(hipMemsetAsync(d_part, 0, (size_t)2 * H * 3 * FIT_GRID_MAX * 8, ctx->stream));
    if (max_steps != 0) {
        const int blocks = (R + (4 / wpr) - 1) / (4 / wpr);
        DISPATCH_DP_WPR(dp, wpr,
                        hipLaunchKernelGGL((k_grad_batch<DP, WPR>), dim3(blocks, H), dim3(256), 0, ctx->stream, p, B,
                                           r->ptr[0], r->idx[0], r->val[0], r->ptr[1], r->idx[1], r->val[1], d_U, d_V,
                                           gU0, gV0));
        APMF_CHECK_LAUNCH();
    }
    // the search: all live hypotheses advance by one trial per launch; the list of live ones is compacted
    // whenever the states are read (a finished hypothesis' state is final: its launches pass it through)
    std::vector<char> live((size_t)H, max_steps != 0 ? 1 : 0);
    long long it = 0;
    for (;;) {
        int done_all = 1;
        for (int pass = 0; pass < 2; pass++) {            // the two block shapes a batch may need
            const int fwpr = pass ? 4 : 1;
            sel.clear();
            for (int h = 0; h < H; h++) if (live[h] && fwpr_of[h] == fwpr) sel.push_back(h);
            if (sel.empty()) continue;
            done_all = 0;
            int *d_sel_p = d_sel + (pass ? H - (int)sel.size() : 0);     // the two lists share the array
            APMF_TRY(apmf_h2d(ctx, d_sel_p, sel.data(), sel.size() * 4));
            FitBatch Bp = B;
            Bp.sel = d_sel_p;
            const int frpb = 4 / fwpr;
            const int g_blocks = std::min((R + frpb - 1) / frpb, FIT_GRID_MAX);
            long long jt = it;
            for (int c = 0; c < 16; c++, jt++) {
                FitState *s_in = d_state + (size_t)(jt & 1) * H, *s_out = d_state + (size_t)((jt + 1) & 1) * H;
                double *p_in = d_part + (size_t)(jt & 1) * H * 3 * FIT_GRID_MAX,
                       *p_out = d_part + (size_t)((jt + 1) & 1) * H * 3 * FIT_GRID_MAX;
#define FIT_ITER_B(WPR_, RPB_)                                                                                   \
    DISPATCH_DP(dp, hipLaunchKernelGGL((k_fit_iter<DP, WPR_, RPB_, 2, true>), dim3(g_blocks, (unsigned)sel.size()), \
                                       dim3(256), 0, ctx->stream, p, s_in, s_out, g_blocks, p_in, p_out, d_desc,  \
                                       r->idx[0], r->val[0], r->idx[1], r->val[1], d_U, d_V, tU, tV, gU0, gV0,   \
                                       gU1, gV1, sigma_sq, (double *)nullptr, Bp))
                if (fwpr == 4) { FIT_ITER_B(4, 1); } else { FIT_ITER_B(1, 4); }
#undef FIT_ITER_B
            }
            APMF_CHECK_LAUNCH();
        }
        if (done_all) break;
        it += 16;                                         // an even number: the parity of the buffers is that of `it`
        APMF_TRY(apmf_d2h(ctx, st.data(), d_state + (size_t)(it & 1) * H, (size_t)H * sizeof(FitState)));
        for (int h = 0; h < H; h++) {
            if (!live[h]) continue;
            fin[h] = st[h];
            if (st[h].converged || (max_steps >= 0 && st[h].steps >= max_steps)) live[h] = 0;
        }
    }
    std::vector<int> cur((size_t)H);
    int any = 0;
    for (int h = 0; h < H; h++) { cur[h] = fin[h].cur; any |= cur[h]; }
    if (any) {
        APMF_TRY(apmf_h2d(ctx, d_cur, cur.data(), (size_t)H * 4));
        const unsigned gx = (unsigned)std::min<size_t>((nu + nv + 255) / 256, 64);
        hipLaunchKernelGGL(k_fit_batch_collect, dim3(gx, H), dim3(256), 0, ctx->stream, d_cur, nu, nv, d_U, tU, d_V, tV);
        APMF_CHECK_LAUNCH();
    }
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (int h = 0; h < H; h++) {
        if (h_ll) h_ll[h] = fin[h].old_ll;
        if (h_steps) h_steps[h] = fin[h].steps;
        if (h_converged) h_converged[h] = fin[h].converged ? 1 : 0;
    }
    return APMF_OK;
}

int apmf_pmf_minibatch_epoch(apmf_ctx *ctx, const double *h_shuffled_ratings, int64_t N, int n,
                             int m, int64_t batch_size, double lr, double momentum, double *d_U,
                             double *d_V, double *d_uinc, double *d_vinc, int d, double sigma_sq,
                             double sigma_u_sq, double sigma_v_sq, int subtract_mean,
                             double mean_rating)
{
    APMF_REQUIRE(ctx && h_shuffled_ratings && d_U && d_V && d_uinc && d_vinc, "NULL operand");
    APMF_REQUIRE(N >= 1 && n >= 1 && m >= 1 && batch_size >= 1, "bad shape");
    APMF_REQUIRE(d >= 1 && d <= APMF_MAX_D, "latent_d=%d unsupported (1..%d)", d, APMF_MAX_D);
    APMF_REQUIRE(sigma_sq != 0. && sigma_u_sq != 0. && sigma_v_sq != 0., "zero variance");
    // Host: per batch, a stable CSR over the rows (users first, then items) that occur in it.
    const int64_t nb = (N + batch_size - 1) / batch_size;
    std::vector<int32_t> rows; rows.reserve((size_t)2 * N);
    std::vector<int64_t> ptr; ptr.reserve((size_t)2 * N + nb);
    std::vector<int32_t> idx((size_t)2 * N);
    std::vector<double> val((size_t)2 * N);
    std::vector<int64_t> b_rows_off((size_t)nb + 1), b_ptr_off((size_t)nb + 1);
    std::vector<int64_t> cnt((size_t)n + m), pos((size_t)n + m);
    int64_t fill = 0;
    for (int64_t b = 0; b < nb; b++) {
        const int64_t lo = b * batch_size, hi = std::min(lo + batch_size, N);
        b_rows_off[b] = (int64_t)rows.size();
        b_ptr_off[b] = (int64_t)ptr.size();
        std::vector<int32_t> present;
        for (int64_t t = lo; t < hi; t++) {
            const double fi = h_shuffled_ratings[3 * t], fj = h_shuffled_ratings[3 * t + 1];
            APMF_REQUIRE(fi >= 0 && fi < n && fj >= 0 && fj < m, "rating index out of range");
            const int32_t i = (int32_t)fi, j = n + (int32_t)fj;
            if (cnt[i]++ == 0) present.push_back(i);
            if (cnt[j]++ == 0) present.push_back(j);
        }
        std::sort(present.begin(), present.end());
        for (int32_t rw : present) {
            rows.push_back(rw);
            ptr.push_back(fill);
            pos[rw] = fill;
            fill += cnt[rw];
        }
        ptr.push_back(fill);
        for (int64_t t = lo; t < hi; t++) {
            const int32_t i = (int32_t)h_shuffled_ratings[3 * t], j = (int32_t)h_shuffled_ratings[3 * t + 1];
            const double rv = h_shuffled_ratings[3 * t + 2];
            idx[pos[i]] = j; val[pos[i]++] = rv;
            idx[pos[n + j]] = i; val[pos[n + j]++] = rv;
        }
        for (int32_t rw : present) cnt[rw] = 0;
    }
    b_rows_off[nb] = (int64_t)rows.size();
    b_ptr_off[nb] = (int64_t)ptr.size();

    const size_t nu = (size_t)n * d, nv = (size_t)m * d;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_rows = al(rows.size() * 4), b_ptr = al(ptr.size() * 8), b_idx = al(idx.size() * 4),
                 b_val = al(val.size() * 8), b_g = al((nu + nv) * 8);
    void *w;
    APMF_TRY(apmf_ws(ctx, 3, b_rows + b_ptr + b_idx + b_val + b_g, &w));
    char *base = (char *)w;
    int32_t *d_rows = (int32_t *)base; base += b_rows;
    int64_t *d_ptr = (int64_t *)base; base += b_ptr;
    int32_t *d_idx = (int32_t *)base; base += b_idx;
    double *d_val = (double *)base; base += b_val;
    double *gacc = (double *)base;
    APMF_TRY(apmf_h2d(ctx, d_rows, rows.data(), rows.size() * 4));
    APMF_TRY(apmf_h2d(ctx, d_ptr, ptr.data(), ptr.size() * 8));
    APMF_TRY(apmf_h2d(ctx, d_idx, idx.data(), idx.size() * 4));
    APMF_TRY(apmf_h2d(ctx, d_val, val.data(), val.size() * 8));
    APMF_CHECK_HIP(hipMemsetAsync(gacc, 0, (nu + nv) * 8, ctx->stream));

    PmfParams p;
    p.n = n; p.m = m; p.d = d; p.sig = sigma_sq; p.sig_u = sigma_u_sq; p.sig_v = sigma_v_sq;
    p.mean_add = subtract_mean ? mean_rating : 0.;
    const int up_blocks = (int)std::min<size_t>((nu + nv + 255) / 256, 2048);
    momentum = (double)(float)momentum;  // `float momentum` in the reference, pmf_cy.pyx:309
    for (int64_t b = 0; b < nb; b++) {
        const int64_t lo = b * batch_size, hi = std::min(lo + batch_size, N);
        const int npres = (int)(b_rows_off[b + 1] - b_rows_off[b]);
        const int blocks = (npres + 3) / 4;
        const int32_t *rws = d_rows + b_rows_off[b];
        const int64_t *pp = d_ptr + b_ptr_off[b];
        DISPATCH_DP(dp_for(d),
                    hipLaunchKernelGGL((k_pmf_grad_rows<DP>), dim3(blocks), dim3(256), 0, ctx->stream, p,
                                       npres, rws, pp, d_idx, d_val, d_U, d_V, gacc, gacc + nu));
        APMF_CHECK_LAUNCH();
        // `lr / n` is evaluated in C float in the reference (cdef float lr; int n), pmf_cy.pyx:309,339
        const double scale = (double)((float)lr / (float)(hi - lo));
        hipLaunchKernelGGL(k_momentum, dim3(up_blocks), dim3(256), 0, ctx->stream, nu, nv, momentum, scale,
                           sigma_u_sq, sigma_v_sq, d_U, d_V, d_uinc, d_vinc, gacc, gacc + nu);
        APMF_CHECK_LAUNCH();
    }
    APMF_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return APMF_OK;
}

int apmf_pmf_predicted_matrix(apmf_ctx *ctx, const double *d_U, const double *d_V, int n, int m,
                              int d, double add, double *d_out, int ld)
{
    APMF_REQUIRE(ctx && d_U && d_V && d_out, "NULL operand");
    APMF_REQUIRE(n >= 1 && m >= 1 && d >= 1 && ld >= m, "bad shape");
    hipLaunchKernelGGL(k_predict, dim3((ld + 255) / 256, (n + 15) / 16), dim3(256), 0, ctx->stream, n, m, d,
                       ld, add, d_U, d_V, d_out);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

} // extern "C"
