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
//   k_full_sweep_q<D>  (9 <= d <= 16: the headline rank 10 among them) four lanes per candidate, the cross block
//                 contracted straight from LDS: see the comment at the kernel.
// d > 16: k_full_grid, one thread per candidate straight from global memory (the first version,
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

// ---- the sweep for 9 <= d <= 16: FOUR lanes per candidate, the cross block read from LDS -------------------
// At d = 10 - the rank of the headline metric - a lane cannot hold its own 10 x 10 cross block, its item's
// covariance block and the staging registers (620 VGPRs), and 64 items x d^2 doubles per wave do not fit LDS at a
// useful occupancy.  So: one wave = 16 adjacent items x ROWS user rows, lane = (q, jj), q = lane >> 4, jj = lane & 15;
// the four lanes of an item split every contraction (k = q, q + 4, ...) and are added with two xor-shuffles.
//   * The cross block of (row i, 16 items) is D row segments of 16 D CONTIGUOUS doubles: coalesced 512-byte
//     nontemporal loads, those of row r + 1 in flight while row r is contracted, staged through LDS at an odd
//     item stride (conflict-free) - and NOT copied to registers: tr(C C) reads C[k][l] and C[l][k] straight from
//     LDS, C a likewise (2 d^2 eight-byte LDS reads per candidate: a fifth of the LDS bandwidth at the HBM rate).
//   * The terms without the cross block, <A + a a', Bm> + b'A b, are ONE inner product of length d (d + 1)
//     between a per-user record X_i = [w (A + a a') | w A]  (upper triangles, w = 2 off the diagonal) and a
//     per-item record Y_j = [Bm | b b'] packed by k_full_prep_q (Y transposed: lane-contiguous over j); a lane keeps
//     its quarter of Y_j in registers for all its rows and reads X_i from an LDS copy of the row's record.
// Per candidate 8 d^2 bytes streamed + 8 written, as before.
template <int D>
struct FullQ {
    static constexpr int TS = D * (D + 1) / 2;         // entries of one symmetric pack
    static constexpr int T = 2 * TS;                   // inner-product length of the cross-block-free terms
    static constexpr int REC = T + D;                  // record: [.. | a] (users) / [.. | b] (items)
    static constexpr int TQ = (T + 3) / 4;             // per lane
    static constexpr int DQ = (D + 3) / 4;
    static constexpr int DP = D | 1;                   // odd item stride in LDS
    static constexpr int NIT = (16 * D + 63) / 64;     // 64-lane load iterations per row segment
};

__global__ __launch_bounds__(256) void k_full_prep_q(int n, int m, int d, int ld, const double *__restrict__ mean,
                                                     const double *__restrict__ cov, double *__restrict__ xrec,
                                                     double *__restrict__ yt)
{
    const size_t K = (size_t)(n + m) * d;
    const int ts = d * (d + 1) / 2, rec = 2 * ts + d;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) {
        const int i = t;
        double *x = xrec + (size_t)i * rec;
        int e = 0;
        for (int k = 0; k < d; k++) {
            const double ak = mean[(size_t)i * d + k];
            x[2 * ts + k] = ak;
            for (int l = k; l < d; l++, e++) {
                const double A = cov[((size_t)i * d + k) * K + (size_t)i * d + l];
                const double w = k == l ? 1. : 2.;
                x[e] = w * (A + ak * mean[(size_t)i * d + l]);
                x[ts + e] = w * A;
            }
        }
    } else if (t < n + ld) {
        const int j = t - n;
        if (j < m) {
            const size_t vb = (size_t)n * d + (size_t)j * d;
            int e = 0;
            for (int k = 0; k < d; k++) {
                const double bk = mean[vb + k];
                yt[(size_t)(2 * ts + k) * ld + j] = bk;
                for (int l = k; l < d; l++, e++) {
                    yt[(size_t)e * ld + j] = cov[(vb + k) * K + vb + l];
                    yt[(size_t)(ts + e) * ld + j] = bk * mean[vb + l];
                }
            }
        } else {
            for (int e = 0; e < rec; e++) yt[(size_t)e * ld + j] = 0.;
        }
    }
}

// The four waves of a block share nothing (each stages its own 16 items in its own LDS region): what orders a
// wave's LDS writes before its lanes' reads of each other's data is the in-order LDS queue of the wave itself - a
// block barrier would only make the waves march in step and idle together on every load.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int D, int ROWS>
__global__ __launch_bounds__(256) void k_full_sweep_q(int n, int m, int ld, const double *__restrict__ cov,
                                                      const double *__restrict__ xrec, const double *__restrict__ yt,
                                                      const uint8_t *__restrict__ mask, double *__restrict__ o_var,
                                                      double *__restrict__ o_mean, int want, int want_sum,
                                                      double *__restrict__ part_sum, double *__restrict__ part_val,
                                                      long long *__restrict__ part_idx)
{
    using Q = FullQ<D>;
    constexpr int DP = Q::DP, NIT = Q::NIT, T = Q::T, REC = Q::REC, TQ = Q::TQ, DQ = Q::DQ;
    extern __shared__ __attribute__((aligned(16))) double q_lds[];
    const int lane = threadIdx.x & 63, w = wave_id_uniform();
    double *sC = q_lds + (size_t)w * (D * 16 * DP + REC);      // [D][16 * DP]
    double *sX = sC + D * 16 * DP;                              // [REC]
    const int q = lane >> 4, jj = lane & 15;
    const int j0 = (blockIdx.x * 4 + w) * 16, j = j0 + jj;
    const size_t K = (size_t)(n + m) * D;
    const bool jin = j < m, jst = j < ld;
    const int span = max(0, min(16, m - j0)) * D;              // valid doubles of a row segment
    // the lane's quarter of the item record: entries q, q + 4, ... (zero beyond m: yt is padded to ld)
    const int jc = jst ? j : 0;
    double yq[TQ], bq[DQ];
#pragma unroll
    for (int s = 0; s < TQ; s++) yq[s] = (4 * s + q < T && jst) ? yt[(size_t)(4 * s + q) * ld + jc] : 0.;
#pragma unroll
    for (int s = 0; s < DQ; s++) bq[s] = (4 * s + q < D && jst) ? yt[(size_t)(T + 4 * s + q) * ld + jc] : 0.;
    double s_acc = 0., bv = 0.;
    long long bi = -1;
    double st[D][NIT];
    double xr[(REC + 63) / 64];
    const int i_first = blockIdx.y * ROWS;
#pragma unroll
    for (int it = 0; it < (REC + 63) / 64; it++) xr[it] = 0.;
    auto fetch = [&](int i) {
        if (i < n && span > 0) {        // (a group of padding columns only has nothing to stage: zeros are written)
#pragma unroll
            for (int k = 0; k < D; k++) {
                const double *src = cov + ((size_t)i * D + k) * K + (size_t)n * D + (size_t)j0 * D;
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int e = it * 64 + lane;
                    // (clamped address: the segment of the grid's last items may be short - masked below;
                    // span > 0, so element 0 of the segment is inside the matrix)
                    st[k][it] = __builtin_nontemporal_load(src + (e < span ? e : 0));
                }
            }
#pragma unroll
            for (int it = 0; it < (REC + 63) / 64; it++) {
                const int e = it * 64 + lane;
                xr[it] = xrec[(size_t)i * REC + (e < REC ? e : 0)];
            }
        }
    };
    fetch(i_first);
    for (int r = 0; r < ROWS; r++) {
        const int i = i_first + r;
        const bool row_ok = i < n;                  // wave-uniform
        wave_lds_fence();                           // the previous row's reads of the tile are over
        if (row_ok) {
#pragma unroll
            for (int k = 0; k < D; k++)
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int e = it * 64 + lane;
                    if (e < 16 * D) {
                        const int ej = e / D, el = e - ej * D;
                        sC[k * 16 * DP + ej * DP + el] = e < span ? st[k][it] : 0.;
                    }
                }
#pragma unroll
            for (int it = 0; it < (REC + 63) / 64; it++) {
                const int e = it * 64 + lane;
                if (e < REC) sX[e] = xr[it];
            }
        }
        wave_lds_fence();
        if (r + 1 < ROWS) fetch(i + 1);
        if (!row_ok) continue;
        const double *cj = sC + jj * DP;            // this item's block: C[k][l] = cj[k * 16 * DP + l]
        double g = 0., ab = 0., trCC = 0., bCa = 0., trC = 0.;
#pragma unroll
        for (int s = 0; s < TQ; s++) g += (4 * s + q < T ? sX[4 * s + q] : 0.) * yq[s];
#pragma unroll
        for (int s = 0; s < DQ; s++) {
            const int k = 4 * s + q;
            if (k < D) {                            // (lanes of the last quarter idle when 4 does not divide D)
                double Ca = 0., cc = 0.;
#pragma unroll
                for (int l = 0; l < D; l++) {
                    const double ckl = cj[k * 16 * DP + l];
                    Ca += ckl * sX[T + l];
                    cc += ckl * cj[l * 16 * DP + k];
                }
                bCa += bq[s] * Ca;
                trCC += cc;
                trC += cj[k * 16 * DP + k];
                ab += sX[T + k] * bq[s];
            }
        }
        double var = g + trCC + 2. * bCa, mn = ab + trC;
        var += __shfl_xor(var, 16, 64);
        var += __shfl_xor(var, 32, 64);
        mn += __shfl_xor(mn, 16, 64);
        mn += __shfl_xor(mn, 32, 64);
        if (q == 0 && jst) {
            if (o_var) o_var[(size_t)i * ld + j] = jin ? var : 0.;
            if (o_mean) o_mean[(size_t)i * ld + j] = jin ? mn : 0.;
        }
        if (want || want_sum) {
            const bool cand = q == 0 && jin && (!mask || mask[(size_t)i * ld + j]);
            if (cand) {
                s_acc += var;
                const long long idx = (long long)i * m + j;
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

// user rows per wave (the item record stays in registers for all of them): as many as still leave >= 2 blocks per CU
static inline int full_q_rows(int n, int ld)
{
    const size_t col_blocks = (size_t)(ld / 16 + 3) / 4;
    for (int rows = 16; rows > 4; rows >>= 1)
        if (col_blocks * ((n + rows - 1) / rows) >= 512) return rows;
    return 4;
}

template <int D, int ROWS>
int launch_full_q_rows(hipStream_t st, dim3 grid, int n, int m, int ld, const double *cov, const double *xrec,
                       const double *yt, const uint8_t *mask, double *o_var, double *o_mean, int want, int want_sum,
                       double *ps, double *pv, long long *pi)
{
    using Q = FullQ<D>;
    const size_t lds = 4 * (size_t)(D * 16 * Q::DP + Q::REC) * sizeof(double);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)k_full_sweep_q<D, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return 1;
    hipLaunchKernelGGL((k_full_sweep_q<D, ROWS>), grid, dim3(256), lds, st, n, m, ld, cov, xrec, yt, mask, o_var, o_mean,
                       want, want_sum, ps, pv, pi);
    return 0;
}

template <int D>
int launch_full_q(hipStream_t st, dim3 grid, int rows, int n, int m, int ld, const double *cov, const double *xrec,
                  const double *yt, const uint8_t *mask, double *o_var, double *o_mean, int want, int want_sum, double *ps,
                  double *pv, long long *pi)
{
    if (rows == 16) return launch_full_q_rows<D, 16>(st, grid, n, m, ld, cov, xrec, yt, mask, o_var, o_mean, want, want_sum, ps, pv, pi);
    if (rows == 8) return launch_full_q_rows<D, 8>(st, grid, n, m, ld, cov, xrec, yt, mask, o_var, o_mean, want, want_sum, ps, pv, pi);
    return launch_full_q_rows<D, 4>(st, grid, n, m, ld, cov, xrec, yt, mask, o_var, o_mean, want, want_sum, ps, pv, pi);
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
    bool quad = d >= 9 && d <= 16;                // four lanes per candidate, 16 items x qrows rows per wave
    const int qrows = full_q_rows(n, ld);
    if (const char *e = getenv("APMF_K2FULL_KERNEL"))
        if (!strcmp(e, "general")) { rows_t = 0; quad = false; }
    dim3 grid = rows_t ? dim3((ld + 63) / 64, (n + 4 * rows_t - 1) / (4 * rows_t))
                       : quad ? dim3((ld / 16 + 3) / 4, (n + qrows - 1) / qrows) : dim3((m + 255) / 256, n);
    const size_t nblk = (size_t)grid.x * grid.y;
    const size_t dd = (size_t)d * d;
    const size_t qrec = (size_t)d * (d + 1) + d;
    const size_t pack_doubles = rows_t ? (size_t)n * (2 * dd + d) + (dd + d) * (size_t)ld
                                       : quad ? qrec * ((size_t)n + ld) : 0;
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
    } else if (quad) {
        double *xrec = (double *)(part_idx + nblk);
        double *yt = xrec + (size_t)n * qrec;
        hipLaunchKernelGGL(k_full_prep_q, dim3((n + ld + 255) / 256), dim3(256), 0, ctx->stream, n, m, d, ld, d_mean,
                           d_cov, xrec, yt);
        ProfScope prof(ctx);
        const int ws = d_sum != nullptr;
        int bad = 0;
        switch (d) {
#define FULL_Q(DV) case DV: bad = launch_full_q<DV>(ctx->stream, grid, qrows, n, m, ld, d_cov, xrec, yt, d_mask, d_var, d_pmean, want, ws, part_sum, part_val, part_idx); break;
        FULL_Q(9) FULL_Q(10) FULL_Q(11) FULL_Q(12) FULL_Q(13) FULL_Q(14) FULL_Q(15) FULL_Q(16)
#undef FULL_Q
        }
        APMF_REQUIRE(!bad, "hipFuncSetAttribute(dynamic LDS) failed");
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
