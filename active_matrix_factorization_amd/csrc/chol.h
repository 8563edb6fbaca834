// chol.h - blocked Cholesky of our own, shared by dense.hip (single matrices) and k6_batch.hip
// (H hypotheses per launch).  gfx950, float64.
//
// rocSOLVER's potrf is not bit-reproducible when the GPU is shared (another stream or process
// running): the same input gave last-place different factors, and the accept / reject searches
// built on it amplify that into different trajectories.  This one is a plain right-looking
// blocked factorisation with fixed summation orders and no atomics:
//   per panel of CH_NB columns: factor the diagonal block and invert it (one wave, everything
//   in registers), multiply the rows below it by that inverse (64 x 64 x 64 products in LDS),
//   update the trailing lower triangle (64 x 64 tiles, the panel staged through LDS).
// A is row-major; only its lower triangle (i >= j) is read, and L overwrites it.
//
// Every kernel works on a LIST of independent matrices (gridDim.y = number of jobs): the
// factorisation is a latency-bound chain of small launches, so any number of matrices that fit
// the machine cost what one does.  A job's arithmetic never depends on the other jobs of the
// launch: the factor of a matrix is bit-identical whether it is factored alone or in a batch.
#pragma once
#include <stdlib.h>
#include <atomic>
#include "common.h"

// every launch of this header goes through APMF_LAUNCH so that a translation unit can count them
// (k6_batch.hip: launches per hypothesis, a tracked number of the lookahead)
#ifndef APMF_LAUNCH
#define APMF_LAUNCH hipLaunchKernelGGL
#endif

constexpr int CH_NB = 64;
constexpr int CH_FUSE_JOBS = 32;   // up to this many matrices per launch: update + next diagonal block in one launch

namespace {   // one private copy of the kernels per translation unit that includes this header

struct ChJob {
    double *A;      // N x N row-major, factored in place (lower triangle); NULL: the job is not to run - its blocks
                    // leave at once (a non-zero info word says the same, but only after the tile loads are in flight)
    int *info;      // 0, or 1-based index of the first non-positive pivot (must be zeroed first;
                    // a non-zero value on entry skips the job)
    double *linv;   // inverse of the diagonal block of the current panel: CH_NB^2 doubles ...
    int linv_keep;  // ... 0: one slot reused by every panel; 1: slot p for panel p (all kept)
    // Block-column route only (chol_route_by_column): when `src` is set, the matrix to factor is NOT in A yet - every
    // tile is READ from src (lower triangle, same pitch) the first time the factorisation touches it, and the factor is
    // written to A as always: the caller saves the pass that would copy src into A (K6: the trial point is written
    // once, and both of its factorisations start from there).  shifted: the diagonal is taken as
    // v - (eig + 1e-10 max(|v|, |eig|)) - the certificate's matrix "src - (min_eig + delta) I".
    int shifted;
    const double *src;
    double eig;
};

// the value the factorisation starts from at (i, j) of a job with `src`
__device__ __forceinline__ double ch_first(const ChJob &job, double v, bool on_diag)
{
    return (job.shifted && on_diag) ? v - (job.eig + 1e-10 * fmax(fabs(v), fabs(job.eig))) : v;
}

// up to two jobs by value (the single-matrix callers), or a device array of jobs
struct ChJobs {
    ChJob j[2];
    const ChJob *list;
};

__device__ __forceinline__ ChJob ch_job(const ChJobs &J, int y) { return J.list ? J.list[y] : J.j[y]; }

__device__ __forceinline__ double *ch_linv(const ChJob &job, int k)
{
    return job.linv + (job.linv_keep ? (size_t)(k / CH_NB) * CH_NB * CH_NB : 0);
}

// XCD-aware (tile, job) of a block of a grid (tiles, jobs).  Workgroups are dealt round-robin over the 8 XCDs, each
// with an L2 of its own (blocks b and b + 8 share one): with the plain mapping the tiles of ONE matrix - which read
// the same panels - land on eight different L2s and every one of them fetches those panels again.  Here the blocks
// an XCD executes one after the other are the tiles of one matrix, then of the next: the shared operands are fetched
// once per matrix.  A bijection of the grid whatever the placement really is (jobs beyond the last full group of 8
// keep the plain mapping): placement changes the speed, never the result.
__device__ __forceinline__ void xcd_tile_job(int &tile, int &job)
{
    const unsigned nx = gridDim.x, ny = gridDim.y;
    const unsigned lin = blockIdx.y * nx + blockIdx.x;
    const unsigned nfull = ny & ~7u;
    if (lin < nfull * nx) {
        const unsigned s = lin >> 3;
        job = (int)((lin & 7u) + 8u * (s / nx));
        tile = (int)(s % nx);
    } else {
        job = (int)blockIdx.y;
        tile = (int)blockIdx.x;
    }
}

// lane `src` 's value of a wave-resident double, as a wave-uniform value
__device__ __forceinline__ double lane_value(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}


// ---- 64 x 64 x K tile products on the matrix cores (v_mfma_f64_16x16x4_f64) -------------------
// A block is 4 waves; wave w owns the 32 x 32 quadrant (w >> 1, w & 1) of the 64 x 64 output as
// 2 x 2 accumulators of 16 x 16.  Operands are LDS tiles of pitch TP (66: an [index][k] operand
// read - lane l takes index l & 15, k = k0 + (l >> 4) - touches 32 distinct banks per half wave).
// C/D layout of the instruction: column = lane & 15, row = (lane >> 4) + 4 * register.
// Same peak as the f64 vector pipe (78.6 TFLOP/s) but 4 LDS reads per 4 x 2048 flops instead of 8
// per 16 x 128: the VALU 4 x 4 micro-tile these kernels started with ran at 8 % of peak, LDS-bound.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int TP = 66;
enum { MM_NT = 0, MM_NN = 1, MM_TN = 2 };

// acc += A op B over k in [0, K) (K a multiple of 4, tiles zero-padded):
//   MM_NT: C[i][j] += A[i][k] B[j][k]     MM_NN: C[i][j] += A[i][k] B[k][j]     MM_TN: C[i][j] += A[k][i] B[k][j]
// rmax / cmax: rows / columns of the 64 x 64 output the caller keeps (a short last block of the matrix: 519 =
// 8 x 64 + 7 leaves 7-row and 7-column strips of tiles, 30 % of all tile products of a factorisation).  A 16 x 16
// accumulator wholly outside them is skipped - its values would be dropped anyway - and the others are
// computed exactly as before: the kept results are the same bits.
template <int FORM>
__device__ __forceinline__ void tile_mfma(f64x4 (&acc)[2][2], const double (*A)[TP], const double (*B)[TP], int K,
                                          int rmax = 64, int cmax = 64)
{
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 15, g = lane >> 4;
    const int r0 = 32 * (w >> 1) + li, c0 = 32 * (w & 1) + li;
    const bool ru[2] = {32 * (w >> 1) < rmax, 32 * (w >> 1) + 16 < rmax};      // wave-uniform
    const bool cv[2] = {32 * (w & 1) < cmax, 32 * (w & 1) + 16 < cmax};
    if (!ru[0] || !cv[0]) return;
    if (ru[1] && cv[1]) {
        // all four accumulators kept (every tile but those of a short last block): straight-line code - with the
        // wave-uniform tests below every MFMA sits in a basic block of its own, and no operand read of the next
        // k-step can be hoisted above the products of this one
        // operands of k-step s + 1 are read while the four products of step s are issued (two register sets)
        auto fetch = [&](int k0, double (&a)[2], double (&b)[2]) {
            const int k = k0 + g;
            a[0] = FORM == MM_TN ? A[k][r0] : A[r0][k];
            a[1] = FORM == MM_TN ? A[k][r0 + 16] : A[r0 + 16][k];
            b[0] = FORM == MM_NT ? B[c0][k] : B[k][c0];
            b[1] = FORM == MM_NT ? B[c0 + 16][k] : B[k][c0 + 16];
        };
        auto mult = [&](const double (&a)[2], const double (&b)[2]) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc[1][1], 0, 0, 0);
        };
        if (K == 64) {
            double a0[2], b0[2], a1[2], b1[2];
            fetch(0, a0, b0);
#pragma unroll
            for (int k0 = 0; k0 < 64; k0 += 8) {
                // (the fences keep the order written: left alone, the scheduler moves every read back down to
                // just above its use - one register set, and the LDS latency exposed before each group of four)
                fetch(k0 + 4, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mult(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (k0 + 8 < 64) fetch(k0 + 8, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mult(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int k0 = 0; k0 < K; k0 += 4) {
                double a[2], b[2];
                fetch(k0, a, b);
                mult(a, b);
            }
        }
        return;
    }
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + g;
        double a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            a[u] = ru[u] ? (FORM == MM_TN ? A[k][r0 + 16 * u] : A[r0 + 16 * u][k]) : 0.;
            b[u] = cv[u] ? (FORM == MM_NT ? B[c0 + 16 * u][k] : B[k][c0 + 16 * u]) : 0.;
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int v = 0; v < 2; v++)
                if (ru[u] && cv[v]) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
    }
}

// (row, column) within the 64 x 64 tile of accumulator element acc[u][v][r] of this lane
__device__ __forceinline__ int tile_row(int u, int r) { return 32 * ((int)(threadIdx.x >> 6) >> 1) + 16 * u + (int)((threadIdx.x & 63) >> 4) + 4 * r; }
__device__ __forceinline__ int tile_col(int v) { return 32 * ((int)(threadIdx.x >> 6) & 1) + 16 * v + (int)(threadIdx.x & 15); }

#define TILE_FOR_EACH(u, v, r)                                   \
    _Pragma("unroll") for (int u = 0; u < 2; u++)                \
    _Pragma("unroll") for (int v = 0; v < 2; v++)                \
    _Pragma("unroll") for (int r = 0; r < 4; r++)

// The thread's share of an R x C tile (elements e = tid + 256 q, row-major) of a row-major matrix,
// ALL requested before any is used: rows / columns past row_end / col_end are read at the last
// valid one (the caller masks them).  Written as `T[i][j] = ok ? src[..] : fill` in a loop, every
// element is a branch around its load and a wait before its LDS store - a chain of R * C / 256
// round trips, which was most of the run time of the kernels below at one matrix per launch.
template <int R, int C>
struct TileRegs {
    static constexpr int PER = R * C / 256;
    double v[PER];
    __device__ __forceinline__ void load(const double *src, size_t ld, int row0, int col0, int row_end, int col_end)
    {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int e = (int)threadIdx.x + 256 * q;
            const int i = min(row0 + e / C, row_end - 1), j = min(col0 + e % C, col_end - 1);
            v[q] = gmem(src)[(size_t)i * ld + j];
        }
    }
    static __device__ __forceinline__ int row(int q) { return ((int)threadIdx.x + 256 * q) / C; }
    static __device__ __forceinline__ int col(int q) { return ((int)threadIdx.x + 256 * q) % C; }
};

// 16 x 16 x 16 sub-block products on one wave, operands in LDS tiles of pitch TP, result in the
// instruction's C layout (column = lane & 15, row = (lane >> 4) + 4 * register).
//   NT: C[i][j] += A[ra + i][ka + k] B[rb + j][kb + k]      NN: C[i][j] += A[ra + i][ka + k] B[rb + k][cb + j]
__device__ __forceinline__ f64x4 sub_nt(const double (*A)[TP], int ra, int ka, const double (*B)[TP], int rb, int kb,
                                        f64x4 acc)
{
    const int li = threadIdx.x & 15, g = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ra + li][ka + k0 + g], B[rb + li][kb + k0 + g], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ f64x4 sub_nn(const double (*A)[TP], int ra, int ka, const double (*B)[TP], int rb, int cb,
                                        f64x4 acc)
{
    const int li = threadIdx.x & 15, g = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ra + li][ka + k0 + g], B[rb + k0 + g][cb + li], acc, 0, 0, 0);
    return acc;
}

// Diagonal block k..k+nb: one workgroup of 4 waves, the block in LDS.  Four panels of 16 columns:
// wave 0 holds the panel with one ROW per lane (16 registers) and runs the right-looking
// elimination on it - the pivots and multipliers of other lanes arrive by v_readlane, and the rows
// below the 16 x 16 diagonal sub-block are scaled along, so the panel comes out factored AND solved
// (120 readlane-fma steps instead of the 2016 of a 64-column sweep) - then the four waves update
// the trailing sub-blocks on the matrix cores.  L_kk^-1 for the panel solve of the big matrix: the
// four 16 x 16 diagonal inverses one per wave (column per lane), then wave b builds block column b
// below its diagonal block from products.  13 us per block; the formulations before it: one
// wave with the whole 64 x 64 block in registers 61 us, one wave walking its rows serially
// through LDS 113 us (rocSOLVER's potf2_kernel_small is of this kind, 135 us), four waves sharing
// a 64-column trailing update 87 us (320 block barriers).  Writes L_kk in place and L_kk^-1
// (64 x 64, row-major, zero above the diagonal) to the job's linv slot.  A short last block is
// padded with the identity.  *info = k + j + 1 at the first non-positive pivot.
// the block (identity-padded, in T) -> L in T and L^-1 in X (zeroed by the caller); returns 0 or
// 1 + the index of the first non-positive pivot (block-uniform).  256 threads.
__device__ __forceinline__ int diag_factor_invert(double (*T)[TP], double (*X)[TP], int *s_fail)
{
    __shared__ double dinv[CH_NB];      // 1 / L[j][j]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        constexpr int SB = 16;
        const int c0 = SB * kb;
        if (w == 0) {
            double r[SB];
#pragma unroll
            for (int c = 0; c < SB; c++) r[c] = T[lane][c0 + c];
            int fail = 0;
#pragma unroll
            for (int j = 0; j < SB; j++) {
                const double pj = lane_value(r[j], c0 + j);
                if (!fail && !(pj > 0.)) fail = c0 + j + 1;      // wave-uniform
                // 1 / sqrt(pivot) by v_rsq_f64 + two Newton steps, the pivot's root from it with one
                // Heron correction: the square root -> division chain of the IEEE routines is the serial
                // latency of this kernel (64 pivots one after the other), 250 cycles per pivot against 110
                double y = __builtin_amdgcn_rsq(pj);
                const double hp = .5 * pj;
                y = fma(y, fma(-hp * y, y, .5), y);
                y = fma(y, fma(-hp * y, y, .5), y);
                double rj = pj * y;
                rj = fma(fma(-rj, rj, pj), .5 * y, rj);
                r[j] = lane == c0 + j ? rj : r[j] * y;
                if (lane == 0) dinv[c0 + j] = y;
#pragma unroll
                for (int c = j + 1; c < SB; c++) r[c] -= r[j] * lane_value(r[j], c0 + c);
            }
            if (lane >= c0)
#pragma unroll
                for (int c = 0; c < SB; c++) T[lane][c0 + c] = r[c];
            if (fail && lane == 0) *s_fail = fail;
        }
        __syncthreads();
        if (*s_fail) break;               // block-uniform
        // trailing sub-blocks (i, j), kb < j <= i < 4:  T_ij -= P_i P_j'
        const int nrem = 3 - kb;
        for (int bq = w; bq < nrem * (nrem + 1) / 2; bq += 4) {
            int bi = 0, bj = bq;
            while (bj > bi) { bj -= bi + 1; bi++; }     // bq -> (bi >= bj) within the remaining triangle
            const int i0 = SB * (kb + 1 + bi), j0 = SB * (kb + 1 + bj);
            f64x4 acc = {0., 0., 0., 0.};
            acc = sub_nt(T, i0, c0, T, j0, c0, acc);
#pragma unroll
            for (int q = 0; q < 4; q++) T[i0 + g + 4 * q][j0 + li] -= acc[q];
        }
        __syncthreads();
    }
    if (*s_fail) return *s_fail;
    // X = L^-1.  Diagonal sub-block w: lane c < 16 owns column c,
    // X[i][c] = (delta_ic - sum_{m=c}^{i-1} L[i][m] X[m][c]) / L[i][i]; lane i holds row i of L_ww
    {
        constexpr int SB = 16;
        const int b0 = SB * w;
        double r[SB], x[SB];
#pragma unroll
        for (int c = 0; c < SB; c++) r[c] = T[b0 + li][b0 + c];
#pragma unroll
        for (int i = 0; i < SB; i++) {
            double v = i == lane ? 1. : 0.;
#pragma unroll
            for (int m = 0; m < i; m++) v -= lane_value(r[m], i) * x[m];    // x[m] = 0 for m < c
            x[i] = i < lane ? 0. : v * dinv[b0 + i];
        }
        if (lane < SB)
#pragma unroll
            for (int i = 0; i < SB; i++) X[b0 + i][b0 + lane] = x[i];
    }
    __syncthreads();
    // block column w below its diagonal block: X_iw = -X_ii (sum_{m=w}^{i-1} L_im X_mw), top down
    for (int bi = w + 1; bi < 4; bi++) {
        f64x4 sacc = {0., 0., 0., 0.};
        for (int bm = w; bm < bi; bm++) sacc = sub_nn(T, 16 * bi, 16 * bm, X, 16 * bm, 16 * w, sacc);
#pragma unroll
        for (int q = 0; q < 4; q++) X[16 * bi + g + 4 * q][16 * w + li] = sacc[q];      // S, in place of X_iw
        f64x4 xacc = {0., 0., 0., 0.};
        xacc = sub_nn(X, 16 * bi, 16 * bi, X, 16 * bi, 16 * w, xacc);
#pragma unroll
        for (int q = 0; q < 4; q++) X[16 * bi + g + 4 * q][16 * w + li] = -xacc[q];
    }
    __syncthreads();
    return 0;
}

__global__ __launch_bounds__(256) void k_chol_diag(const ChJobs J, int N, int k)
{
    const ChJob job = ch_job(J, blockIdx.y);
    if (!job.A) return;              // a job that is not to run (ChJob::A): nothing is requested for it
    double *__restrict__ A = job.A;
    int *__restrict__ info = job.info;
    double *__restrict__ linv = ch_linv(job, k);
    __shared__ double T[CH_NB][TP], X[CH_NB][TP];
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    const int nb = min(CH_NB, N - k);
    TileRegs<CH_NB, CH_NB> t;
    const bool first = k == 0 && job.src != nullptr;       // block column 0 is first touched here
    t.load(first ? job.src : A, N, k, k, N, N);
    if (*info != 0) return;          // an earlier panel already failed (block-uniform)
#pragma unroll
    for (int q = 0; q < t.PER; q++) {
        const int i = t.row(q), j = t.col(q);
        const double v = first ? ch_first(job, t.v[q], i == j) : t.v[q];
        T[i][j] = (i < nb && j <= i) ? v : (i == j ? 1. : 0.);
        X[i][j] = 0.;
    }
    if (tid == 0) s_fail = 0;
    __syncthreads();
    const int fail = diag_factor_invert(T, X, &s_fail);
    if (fail) {
        if (tid == 0) *info = k + fail;
        return;
    }
    for (int e = tid; e < CH_NB * CH_NB; e += 256) {
        const int i = e / CH_NB, j = e % CH_NB;
        if (i < nb && j <= i) A[(size_t)(k + i) * N + k + j] = T[i][j];
        linv[e] = X[i][j];
    }
}

// rows r >= k + nb: A[r][k..k+nb) <- A[r][k..k+nb) L_kk^-T = (rows) x (L_kk^-1)': a 64 x 64 x 64 product
// per block of 64 rows on the matrix cores, both operands in LDS
__global__ __launch_bounds__(256) void k_chol_panel(const ChJobs J, int N, int k)
{
    int bx, by;
    xcd_tile_job(bx, by);
    const ChJob job = ch_job(J, by);
    if (!job.A) return;              // a job that is not to run (ChJob::A): nothing is requested for it
    double *__restrict__ A = job.A;
    const double *__restrict__ linv = ch_linv(job, k);
    __shared__ double P[CH_NB][TP], X[CH_NB][TP];
    const int nb = min(CH_NB, N - k);
    const int info0 = *gmem(job.info);   // requested first, tested after the tiles are requested: one round trip, not two
    const int r0 = k + nb + bx * 64;
    const int rows = min(64, N - r0);
    {
        TileRegs<CH_NB, CH_NB> x, t;
        x.load(linv, CH_NB, 0, 0, CH_NB, CH_NB);
        t.load((k == 0 && job.src != nullptr) ? job.src : A, N, r0, k, N, N);     // (off-diagonal tiles: no shift)
        if (info0 != 0) return;
#pragma unroll
        for (int q = 0; q < t.PER; q++) {
            const int i = t.row(q), j = t.col(q);
            X[i][j] = x.v[q];
            P[i][j] = (i < rows && j < nb) ? t.v[q] : 0.;
        }
    }
    __syncthreads();
    f64x4 acc[2][2] = {};
    tile_mfma<MM_NT>(acc, P, X, CH_NB, rows, nb);
    TILE_FOR_EACH(u, v, r) {
        const int i = tile_row(u, r), j = tile_col(v);
        if (i < rows && j < nb) gmem(A)[(size_t)(r0 + i) * N + k + j] = acc[u][v][r];
    }
}

// acc += Pi Pj' over one staged half-panel of KS columns (the wave's 2 x 2 accumulators; see tile_mfma for the
// skipping of accumulators outside a short last block and for why the all-kept case is straight-line code)
template <int KS, int KP>
__device__ __forceinline__ void half_panel_mfma(f64x4 (&acc)[2][2], const double (*Pi)[KP], const double (*Pj)[KP],
                                                int r0, int c0, int g, const bool (&ru)[2], const bool (&cv)[2])
{
    if (!ru[0] || !cv[0]) return;
    if (ru[1] && cv[1]) {
        // two operand register sets, the reads of k-step s + 1 fenced above the products of step s (see tile_mfma)
        static_assert(KS % 8 == 0, "two k-steps per round");
        auto fetch = [&](int k0, double (&a)[2], double (&b)[2]) {
            const int kk = k0 + g;
            a[0] = Pi[r0][kk]; a[1] = Pi[r0 + 16][kk]; b[0] = Pj[c0][kk]; b[1] = Pj[c0 + 16][kk];
        };
        auto mult = [&](const double (&a)[2], const double (&b)[2]) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc[1][1], 0, 0, 0);
        };
        double a0[2], b0[2], a1[2], b1[2];
        fetch(0, a0, b0);
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += 8) {
            fetch(k0 + 4, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mult(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 + 8 < KS) fetch(k0 + 8, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mult(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += 4) {
        const int kk = k0 + g;
        double a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; u++) { a[u] = ru[u] ? Pi[r0 + 16 * u][kk] : 0.; b[u] = cv[u] ? Pj[c0 + 16 * u][kk] : 0.; }
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int v = 0; v < 2; v++)
                if (ru[u] && cv[v]) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
    }
}

// triangular tile index -> (ti >= tj)
__device__ __forceinline__ void tri_tile(int b, int &ti, int &tj)
{
    ti = (int)((sqrt(8. * b + 1.) - 1.) / 2.);
    while ((ti + 1) * (ti + 2) / 2 <= b) ti++;
    while (ti * (ti + 1) / 2 > b) ti--;
    tj = b - ti * (ti + 1) / 2;
}

// trailing update: for i >= j >= k + nb:  A[i][j] -= sum_t P[i][t] P[j][t],  P = A[:, k..k+nb)
// one 64 x 64 tile per block (tiles on or below the diagonal), on the matrix cores.
// NEXT (few matrices per launch, where the chain of launches is what costs): the block of tile (0, 0)
// - the next diagonal block, which nobody else touches - goes straight on to factor and invert it, so
// the next panel solve can follow without a launch of k_chol_diag in between.
template <bool NEXT>
__global__ __launch_bounds__(256) void k_chol_update(const ChJobs J, int N, int k)
{
    int bx, by;
    xcd_tile_job(bx, by);
    const ChJob job = ch_job(J, by);
    if (!job.A) return;              // a job that is not to run (ChJob::A): nothing is requested for it
    double *__restrict__ A = job.A;
    // the panel is staged 32 columns at a time: 35 KB of LDS per block -> 4 blocks (16 waves) per CU
    // instead of 2; this kernel is latency-bound (PMC: ~1.7 TB/s of HBM traffic), occupancy is what helps
    constexpr int KS = 32, KP = KS + 2;
    extern __shared__ double dyn_lds[];          // NEXT: T and X of the diagonal block (the staging tiles alias them)
    __shared__ double stat_lds[NEXT ? 1 : 2 * 64 * KP];
    double (*Pi)[KP] = (double (*)[KP])(NEXT ? dyn_lds : stat_lds);
    double (*Pj)[KP] = Pi + 64;
    __shared__ int s_fail;
    const int nb = min(CH_NB, N - k);
    const int base = k + nb;
    int ti, tj;
    tri_tile(bx, ti, tj);
    const int i0 = base + ti * 64, j0 = base + tj * 64;
    // everything the block reads is requested up front: the tile being updated and the first
    // half-panels; the second half-panels while the first are multiplied
    const int info0 = *gmem(job.info);   // (first: loads return in order, its wait leaves the tiles in flight)
    f64x4 old[2][2];
    TILE_FOR_EACH(u, v, r) {
        const int i = min(i0 + tile_row(u, r), N - 1), j = min(j0 + tile_col(v), N - 1);
        old[u][v][r] = gmem(A)[(size_t)i * N + j];
    }
    TileRegs<64, KS> pi, pj;
    pi.load(A, N, i0, k, N, N);
    pj.load(A, N, j0, k, N, N);
    if (info0 != 0) return;
    f64x4 acc[2][2] = {};
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 15, g = lane >> 4;
    const int r0 = 32 * (w >> 1) + li, c0 = 32 * (w & 1) + li;
    // 16 x 16 accumulators wholly outside the matrix (a short last block) are skipped: see tile_mfma
    const int rmax = N - i0, cmax = N - j0;
    const bool ru[2] = {32 * (w >> 1) < rmax, 32 * (w >> 1) + 16 < rmax};
    const bool cv[2] = {32 * (w & 1) < cmax, 32 * (w & 1) + 16 < cmax};
    for (int ks = 0; ks < nb; ks += KS) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < pi.PER; q++) {
            const int r = pi.row(q), c = pi.col(q);
            Pi[r][c] = (i0 + r < N && ks + c < nb) ? pi.v[q] : 0.;
            Pj[r][c] = (j0 + r < N && ks + c < nb) ? pj.v[q] : 0.;
        }
        if (ks + KS < nb) {
            pi.load(A, N, i0, k + ks + KS, N, N);
            pj.load(A, N, j0, k + ks + KS, N, N);
        }
        __syncthreads();
        half_panel_mfma<KS, KP>(acc, Pi, Pj, r0, c0, g, ru, cv);
    }
    if (!NEXT || bx != 0) {
        TILE_FOR_EACH(u, v, r) {
            const int i = i0 + tile_row(u, r), j = j0 + tile_col(v);
            if (i < N && j <= i) gmem(A)[(size_t)i * N + j] = old[u][v][r] - acc[u][v][r];
        }
        return;
    }
    // tile (0, 0): the next diagonal block, factored and inverted here
    double (*T)[TP] = (double (*)[TP])dyn_lds;
    double (*X)[TP] = T + CH_NB;
    const int nbn = min(CH_NB, N - base);
    __syncthreads();                 // the staging tiles (aliases of T / X) have been read
    for (int e = threadIdx.x; e < CH_NB * CH_NB; e += 256) X[e / CH_NB][e % CH_NB] = 0.;
    TILE_FOR_EACH(u, v, r) {
        const int i = tile_row(u, r), j = tile_col(v);
        T[i][j] = (i < nbn && j <= i) ? old[u][v][r] - acc[u][v][r] : (i == j ? 1. : 0.);
    }
    if (threadIdx.x == 0) s_fail = 0;
    __syncthreads();
    const int fail = diag_factor_invert(T, X, &s_fail);
    if (fail) {
        if (threadIdx.x == 0) *job.info = base + fail;
        return;
    }
    double *__restrict__ linv = ch_linv(job, base);
    for (int e = threadIdx.x; e < CH_NB * CH_NB; e += 256) {
        const int i = e / CH_NB, j = e % CH_NB;
        if (i < nbn && j <= i) A[(size_t)(base + i) * N + base + j] = T[i][j];
        linv[e] = X[i][j];
    }
}

// The same updates, one block COLUMN at a time (many matrices per launch, where memory traffic is what costs):
// tile (i, j) of block column j takes the updates of ALL earlier panels k < j in one go - old -= P_ik P_jk',
// panel after panel in the order and with the roundings of k_chol_update, so the factor is the same bits -
// but stays in registers in between instead of going through memory once per panel: half the traffic.
__global__ __launch_bounds__(256) void k_chol_col(const ChJobs J, int N, int jb)
{
    int bx, by;
    xcd_tile_job(bx, by);
    const ChJob job = ch_job(J, by);
    if (!job.A) return;              // a job that is not to run (ChJob::A): nothing is requested for it
    double *__restrict__ A = job.A;
    constexpr int KS = 32, KP = KS + 2;
    __shared__ double stat_lds[2 * 64 * KP];
    double (*Pi)[KP] = (double (*)[KP])stat_lds;
    double (*Pj)[KP] = Pi + 64;
    const int j0 = jb * CH_NB, i0 = j0 + bx * 64;
    const int info0 = *gmem(job.info);   // (first: loads return in order, its wait leaves the tiles in flight)
    f64x4 old[2][2];
    {
        // the tiles of block column jb >= 1 are touched here for the first time: from src when the job has one
        const double *__restrict__ A0 = job.src ? job.src : A;
        TILE_FOR_EACH(u, v, r) {
            const int i = min(i0 + tile_row(u, r), N - 1), j = min(j0 + tile_col(v), N - 1);
            old[u][v][r] = gmem(A0)[(size_t)i * N + j];
        }
        if (job.src && job.shifted && bx == 0)
            TILE_FOR_EACH(u, v, r)
                if (tile_row(u, r) == tile_col(v)) old[u][v][r] = ch_first(job, old[u][v][r], true);
    }
    // Two half-panels of operands are in flight: the loads of half-panel h + 2 are issued while h is multiplied, so a
    // load has two multiply phases to arrive in (one was not enough: a block spent 2.8 us per half-panel, 0.85 of them
    // in its 32 MFMAs, and with two blocks per CU the matrix pipe sat idle 59 % of the time)
    TileRegs<64, KS> pi0, pj0, pi1, pj1;
    pi0.load(A, N, i0, 0, N, N);
    pj0.load(A, N, j0, 0, N, N);
    if (j0 > KS) {
        pi1.load(A, N, i0, KS, N, N);
        pj1.load(A, N, j0, KS, N, N);
    }
    if (info0 != 0) return;
    f64x4 acc[2][2] = {};
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 15, g = lane >> 4;
    const int r0 = 32 * (w >> 1) + li, c0 = 32 * (w & 1) + li;
    const int rmax = N - i0, cmax = N - j0;            // accumulators wholly outside the matrix are skipped
    const bool ru[2] = {32 * (w >> 1) < rmax, 32 * (w >> 1) + 16 < rmax};
    const bool cv[2] = {32 * (w & 1) < cmax, 32 * (w & 1) + 16 < cmax};
    auto half = [&](TileRegs<64, KS> &pi, TileRegs<64, KS> &pj, int next_kc) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < pi.PER; q++) {
            const int r = pi.row(q), c = pi.col(q);
            Pi[r][c] = i0 + r < N ? pi.v[q] : 0.;
            Pj[r][c] = j0 + r < N ? pj.v[q] : 0.;
        }
        if (next_kc < j0) {
            pi.load(A, N, i0, next_kc, N, N);
            pj.load(A, N, j0, next_kc, N, N);
        }
        __syncthreads();
        half_panel_mfma<KS, KP>(acc, Pi, Pj, r0, c0, g, ru, cv);
    };
    for (int kp = 0; kp < j0; kp += CH_NB) {           // one panel = two half-panels (j0 is a multiple of 64)
        half(pi0, pj0, kp + CH_NB);
        half(pi1, pj1, kp + CH_NB + KS);
        TILE_FOR_EACH(u, v, r) {                       // the panel's update, rounded like k_chol_update's
            old[u][v][r] = old[u][v][r] - acc[u][v][r];
            acc[u][v][r] = 0.;
        }
    }
    TILE_FOR_EACH(u, v, r) {
        const int i = i0 + tile_row(u, r), j = j0 + tile_col(v);
        if (i < N && j <= i) gmem(A)[(size_t)i * N + j] = old[u][v][r];
    }
}

} // namespace

// Enqueue the factorisation of `njobs` matrices (all N x N).  The jobs' info words must have
// been zeroed (or preset non-zero to skip a job) by the caller, on the same stream.
// whether chol_lower_jobs[_on] takes the block-column route for this many jobs (the only route that honours ChJob::src)
static inline bool chol_route_by_column(int njobs)
{
    static const int fuse_jobs = getenv("APMF_CHOL_FUSE_JOBS") ? atoi(getenv("APMF_CHOL_FUSE_JOBS")) : CH_FUSE_JOBS;
    static const bool by_column = !(getenv("APMF_CHOL_COL") && getenv("APMF_CHOL_COL")[0] == '0');
    return njobs > fuse_jobs && by_column;
}

static inline int chol_lower_jobs_on(hipStream_t stream, const ChJobs &J, int njobs, int N)
{
    if (njobs <= 0) return APMF_OK;
    // few matrices: the chain of launches is what costs, so each update launch also factors the next
    // diagonal block; many: the block in a launch of its own (and 4 update blocks per CU)
    // (APMF_CHOL_FUSE_JOBS overrides the threshold: tuning runs; the two routes give the same bits)
    static const int fuse_jobs = getenv("APMF_CHOL_FUSE_JOBS") ? atoi(getenv("APMF_CHOL_FUSE_JOBS")) : CH_FUSE_JOBS;
    const bool next = njobs <= fuse_jobs;
    constexpr size_t next_lds = 2 * (size_t)CH_NB * TP * sizeof(double);
    if (next && next_lds > 64 * 1024) {
        // 67.6 KB of dynamic LDS: the attribute is per device and stays set - once per device and process, not per
        // call (several host threads drive batches of their own through here at the same time)
        static std::atomic<unsigned long long> attr_set{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(attr_set.load(std::memory_order_acquire) & bit)) {
            (void)hipFuncSetAttribute((const void *)k_chol_update<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)next_lds);
            attr_set.fetch_or(bit, std::memory_order_release);
        }
    }
    static const bool by_column = !(getenv("APMF_CHOL_COL") && getenv("APMF_CHOL_COL")[0] == '0');   // 0: comparison runs
    if (!next && by_column) {
        // many matrices: block column by block column, each tile updated once with all earlier panels
        const int nbk = (N + CH_NB - 1) / CH_NB;
        for (int jb = 0; jb < nbk; jb++) {
            const int k = jb * CH_NB;
            if (jb > 0) APMF_LAUNCH(k_chol_col, dim3(nbk - jb, njobs), dim3(256), 0, stream, J, N, jb);
            APMF_LAUNCH(k_chol_diag, dim3(1, njobs), dim3(256), 0, stream, J, N, k);
            const int rest = N - k - CH_NB;
            if (rest > 0) APMF_LAUNCH(k_chol_panel, dim3((rest + 63) / 64, njobs), dim3(256), 0, stream, J, N, k);
        }
        APMF_CHECK_LAUNCH();
        return APMF_OK;
    }
    for (int k = 0; k < N; k += CH_NB) {
        const int nb = N - k < CH_NB ? N - k : CH_NB;
        if (k == 0 || !next) APMF_LAUNCH(k_chol_diag, dim3(1, njobs), dim3(256), 0, stream, J, N, k);
        const int rest = N - k - nb;
        if (rest > 0) {
            APMF_LAUNCH(k_chol_panel, dim3((rest + 63) / 64, njobs), dim3(256), 0, stream, J, N, k);
            const int tiles = (rest + 63) / 64;
            if (next)
                APMF_LAUNCH(k_chol_update<true>, dim3(tiles * (tiles + 1) / 2, njobs), dim3(256), next_lds,
                                   stream, J, N, k);
            else
                APMF_LAUNCH(k_chol_update<false>, dim3(tiles * (tiles + 1) / 2, njobs), dim3(256), 0,
                                   stream, J, N, k);
        }
    }
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

static inline int chol_lower_jobs(apmf_ctx *ctx, const ChJobs &J, int njobs, int N)
{
    return chol_lower_jobs_on(ctx->stream, J, njobs, N);
}
