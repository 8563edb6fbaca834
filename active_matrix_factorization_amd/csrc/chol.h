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
#include "common.h"

constexpr int CH_NB = 64;

namespace {   // one private copy of the kernels per translation unit that includes this header

struct ChJob {
    double *A;      // N x N row-major, factored in place (lower triangle)
    int *info;      // 0, or 1-based index of the first non-positive pivot (must be zeroed first;
                    // a non-zero value on entry skips the job)
    double *linv;   // inverse of the diagonal block of the current panel: CH_NB^2 doubles ...
    int linv_keep;  // ... 0: one slot reused by every panel; 1: slot p for panel p (all kept)
};

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
template <int FORM>
__device__ __forceinline__ void tile_mfma(f64x4 (&acc)[2][2], const double (*A)[TP], const double (*B)[TP], int K)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int r0 = 32 * (w >> 1) + li, c0 = 32 * (w & 1) + li;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + g;
        double a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            a[u] = FORM == MM_TN ? A[k][r0 + 16 * u] : A[r0 + 16 * u][k];
            b[u] = FORM == MM_NT ? B[c0 + 16 * u][k] : B[k][c0 + 16 * u];
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int v = 0; v < 2; v++) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
    }
}

// (row, column) within the 64 x 64 tile of accumulator element acc[u][v][r] of this lane
__device__ __forceinline__ int tile_row(int u, int r) { return 32 * ((int)(threadIdx.x >> 6) >> 1) + 16 * u + (int)((threadIdx.x & 63) >> 4) + 4 * r; }
__device__ __forceinline__ int tile_col(int v) { return 32 * ((int)(threadIdx.x >> 6) & 1) + 16 * v + (int)(threadIdx.x & 15); }

#define TILE_FOR_EACH(u, v, r)                                   \
    _Pragma("unroll") for (int u = 0; u < 2; u++)                \
    _Pragma("unroll") for (int v = 0; v < 2; v++)                \
    _Pragma("unroll") for (int r = 0; r < 4; r++)

// Diagonal block k..k+nb, one wave, everything in registers: lane t holds row t of the block
// (64 doubles) and other lanes' entries arrive by v_readlane: 55 us per block.  The LDS
// formulations were slower: one wave walking its rows serially 113 us (dependent LDS round
// trips; rocSOLVER's potf2_kernel_small is of this kind, 135 us), four waves sharing the
// trailing update 87 us (320 block barriers).  Writes L_kk in place and L_kk^-1 (64 x 64,
// row-major, zero above the diagonal) to the job's linv slot for the panel solve.  A short
// last block is padded with the identity.  *info = k + j + 1 at the first non-positive pivot.
__global__ __launch_bounds__(64) void k_chol_diag(const ChJobs J, int N, int k)
{
    const ChJob job = ch_job(J, blockIdx.y);
    double *__restrict__ A = job.A;
    int *__restrict__ info = job.info;
    double *__restrict__ linv = ch_linv(job, k);
    const int t = threadIdx.x;
    const int nb = min(CH_NB, N - k);
    if (*info != 0) return;          // an earlier panel already failed
    double r[CH_NB];
#pragma unroll
    for (int c = 0; c < CH_NB; c++)
        r[c] = (t < nb && c <= t) ? A[(size_t)(k + t) * N + k + c] : (c == t ? 1. : 0.);
    // right-looking factorisation; entries above the diagonal hold don't-care values
#pragma unroll
    for (int j = 0; j < CH_NB; j++) {
        const double pj = lane_value(r[j], j);
        if (!(pj > 0.)) {            // wave-uniform
            if (t == 0) *info = k + j + 1;
            return;
        }
        const double rj = sqrt(pj);
        r[j] = t == j ? rj : r[j] / rj;
#pragma unroll
        for (int c = j + 1; c < CH_NB; c++) r[c] -= r[j] * lane_value(r[j], c);
    }
#pragma unroll
    for (int c = 0; c < CH_NB; c++)
        if (t < nb && c <= t) A[(size_t)(k + t) * N + k + c] = r[c];
    // X = L^-1: lane c owns column c, X[i][c] = (delta_ic - sum_{m=c}^{i-1} L[i][m] X[m][c]) / L[i][i]
    double x[CH_NB];
#pragma unroll
    for (int i = 0; i < CH_NB; i++) {
        double v = i == t ? 1. : 0.;
#pragma unroll
        for (int m = 0; m < i; m++) v -= lane_value(r[m], i) * x[m];   // x[m] = 0 for m < c
        x[i] = i < t ? 0. : v / lane_value(r[i], i);
    }
#pragma unroll
    for (int i = 0; i < CH_NB; i++) linv[i * CH_NB + t] = x[i];
}

// rows r >= k + nb: A[r][k..k+nb) <- A[r][k..k+nb) L_kk^-T = (rows) x (L_kk^-1)': a 64 x 64 x 64 product
// per block of 64 rows on the matrix cores, both operands in LDS
__global__ __launch_bounds__(256) void k_chol_panel(const ChJobs J, int N, int k)
{
    const ChJob job = ch_job(J, blockIdx.y);
    double *__restrict__ A = job.A;
    const double *__restrict__ linv = ch_linv(job, k);
    __shared__ double P[CH_NB][TP], X[CH_NB][TP];
    const int tid = threadIdx.x;
    const int nb = min(CH_NB, N - k);
    if (*job.info != 0) return;
    const int r0 = k + nb + blockIdx.x * 64;
    const int rows = min(64, N - r0);
    for (int e = tid; e < CH_NB * CH_NB; e += 256) {
        const int i = e / CH_NB, j = e % CH_NB;
        X[i][j] = linv[e];
        P[i][j] = (i < rows && j < nb) ? A[(size_t)(r0 + i) * N + k + j] : 0.;
    }
    __syncthreads();
    f64x4 acc[2][2] = {};
    tile_mfma<MM_NT>(acc, P, X, CH_NB);
    TILE_FOR_EACH(u, v, r) {
        const int i = tile_row(u, r), j = tile_col(v);
        if (i < rows && j < nb) A[(size_t)(r0 + i) * N + k + j] = acc[u][v][r];
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
// one 64 x 64 tile per block (tiles on or below the diagonal), on the matrix cores
__global__ __launch_bounds__(256) void k_chol_update(const ChJobs J, int N, int k)
{
    const ChJob job = ch_job(J, blockIdx.y);
    double *__restrict__ A = job.A;
    // the panel is staged 32 columns at a time: 35 KB of LDS per block -> 4 blocks (16 waves) per CU
    // instead of 2; this kernel is latency-bound (PMC: ~1.7 TB/s of HBM traffic), occupancy is what helps
    constexpr int KS = 32, KP = KS + 2;
    __shared__ double Pi[64][KP], Pj[64][KP];
    if (*job.info != 0) return;
    const int nb = min(CH_NB, N - k);
    const int base = k + nb;
    int ti, tj;
    tri_tile((int)blockIdx.x, ti, tj);
    const int i0 = base + ti * 64, j0 = base + tj * 64;
    f64x4 old[2][2];
    TILE_FOR_EACH(u, v, r) {
        const int i = i0 + tile_row(u, r), j = j0 + tile_col(v);
        old[u][v][r] = (i < N && j <= i) ? A[(size_t)i * N + j] : 0.;
    }
    f64x4 acc[2][2] = {};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int r0 = 32 * (w >> 1) + li, c0 = 32 * (w & 1) + li;
    for (int ks = 0; ks < nb; ks += KS) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * KS; e += 256) {
            const int r = e / KS, c = e % KS;
            Pi[r][c] = (i0 + r < N && ks + c < nb) ? A[(size_t)(i0 + r) * N + k + ks + c] : 0.;
            Pj[r][c] = (j0 + r < N && ks + c < nb) ? A[(size_t)(j0 + r) * N + k + ks + c] : 0.;
        }
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += 4) {
            const int kk = k0 + g;
            double a[2], b[2];
#pragma unroll
            for (int u = 0; u < 2; u++) { a[u] = Pi[r0 + 16 * u][kk]; b[u] = Pj[c0 + 16 * u][kk]; }
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int v = 0; v < 2; v++)
                    acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
        }
    }
    TILE_FOR_EACH(u, v, r) {
        const int i = i0 + tile_row(u, r), j = j0 + tile_col(v);
        if (i < N && j <= i) A[(size_t)i * N + j] = old[u][v][r] - acc[u][v][r];
    }
}

} // namespace

// Enqueue the factorisation of `njobs` matrices (all N x N).  The jobs' info words must have
// been zeroed (or preset non-zero to skip a job) by the caller, on the same stream.
static inline int chol_lower_jobs(apmf_ctx *ctx, const ChJobs &J, int njobs, int N)
{
    if (njobs <= 0) return APMF_OK;
    for (int k = 0; k < N; k += CH_NB) {
        const int nb = N - k < CH_NB ? N - k : CH_NB;
        hipLaunchKernelGGL(k_chol_diag, dim3(1, njobs), dim3(64), 0, ctx->stream, J, N, k);
        const int rest = N - k - nb;
        if (rest > 0) {
            hipLaunchKernelGGL(k_chol_panel, dim3((rest + 63) / 64, njobs), dim3(256), 0, ctx->stream, J, N, k);
            const int tiles = (rest + 63) / 64;
            hipLaunchKernelGGL(k_chol_update, dim3(tiles * (tiles + 1) / 2, njobs), dim3(256), 0, ctx->stream, J,
                               N, k);
        }
    }
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}
