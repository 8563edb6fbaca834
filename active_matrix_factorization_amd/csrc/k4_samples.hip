// k4_samples.hip - statistics over S posterior samples of (U, V): BayesianPMF.predict /
// pred_variance / prob_ge_cutoff (bayes_pmf.py:433-448, 528-538), float64.
//
// The reference materialises S dense (n, m) products U_s V_s' and reduces over s with
// np.mean / np.var / a >= count.  Here every candidate (i, j) keeps running statistics of
//   x_s = U_s[i] . V_s[j] + add
// in registers; nothing of size S x n x m ever exists.  The contraction (2 d S flops per
// candidate, S ~ 128) makes this the one compute-bound kernel of the path, and it is a batch
// of small GEMMs on the matrix cores (v_mfma_f64_16x16x4_f64, the latent dimension
// zero-padded to a multiple of 4).
// A block is 2 x 2 waves = a 32-row x 64-column tile of the grid; a wave owns 16 x 32 of it
// (two 16 x 16 accumulators, 8 candidates per lane: the 7 registers of running statistics per
// candidate are what bounds the tile).  Per sample the block's operands are two CONTIGUOUS
// chunks of the row-major sample arrays (32 rows of U_s, 64 rows of V_s): they are staged
// into LDS (double-buffered, the next sample's chunk in flight during this sample's
// MFMAs) and the 4 waves read their MFMA operands from there (lane l: row / column l & 15,
// k = 4 q + (l >> 4)).  Measured 943 x 1682, d = 15, S = 128: 0.22-0.23 ms = 28-29 TFLOP/s (round 2: 0.25-0.27 ms;
// with parts of the sample loop compiled out: no staging 0.17 ms, no statistics 0.21, no barrier 0.215, LDS reads +
// MFMAs alone 0.15 - against 0.083 for the MFMAs at their peak rate and 0.11 for the 4-blocks-on-some-CUs
// placement of 810 blocks; the 8 x 2 VALU
// micro-tile kernel: 0.80 ms; per-wave operand gathers from global memory: 0.49 ms; a flat LDS copy
// with run-time row length: 0.40 ms, VALU-bound on address arithmetic).
// Variance is accumulated around the first sample (shifted sums), which matches numpy's
// two-pass np.var to rounding.
#include "common.h"

namespace {

using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int CT = 2;          // 16 x 16 tiles per wave, side by side
constexpr int BR = 32;         // block tile rows  (2 waves)
constexpr int BC = 32 * CT;    // block tile columns (2 waves x CT tiles)

// workgroup barrier ordering LDS traffic only (see k2_score.hip: __syncthreads() would also
// drain the global loads that are deliberately left in flight)
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// NQ = ceil(d / 4) k-steps of the 16 x 16 x 4 instruction.  LDS image of one sample: BR rows of U_s
// then BC rows of V_s, every row zero-padded to 4 NQ doubles at a pitch of 4 NQ + 2 (16 rows of an
// operand read then fall on distinct banks).  With the row length a compile-time constant the
// MFMA operands are read at immediate offsets from one per-lane base and the padding needs no masks:
// the contraction loop has no VALU instruction left.  (The flat copy with run-time d it replaces
// spent 242 VALU instructions per wave and sample - address arithmetic - against 8 MFMAs: the
// kernel was VALU-bound at 16 TFLOP/s.)
//
// Staging (round 3).  Thread (rp, sk) copies element sk of rows rp, rp + RPP, ... of the U part and of the V part:
// per pass ONE global load at (scalar base of the sample) + (a 32-bit offset fixed for the whole launch) and ONE
// LDS store at a fixed offset.  There are no masks and no branches: a thread without a slot of its own - beyond the
// RPP x DP threads that tile a pass, a padding column k >= d, a row past the part or past the matrix edge - repeats
// the copy of a thread that has one (same source, same destination, same value), and the padding columns are zeroed
// once before the loop.  Rows past the matrix edge so hold a copy of its last row; their products are never written.
// The sample loop is one basic block with the loads fenced above the MFMAs: in the round-2 form the compiler sank
// them into the conditional store block BELOW the MFMAs (next to their use), so that every sample waited out the
// whole L2 latency, and spent ~50 VALU instructions per sample on addresses and selects beside 8 MFMAs.
// PROB: the exceedance counts (prob_ge_cutoff) are compiled in only where asked for (two VALU instructions and a
// register per candidate and sample).
// (NQ <= 5 is held to 128 VGPRs = 4 blocks per CU: at 943 x 1682 the 810 blocks then run as one wave
// of blocks instead of two)
template <int NQ, bool PROB>
__global__ __launch_bounds__(256, NQ <= 5 ? 4 : 2) void k_sample_stats(int S, int n, int m, int d, int ld,
                                                      const double *__restrict__ Us,
                                                      const double *__restrict__ Vs, double add,
                                                      double cutoff, double *__restrict__ o_mean,
                                                      double *__restrict__ o_var,
                                                      double *__restrict__ o_prob)
{
    constexpr int DP = 4 * NQ;             // padded row length
    constexpr int LS = DP + 2;             // LDS row pitch (doubles)
    constexpr int RPP = 256 / DP;          // rows staged per pass of the block
    constexpr int NITU = (BR + RPP - 1) / RPP, NITV = (BC + RPP - 1) / RPP;
    constexpr int TILE = (BR + BC) * LS;
    extern __shared__ __attribute__((aligned(16))) double sm[];   // 2 x (BR + BC) x LS
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int wr = wave >> 1, wc = wave & 1;
    const int ri = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.y * BR, col0 = blockIdx.x * BC;
    const int u_rows = min(BR, n - row0), v_rows = min(BC, m - col0);   // >= 1: the grid covers ld < m + BC columns

    const int tq = (int)threadIdx.x % (RPP * DP);
    const int sk = min(tq % DP, d - 1), rp = tq / DP;
    unsigned gu[NITU], gv[NITV];           // byte offsets from the sample's first row of the block
    unsigned lu[NITU], lv[NITV];           // byte offsets into an LDS image
#pragma unroll
    for (int it = 0; it < NITU; it++) {
        const int row = min(it * RPP + rp, BR - 1);
        gu[it] = (unsigned)((min(row, u_rows - 1) * d + sk) * 8);
        lu[it] = (unsigned)((row * LS + sk) * 8);
    }
#pragma unroll
    for (int it = 0; it < NITV; it++) {
        const int row = min(it * RPP + rp, BC - 1);
        gv[it] = (unsigned)((min(row, v_rows - 1) * d + sk) * 8);
        lv[it] = (unsigned)(((BR + row) * LS + sk) * 8);
    }
    double su[NITU], sv[NITV];
    auto fetch = [&](int s) {
        const char *ub = (const char *)(Us + ((size_t)s * n + row0) * d);
        const char *vb = (const char *)(Vs + ((size_t)s * m + col0) * d);
#pragma unroll
        for (int it = 0; it < NITU; it++) su[it] = *(const double *)(ub + gu[it]);
#pragma unroll
        for (int it = 0; it < NITV; it++) sv[it] = *(const double *)(vb + gv[it]);
    };
    auto commit = [&](int buf) {
        char *dst = (char *)(sm + (size_t)buf * TILE);
#pragma unroll
        for (int it = 0; it < NITU; it++) *(double *)(dst + lu[it]) = su[it];
#pragma unroll
        for (int it = 0; it < NITV; it++) *(double *)(dst + lv[it]) = sv[it];
    };

    // operand slots of this lane within an LDS image: U row (16 wr + ri), V rows (32 wc + 16 t + ri),
    // element kq of every group of four
    const int aoff = (16 * wr + ri) * LS + kq;
    int boff[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) boff[t] = (BR + 32 * wc + 16 * t + ri) * LS + kq;

    double c[CT][4], s1[CT][4], s2[CT][4];
    int cnt[CT][4];
#pragma unroll
    for (int t = 0; t < CT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) { c[t][r] = 0.; s1[t][r] = 0.; s2[t][r] = 0.; cnt[t][r] = 0; }

    fetch(0);
    for (int k = threadIdx.x; k < 2 * TILE; k += 256) sm[k] = 0.;      // the padding columns stay zero from here on
    __syncthreads();
    commit(0);
    __syncthreads();
    for (int s = 0; s < S; s++) {
        const int buf = s & 1;
        fetch(min(s + 1, S - 1));            // the last sample re-reads itself: no branch
        __builtin_amdgcn_sched_barrier(0);   // the loads are in flight during this sample's MFMAs
        const double *img = sm + (size_t)buf * TILE;
        f64x4 acc[CT];
#pragma unroll
        for (int t = 0; t < CT; t++) acc[t] = (f64x4){add, add, add, add};   // x_s = u.v + add
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const double av = img[aoff + 4 * q];
#pragma unroll
            for (int t = 0; t < CT; t++)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, img[boff[t] + 4 * q], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        commit(buf ^ 1);
        if (s == 0) {            // the shift of the running sums is the first sample
#pragma unroll
            for (int t = 0; t < CT; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) c[t][r] = acc[t][r];
        }
#pragma unroll
        for (int t = 0; t < CT; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double xv = acc[t][r];
                const double dlt = xv - c[t][r];
                s1[t][r] += dlt;
                s2[t][r] = fma(dlt, dlt, s2[t][r]);
                if (PROB) cnt[t][r] += (xv >= cutoff) ? 1 : 0;
            }
        lds_barrier();
    }
    const double invS = 1. / (double)S;
#pragma unroll
    for (int t = 0; t < CT; t++) {
        const int col = col0 + 32 * wc + 16 * t + ri;
        if (col >= ld) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = row0 + 16 * wr + kq + 4 * r;
            if (row >= n) continue;
            const size_t o = (size_t)row * ld + col;
            const double md = s1[t][r] * invS;
            double var = s2[t][r] * invS - md * md;
            if (var < 0.) var = 0.;
            const bool pad = col >= m;          // padding columns stay zero
            if (o_mean) o_mean[o] = pad ? 0. : c[t][r] + md;
            if (o_var) o_var[o] = pad ? 0. : var;
            if (PROB) o_prob[o] = pad ? 0. : (double)cnt[t][r] / (double)S;
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
    APMF_REQUIRE(d <= APMF_MAX_D, "latent_d=%d unsupported (1..%d)", d, APMF_MAX_D);
    dim3 grid((ld + BC - 1) / BC, (n + BR - 1) / BR);
    const int nq = (d + 3) / 4;
    const size_t smem = 2 * (size_t)(BR + BC) * (4 * nq + 2) * sizeof(double);
#define APMF_K4(NQ)                                                                                \
    case NQ: {                                                                                     \
        if (smem > 64 * 1024)                                                                      \
        {                                                                                          \
            APMF_CHECK_HIP(hipFuncSetAttribute((const void *)k_sample_stats<NQ, false>,            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
            APMF_CHECK_HIP(hipFuncSetAttribute((const void *)k_sample_stats<NQ, true>,             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        }                                                                                          \
        ProfScope prof(ctx);                                                                       \
        if (d_prob)                                                                                \
            hipLaunchKernelGGL((k_sample_stats<NQ, true>), grid, dim3(256), smem, ctx->stream, S, n, m, d, ld, \
                               d_Us, d_Vs, add, cutoff, d_mean, d_var, d_prob);                    \
        else                                                                                       \
            hipLaunchKernelGGL((k_sample_stats<NQ, false>), grid, dim3(256), smem, ctx->stream, S, n, m, d, ld, \
                               d_Us, d_Vs, add, cutoff, d_mean, d_var, d_prob);                    \
    } break;
    switch (nq) {
        APMF_K4(1) APMF_K4(2) APMF_K4(3) APMF_K4(4) APMF_K4(5) APMF_K4(6) APMF_K4(7) APMF_K4(8)
        APMF_K4(9) APMF_K4(10) APMF_K4(11) APMF_K4(12) APMF_K4(13) APMF_K4(14) APMF_K4(15) APMF_K4(16)
    default:
        apmf_set_error("latent_d=%d unsupported", d);
        return APMF_EINVAL;
    }
#undef APMF_K4
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}
