// k3_gibbs.hip - Bayesian-PMF Gibbs row updates: one half-sweep of samples()
// (bayes_pmf.py:280-296) = `rows` independent sample_feature() calls (bayes_pmf.py:189-216),
// float64 like the reference.
//
// Per row t:  F = oth[idx(t)]  (gathered n_t x d),
//     A    = alpha + beta F'F                     (d x d normal-equation build)
//     cov  = A^-1,  mean = cov (beta F'(r - sub) + alpha mu)
//     out  = chol(cov) z + mean                   (LOWER Cholesky of the covariance, as the
//                                                  reference draws it: bayes_pmf.py:214-216)
// One wave (64-thread block) owns a row.  The normal-equation build F'F runs on the matrix
// cores: v_mfma_f64_16x16x4_f64 tiles with the latent dimension zero-padded to 16 (this is the
// one place of the path where MFMA fits: a rank-4 update of a d x d accumulator per
// instruction), fed straight from coalesced gathers of the other side's rows.  The d x d
// factorisations then run in LDS, one lane per matrix row.  d^3 work per row is tiny: the
// sweep is bound by the gather (nnz*(8+8d) bytes) and by launch latency at MovieLens size.
#include "common.h"


namespace {


// in-place lower Cholesky of the symmetric matrix held in LDS (leading dim lda); lane = row.
// Returns false (wave-uniform) if a pivot is not positive.
__device__ __forceinline__ bool chol_lds(double *A, int d, int lda, int lane)
{
    bool ok = true;
    for (int j = 0; j < d; j++) {
        const double piv = A[j * lda + j];
        if (!(piv > 0.)) { ok = false; break; }  // same value in every lane: uniform
        const double s = sqrt(piv);
        __syncthreads();
        if (lane == j) A[j * lda + j] = s;
        if (lane > j && lane < d) A[lane * lda + j] /= s;
        __syncthreads();
        if (lane > j && lane < d) {
            const double lij = A[lane * lda + j];
            for (int c = j + 1; c <= lane; c++) A[lane * lda + c] -= lij * A[c * lda + j];
        }
        __syncthreads();
    }
    return ok;
}

// lane `src` 's value of a wave-resident double, as a wave-uniform value
__device__ __forceinline__ double lane_value(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// The same factorisation for d <= 16 with the matrix in REGISTERS: lane t holds row t (entries above
// the diagonal are don't-care), other lanes' entries arrive by v_readlane.  Same operations on every
// element in the same order as chol_lds - whose inner loop is a chain of dependent LDS round trips,
// 22 us per 15 x 15 matrix and two thirds of a half-sweep - in about 1.5 us.  Rows >= d must hold the
// identity.  Returns false (wave-uniform) if a pivot is not positive.
__device__ __forceinline__ bool chol_regs16(double (&r)[16], int lane)
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double piv = lane_value(r[j], j);
        ok = ok && piv > 0.;
        const double s = sqrt(piv);
        r[j] = lane == j ? s : r[j] / s;
#pragma unroll
        for (int c = j + 1; c < 16; c++) r[c] -= r[j] * lane_value(r[j], c);
    }
    return ok;
}

using f64x4 = __attribute__((ext_vector_type(4))) double;

// per-hypothesis layout of a batched half-sweep (all zero / NULL for the plain one)
struct GibbsBatch {
    size_t oth_stride;      // doubles between the hypotheses' other-side feature matrices
    size_t row_stride;      // rows * d: between their z and out arrays
    const double *mean_sub; // (H) or NULL: the scalar argument
    const int *ei, *ej;     // (H) extra rating, or NULL
    const double *ev;
    int side;
};

// TL = ceil(d / 16): F'F is built as TL x TL tiles of v_mfma_f64_16x16x4_f64 (upper triangle
// only).  For the tile (ti, tj) and 4 consecutive ratings r0..r0+3, lane l supplies
// F[r0 + (l >> 4)][16 t + (l & 15)] - the same element pattern for the A and the B operand - so
// one coalesced load per lane per tile column feeds the matrix core, and the latent dimension
// is simply zero-padded to the 16-wide tile (d = 15 -> one tile, d = 64 -> 10 tiles).
template <int TL>
__global__ __launch_bounds__(64) void k_gibbs_rows(int row_lo, int rows, int d, const int64_t *__restrict__ ptr,
                                                   const int32_t *__restrict__ idx,
                                                   const double *__restrict__ val,
                                                   const double *__restrict__ oth,
                                                   const double *__restrict__ alpha,
                                                   const double *__restrict__ mu, double beta,
                                                   double mean_sub, const double *__restrict__ z,
                                                   double *__restrict__ out, int *__restrict__ status,
                                                   const GibbsBatch gb)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const int row = row_lo + blockIdx.x;
    if (row >= rows) return;
    // hypothesis of a batched half-sweep (blockIdx.y): its own operands, and ONE rating on top of the
    // shared container - taken as the last entry of its row, where add_rating() + a rebuilt container
    // would put it (bayes_pmf.py:490-500: deepcopy, add_rating, samples)
    const int h = blockIdx.y;
    oth += (size_t)h * gb.oth_stride;
    alpha += (size_t)h * d * d;
    mu += (size_t)h * d;
    z += (size_t)h * gb.row_stride;
    out += (size_t)h * gb.row_stride;
    if (gb.mean_sub) mean_sub = gb.mean_sub[h];
    int xsrc = -1;
    double xval = 0.;
    if (gb.ei) {
        const int own = gb.side == 0 ? gb.ei[h] : gb.ej[h];
        if (own == row) { xsrc = gb.side == 0 ? gb.ej[h] : gb.ei[h]; xval = gb.ev[h]; }
    }
    const int lda = d + 1;
    double *A = smem;                     // d x lda : A, then its Cholesky factor, then chol(cov)
    double *X = A + d * lda;              // d x lda : L^-1, then cov
    double *rhs = X + d * lda;            // d
    double *mean = rhs + d;               // d

    const int q = lane >> 4, c = lane & 15;   // rating slot within a group of 4, column in a tile
    f64x4 acc[TL][TL];
#pragma unroll
    for (int ti = 0; ti < TL; ti++)
#pragma unroll
        for (int tj = 0; tj < TL; tj++) acc[ti][tj] = (f64x4){0., 0., 0., 0.};
    double racc[TL];                       // sum_r F[r][16 t + c] * (rating_r - sub), this lane's slot
#pragma unroll
    for (int t = 0; t < TL; t++) racc[t] = 0.;

    const int64_t beg = ptr[row], end0 = ptr[row + 1];
    const int64_t end = end0 + (xsrc >= 0 ? 1 : 0);
    // 4 MFMA steps (16 ratings) per trip; every load is unconditional (entries past the end are read at
    // the row's first entry and masked) so that a trip's 4 index loads, then its 4 x TL gathers, are in
    // flight together, and the next trip's indices are requested before this trip's gathers are used
    constexpr int UR = 4;
    int32_t src[UR];
    double rv[UR];
    bool ok[UR];
    auto entries = [&](int64_t r0) {
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const int64_t r = r0 + 4 * u + q;
            ok[u] = r < end;
            const bool ex = ok[u] && r == end0;          // only reachable with an extra rating
            const int64_t rc = (ok[u] && !ex) ? r : beg;  // beg < end0: no row is empty (checked by the host)
            const int32_t si = idx[rc];
            const double vi = val[rc];
            src[u] = ex ? xsrc : (ok[u] ? si : 0);
            rv[u] = ok[u] ? (ex ? xval : vi) - mean_sub : 0.;
        }
    };
    entries(beg);
    for (int64_t r0 = beg; r0 < end; r0 += 4 * UR) {
        double f[UR][TL];
#pragma unroll
        for (int u = 0; u < UR; u++)
#pragma unroll
            for (int t = 0; t < TL; t++) {
                const int col = 16 * t + c;
                f[u][t] = oth[(size_t)src[u] * d + (col < d ? col : 0)];
            }
        double rvc[UR];
        bool okc[UR];
#pragma unroll
        for (int u = 0; u < UR; u++) { rvc[u] = rv[u]; okc[u] = ok[u]; }
        entries(r0 + 4 * UR);
#pragma unroll
        for (int u = 0; u < UR; u++) {
#pragma unroll
            for (int t = 0; t < TL; t++) {
                const int col = 16 * t + c;
                f[u][t] = (okc[u] && col < d) ? f[u][t] : 0.;
                racc[t] += f[u][t] * rvc[u];
            }
#pragma unroll
            for (int ti = 0; ti < TL; ti++)
#pragma unroll
                for (int tj = ti; tj < TL; tj++)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[u][ti], f[u][tj], acc[ti][tj], 0, 0, 0);
        }
    }
    // A = alpha + beta F'F : C/D layout of the f64 MFMA is col = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
    for (int ti = 0; ti < TL; ti++)
#pragma unroll
        for (int tj = ti; tj < TL; tj++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int i = 16 * ti + q + 4 * reg, j = 16 * tj + c;
                if (i < d && j < d) {
                    const double v = beta * acc[ti][tj][reg];
                    A[i * lda + j] = alpha[i * d + j] + v;
                    if (ti != tj) A[j * lda + i] = alpha[j * d + i] + v;
                }
            }
    // rhs = beta F'r + alpha mu : combine the 4 rating slots
#pragma unroll
    for (int t = 0; t < TL; t++) {
        double v = racc[t];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int k = 16 * t + c;
        if (q == 0 && k < d) {
            double am = 0.;
            for (int l = 0; l < d; l++) am += alpha[k * d + l] * mu[l];
            rhs[k] = beta * v + am;
        }
    }
    __syncthreads();

    if (TL == 1) {
        // d <= 16: everything below in registers, lane t = row t (or column t); rows >= d are identity
        double r[16];
#pragma unroll
        for (int k = 0; k < 16; k++) r[k] = (lane < d && k < d) ? A[lane * lda + k] : (k == lane ? 1. : 0.);
        // A = R R'
        if (!chol_regs16(r, lane)) {
            if (lane == 0) atomicExch(status, 1);
            return;
        }
        // X = R^-1 (lower): lane c owns column c, x[i] = X[i][c]
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            double v = i == lane ? 1. : 0.;
#pragma unroll
            for (int m = 0; m < i; m++) v -= lane_value(r[m], i) * x[m];      // x[m] = 0 for m < c
            x[i] = i < lane ? 0. : v / lane_value(r[i], i);
        }
        // cov = X' X: lane a holds row a, cov[a][b] = sum_r X[r][a] X[r][b] (terms below max(a, b) are zero)
        double cv[16];
#pragma unroll
        for (int b = 0; b < 16; b++) {
            double sum = 0.;
#pragma unroll
            for (int k = 0; k < 16; k++) sum += x[k] * lane_value(x[k], b);
            cv[b] = sum;
        }
        // mean = cov rhs
        const double rl = lane < d ? rhs[lane] : 0.;
        double mn = 0.;
#pragma unroll
        for (int l = 0; l < 16; l++) mn += cv[l] * lane_value(rl, l);
        // lower Cholesky of the covariance (identity outside d), then the draw
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (lane >= d || k >= d) cv[k] = k == lane ? 1. : 0.;
        if (!chol_regs16(cv, lane)) {
            if (lane == 0) atomicExch(status, 1);
            return;
        }
        const double zl = lane < d ? z[(size_t)row * d + lane] : 0.;
        double sum = 0.;
#pragma unroll
        for (int l = 0; l < 16; l++) {
            const double t = fma(cv[l], lane_value(zl, l), sum);
            sum = l <= lane ? t : sum;
        }
        if (lane < d) out[(size_t)row * d + lane] = sum + mn;
        return;
    }
    // A = R R'
    if (!chol_lds(A, d, lda, lane)) {
        if (lane == 0) atomicExch(status, 1);
        return;
    }
    // X = R^-1 (lower): lane c solves R x = e_c
    if (lane < d) {
        const int c = lane;
        for (int r = 0; r < c; r++) X[r * lda + c] = 0.;
        X[c * lda + c] = 1. / A[c * lda + c];
        for (int r = c + 1; r < d; r++) {
            double s = 0.;
            for (int k = c; k < r; k++) s += A[r * lda + k] * X[k * lda + c];
            X[r * lda + c] = -s / A[r * lda + r];
        }
    }
    __syncthreads();
    // cov = X' X  -> into A (lane = row a); symmetric by construction
    if (lane < d) {
        const int a = lane;
        for (int b = 0; b < d; b++) {
            double s = 0.;
            for (int r = max(a, b); r < d; r++) s += X[r * lda + a] * X[r * lda + b];
            A[a * lda + b] = s;
        }
    }
    __syncthreads();
    if (lane < d) {
        double s = 0.;
        for (int l = 0; l < d; l++) s += A[lane * lda + l] * rhs[l];
        mean[lane] = s;
    }
    __syncthreads();
    // lower Cholesky of the covariance, then the draw
    if (!chol_lds(A, d, lda, lane)) {
        if (lane == 0) atomicExch(status, 1);
        return;
    }
    if (lane < d) {
        double s = 0.;
        for (int l = 0; l <= lane; l++) s += A[lane * lda + l] * z[(size_t)row * d + l];
        out[(size_t)row * d + lane] = s + mean[lane];
    }
}

size_t gibbs_smem(int d)
{
    return sizeof(double) * (2 * (size_t)d * (d + 1) + 2 * d);
}

// ---- optional device-side N(0,1) draws ---------------------------
// Philox4x32-10 counter-based generator + Box-Muller, two doubles per counter.  NOT the
// reference's random stream (NumPy's MT19937): BayesianPMF only uses it when device_rng is
// switched on, trading stream parity for not generating and uploading (n+m) d normals on the
// host per half-sweep pair.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void k_fill_normal(double *__restrict__ out, int64_t count, uint64_t seed,
                                                     uint64_t offset)
{
    const int64_t pair = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (2 * pair >= count) return;
    const uint64_t ctr = offset + (uint64_t)pair;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x243F6A88u, 0x85A308D3u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    // 53-bit uniforms in (0, 1]
    const uint64_t a = (((uint64_t)c[0] << 32) | c[1]) >> 11, b = (((uint64_t)c[2] << 32) | c[3]) >> 11;
    const double u1 = ((double)a + 1.) * (1. / 9007199254740992.), u2 = ((double)b + 1.) * (1. / 9007199254740992.);
    const double rad = sqrt(-2. * log(u1)), ang = 6.283185307179586476925 * u2;
    out[2 * pair] = rad * cos(ang);
    if (2 * pair + 1 < count) out[2 * pair + 1] = rad * sin(ang);
}

} // namespace

static int gibbs_rows(apmf_ctx *ctx, const apmf_ratings *r, int side, int row_lo, int row_hi,
                      const double *d_oth, int d, const double *d_alpha, const double *d_mu, double beta,
                      double mean_sub, const double *d_z, double *d_out, int H = 1,
                      const double *d_mean_sub = nullptr, const int *d_ei = nullptr, const int *d_ej = nullptr,
                      const double *d_ev = nullptr);

extern "C" int apmf_gibbs_half_sweep(apmf_ctx *ctx, const apmf_ratings *r, int side, const double *d_oth,
                                     int d, const double *d_alpha, const double *d_mu, double beta,
                                     double mean_sub, const double *d_z, double *d_out)
{
    APMF_REQUIRE(ctx && r, "ctx/ratings is NULL");
    APMF_REQUIRE(side == 0 || side == 1, "side must be 0 (users) or 1 (items)");
    return gibbs_rows(ctx, r, side, 0, side == 0 ? r->n : r->m, d_oth, d, d_alpha, d_mu, beta, mean_sub,
                      d_z, d_out);
}

extern "C" int apmf_gibbs_half_sweep_rows(apmf_ctx *ctx, const apmf_ratings *r, int side, int row_lo,
                                          int row_hi, const double *d_oth, int d, const double *d_alpha,
                                          const double *d_mu, double beta, double mean_sub,
                                          const double *d_z, double *d_out)
{
    APMF_REQUIRE(ctx && r, "ctx/ratings is NULL");
    APMF_REQUIRE(side == 0 || side == 1, "side must be 0 (users) or 1 (items)");
    const int rows = side == 0 ? r->n : r->m;
    APMF_REQUIRE(0 <= row_lo && row_lo <= row_hi && row_hi <= rows, "row range [%d,%d) outside [0,%d)",
                 row_lo, row_hi, rows);
    return gibbs_rows(ctx, r, side, row_lo, row_hi, d_oth, d, d_alpha, d_mu, beta, mean_sub, d_z, d_out);
}

static int gibbs_rows(apmf_ctx *ctx, const apmf_ratings *r, int side, int row_lo, int row_hi,
                      const double *d_oth, int d, const double *d_alpha, const double *d_mu, double beta,
                      double mean_sub, const double *d_z, double *d_out, int H, const double *d_mean_sub,
                      const int *d_ei, const int *d_ej, const double *d_ev)
{
    APMF_REQUIRE(ctx && r, "ctx/ratings is NULL");
    APMF_REQUIRE(side == 0 || side == 1, "side must be 0 (users) or 1 (items)");
    APMF_REQUIRE(d >= 1 && d <= APMF_MAX_D, "latent_d=%d unsupported (1..%d)", d, APMF_MAX_D);
    APMF_REQUIRE(d_oth && d_alpha && d_mu && d_z && d_out, "NULL operand");
    // bayes_pmf.py:287,296: items_by_user[user_id] raises KeyError for a row without ratings
    APMF_REQUIRE(!r->has_empty[side], "a %s has no rating: the Gibbs sweep needs >= 1 per row",
                 side == 0 ? "user" : "item");
    APMF_REQUIRE(H >= 1 && H <= 65535, "bad hypothesis count %d", H);
    if (row_hi == row_lo) return APMF_OK;
    GibbsBatch gb;
    const int rows_own = side == 0 ? r->n : r->m, rows_oth = side == 0 ? r->m : r->n;
    gb.oth_stride = H > 1 ? (size_t)rows_oth * d : 0;
    gb.row_stride = H > 1 ? (size_t)rows_own * d : 0;
    gb.mean_sub = d_mean_sub;
    gb.ei = d_ei; gb.ej = d_ej; gb.ev = d_ev;
    gb.side = side;
    void *w;
    APMF_TRY(apmf_ws(ctx, 2, 64, &w));
    int *d_status = (int *)w;
    APMF_CHECK_HIP(hipMemsetAsync(d_status, 0, sizeof(int), ctx->stream));
    const int tl = (d + 15) / 16;
    const size_t smem = gibbs_smem(d);
#define LAUNCH_GIBBS(TLV)                                                                         \
    hipLaunchKernelGGL((k_gibbs_rows<TLV>), dim3(row_hi - row_lo, H), dim3(64), smem, ctx->stream, \
                       row_lo, row_hi, d, r->ptr[side], r->idx[side], r->val[side], d_oth, d_alpha, \
                       d_mu, beta, mean_sub, d_z, d_out, d_status, gb)
    if (smem > 64 * 1024)
        APMF_CHECK_HIP(hipFuncSetAttribute((const void *)k_gibbs_rows<4>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    {
        ProfScope prof(ctx);
        if (tl == 1) LAUNCH_GIBBS(1);
        else if (tl == 2) LAUNCH_GIBBS(2);
        else if (tl == 3) LAUNCH_GIBBS(3);
        else LAUNCH_GIBBS(4);
    }
#undef LAUNCH_GIBBS
    APMF_CHECK_LAUNCH();
    int status = 0;
    APMF_TRY(apmf_d2h(ctx, &status, d_status, sizeof(int)));
    if (status) {
        apmf_set_error("alpha + beta F'F (or its inverse) is not positive definite");
        return APMF_ENOTPD;
    }
    return APMF_OK;
}

// H half-sweeps in one launch: the hypotheses of BayesianPMF.exp_variance (bayes_pmf.py:457-525, 560-598)
extern "C" int apmf_gibbs_half_sweep_batched(apmf_ctx *ctx, const apmf_ratings *r, int side, int H,
                                             const double *d_oth, int d, const double *d_alpha, const double *d_mu,
                                             double beta, const double *d_mean_sub, const double *d_z, double *d_out,
                                             const int32_t *d_ei, const int32_t *d_ej, const double *d_ev)
{
    APMF_REQUIRE(ctx && r, "ctx/ratings is NULL");
    APMF_REQUIRE(side == 0 || side == 1, "side must be 0 (users) or 1 (items)");
    APMF_REQUIRE(d_mean_sub, "d_mean_sub is NULL");
    APMF_REQUIRE((d_ei == nullptr) == (d_ej == nullptr) && (d_ei == nullptr) == (d_ev == nullptr),
                 "the extra rating needs all of d_ei, d_ej, d_ev");
    return gibbs_rows(ctx, r, side, 0, side == 0 ? r->n : r->m, d_oth, d, d_alpha, d_mu, beta, 0., d_z, d_out, H,
                      d_mean_sub, d_ei, d_ej, d_ev);
}

extern "C" int apmf_fill_normal_f64(apmf_ctx *ctx, double *d_out, int64_t count, uint64_t seed, uint64_t offset)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (count == 0) return APMF_OK;
    APMF_REQUIRE(d_out && count > 0, "bad operand");
    const int64_t pairs = (count + 1) / 2;
    hipLaunchKernelGGL(k_fill_normal, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, ctx->stream, d_out, count,
                       seed, offset);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}
