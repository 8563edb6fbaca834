// k5_klgrad.hip - KL objective and gradient of the matrix-normal variational fit
// (matrix_normal_exps_cy.pyx:159-213 mn_kl_divergence, :216-485 matrixnormal_gradient with
// _quadexp_grad / _squareexp_grad / _mnormal_grad), float64, part of libapmf_dense.so (the
// gradient needs the (n+m) x (n+m) inverse).
//
// The reference loops over ratings x latent pairs (k < l) and scatters ~30 terms per pair.
// With O' = cov_latents (upper triangle mirrored - the only entries the reference reads),
// dg = diag(O'), off = O' - diag, T = tr O', and per ROW of M the constants
//     xoff = x off,   c = (x.xoff + sum_k x_k^2 dg_k) / 2,   q = x' O' x
// every pair sum collapses, and what is left per rating (i, j, r) with a = M_i, b = M_{n+j},
// s = S_ij, vi = S_ii, vj = S_jj (S = cov_useritems) is, for the USER row i:
//     gm_i  += b (P - p) + s boff_j + vj aoff_i + b s (T - dg) + 2 b s dg + a (b^2 + vj dg) - r b
//     gs_ii += c_j + vj (off2 + D2/2)
//     g_S[i, n+j] (= [n+j, i]) = [ a . (b (T + dg) + boff_j) + 2 s (sdd + off2 + D2) - r T ] / sigma^2
//     w_i   += s b,   cvj_i += vj,   s2_i += s^2,   rs_i += r s          (for the d x d part)
// and symmetrically for the ITEM row j (p = a b, P = sum p, off2 = sum_{k<l} O'_kl^2,
// D2 = sum dg^2, sdd = sum_{k<l} dg_k dg_l).  cov_latents' gradient is then a handful of
// d x d contractions over the per-row sums.  Row ownership (one wave per row, CSR order)
// makes every sum deterministic; there are no atomics.
//
// Reference quirks kept on purpose: mn_kl_divergence's `num_items = 0` (the item-covariance
// trace term is dropped) and its use of sigma_u_sq for the item prior (:176, :191-197).
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>

#include "common.h"

extern "C" int apmf_dense_inv_logdet(apmf_ctx *ctx, const double *d_mat, int N, double *d_inv,
                                     double *h_logdet);

namespace {

constexpr int UN5 = 2;

template <int CTRL>
__device__ __forceinline__ double dpp5(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int DP>
__device__ __forceinline__ double gsum(double v)
{
    if (DP >= 2) v += dpp5<0xB1>(v);
    if (DP >= 4) v += dpp5<0x4E>(v);
    if (DP >= 8) v += dpp5<0x141>(v);
    if (DP >= 16) v += dpp5<0x140>(v);
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

// scalars derived from cov_latents, computed once per call
struct OmegaConsts {
    double T, D2, off2, sdd, t2;
};

// per-row constants: xoff (N x d), c (N), q (N), diag(S) (N); block 0 thread 0 also fills oc
__global__ __launch_bounds__(256) void k_kl_rowconst(int N, int d, const double *__restrict__ M,
                                                     const double *__restrict__ S,
                                                     const double *__restrict__ O, double *__restrict__ xoff,
                                                     double *__restrict__ crow, double *__restrict__ qrow,
                                                     double *__restrict__ sdiag, OmegaConsts *__restrict__ oc)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) {
        double T = 0, D2 = 0, off2 = 0;
        for (int k = 0; k < d; k++) {
            T += O[k * d + k];
            D2 += O[k * d + k] * O[k * d + k];
            for (int l = k + 1; l < d; l++) off2 += O[k * d + l] * O[k * d + l];
        }
        oc->T = T; oc->D2 = D2; oc->off2 = off2; oc->sdd = (T * T - D2) / 2; oc->t2 = D2 + 2 * off2;
    }
    if (t >= N) return;
    const double *x = M + (size_t)t * d;
    double xo_dot = 0, x2dg = 0;
    for (int k = 0; k < d; k++) {
        double w = 0;
        for (int l = 0; l < d; l++)
            if (l != k) w += O[l < k ? l * d + k : k * d + l] * x[l];
        xoff[(size_t)t * d + k] = w;
        xo_dot += x[k] * w;
        x2dg += x[k] * x[k] * O[k * d + k];
    }
    crow[t] = (xo_dot + x2dg) / 2;
    qrow[t] = xo_dot + x2dg;
    sdiag[t] = S[(size_t)t * N + t];
}

struct KlDims {
    int n, m, d, N;
};

// One wave per row.  SIDE 0: user rows (also writes the per-rating cross entries of g_S and
// accumulates the sums cov_latents' gradient needs); SIDE 1: item rows.
// acc layout per row: users [gm (d) | w (d) | gs_ii, cvj, s2, rs]; items [gm (d) | gs_jj, cvi]
template <int DP, int SIDE>
__global__ __launch_bounds__(256) void k_kl_rows(const KlDims q, const int64_t *__restrict__ ptr,
                                                 const int32_t *__restrict__ idx,
                                                 const double *__restrict__ val,
                                                 const double *__restrict__ M,
                                                 const double *__restrict__ S,
                                                 const double *__restrict__ xoff,
                                                 const double *__restrict__ crow,
                                                 const double *__restrict__ sdiag,
                                                 const double *__restrict__ dgv,
                                                 const OmegaConsts *__restrict__ ocp, double inv_sig,
                                                 double *__restrict__ gS, double *__restrict__ racc)
{
    constexpr int G = 64 / DP;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave_id_uniform();
    const int rows = SIDE == 0 ? q.n : q.m;
    if (row >= rows) return;
    const int d = q.d, N = q.N;
    const OmegaConsts oc = *ocp;
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    const int self = SIDE == 0 ? row : q.n + row;
    const double xk = kok ? M[(size_t)self * d + k] : 0.;        // a_k (users) / b_k (items)
    const double xok = kok ? xoff[(size_t)self * d + k] : 0.;     // own xoff
    const double dgk = kok ? dgv[k] : 0.;
    const double vself = sdiag[self];
    const int64_t beg = ptr[row], end = ptr[row + 1];
    double gm = 0., wv = 0., gs_d = 0., cv = 0., s2 = 0., rs = 0.;
    for (int64_t t0 = beg + g; t0 < end; t0 += (int64_t)G * UN5) {
#pragma unroll
        for (int u = 0; u < UN5; u++) {
            const int64_t t = t0 + (int64_t)u * G;
            const bool ok = t < end;
            const int c = ok ? idx[t] : 0;
            const double r = ok ? val[t] : 0.;
            const int other = SIDE == 0 ? q.n + c : c;
            const double yk = (ok && kok) ? M[(size_t)other * d + k] : 0.;
            const double yok = (ok && kok) ? xoff[(size_t)other * d + k] : 0.;
            const double voth = ok ? sdiag[other] : 0.;
            const double coth = ok ? crow[other] : 0.;
            const double s = ok ? (SIDE == 0 ? S[(size_t)row * N + other] : S[(size_t)other * N + self]) : 0.;
            const double p = xk * yk;
            const double P = gsum<DP>(p);
            if (ok) {
                // mean gradient of the own row (see the header): y = other side's vector
                gm += yk * (P - p) + s * yok + voth * xok + yk * s * (oc.T - dgk) + 2. * yk * s * dgk
                      + xk * (yk * yk + voth * dgk) - r * yk;
                gs_d += coth + voth * (oc.off2 + oc.D2 / 2.);
                cv += voth;
            }
            if (SIDE == 0) {
                // a . (b (T + dg) + boff_j)
                const double cross = gsum<DP>(xk * (yk * (oc.T + dgk) + yok));
                if (ok) {
                    wv += s * yk;
                    s2 += s * s;
                    rs += r * s;
                    if (k == 0) {
                        const double gij = inv_sig * (cross + 2. * s * (oc.sdd + oc.off2 + oc.D2) - r * oc.T);
                        gS[(size_t)row * N + other] += gij;
                        gS[(size_t)other * N + row] += gij;
                    }
                }
            }
        }
    }
    gm = xgsum<DP>(gm);
    wv = xgsum<DP>(wv);
    // the scalar sums were accumulated identically in every lane of a group
    gs_d = xgsum<DP>(gs_d); cv = xgsum<DP>(cv); s2 = xgsum<DP>(s2); rs = xgsum<DP>(rs);
    const int stride = SIDE == 0 ? 2 * d + 4 : d + 2;
    double *out = racc + (size_t)row * stride;
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
    (void)vself;
}

// g_S <- -(d/2) * inv * (2 - delta)   (the entropy term, matrix_normal_exps_cy.pyx:477-480);
// the sparse rating terms are added on top by k_kl_rows / k_kl_finish_rows
__global__ __launch_bounds__(256) void k_entropy_term(double *__restrict__ g, int N, double half_mult)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)N * N) return;
    const int i = (int)(t / N), j = (int)(t % N);
    g[t] = -half_mult * g[t] * (i == j ? 1. : 2.);
}

// g_mean rows and the diagonal of g_S from the per-row sums
__global__ __launch_bounds__(256) void k_kl_finish_rows(const KlDims q, const double *__restrict__ M,
                                                        const double *__restrict__ racc_u,
                                                        const double *__restrict__ racc_v, double inv_sig,
                                                        double sig_u, double sig_v, double trO,
                                                        double *__restrict__ gmean, double *__restrict__ gS)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= q.N) return;
    const int d = q.d;
    const bool user = t < q.n;
    const double *acc = user ? racc_u + (size_t)t * (2 * d + 4) : racc_v + (size_t)(t - q.n) * (d + 2);
    const double prior = user ? sig_u : sig_v;
    for (int k = 0; k < d; k++)
        gmean[(size_t)t * d + k] = inv_sig * acc[k] + M[(size_t)t * d + k] / prior;
    const double gs = user ? acc[2 * d] : acc[d];
    gS[(size_t)t * q.N + t] += inv_sig * gs + trO / (2. * prior);
}

// cov_latents' gradient: one block per (k, l); fixed-order sums over the rows
__global__ __launch_bounds__(256) void k_kl_omega(const KlDims q, const double *__restrict__ M,
                                                  const double *__restrict__ racc_u,
                                                  const double *__restrict__ racc_v,
                                                  const double *__restrict__ sdiag,
                                                  const double *__restrict__ O,
                                                  const double *__restrict__ Oinv,
                                                  const OmegaConsts *__restrict__ ocp, double inv_sig,
                                                  double sig_u, double sig_v, double *__restrict__ gO)
{
    const int d = q.d, n = q.n, m = q.m;
    const int k = blockIdx.x / d, l = blockIdx.x % d;
    const OmegaConsts oc = *ocp;
    __shared__ double sh[6][256];
    // per-thread partials: G_kl pieces and the scalars SVV, S2, RS, AW (all blocks recompute
    // the scalars: they are cheap and it keeps the kernel single-pass)
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
    // X = sum_i M[i][k] w_i[k] of the diagonal entries, added in row order by one thread: the operands
    // are fetched by the whole block, 256 rows at a time, and handed over through LDS
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
                for (int q2 = 0; q2 < cnt; q2++) X += sh[0][q2] * sh[1][q2];
            }
        }
    }
    if (threadIdx.x != 0) return;
    const double okl = O[k <= l ? k * d + l : l * d + k];
    double out;
    if (k != l) {
        out = inv_sig * (G + 2. * okl * (SVV + S2));
    } else {
        // G on the diagonal holds  2 sum_i a_k w_k + sum_i cvj a_k^2 + sum_j cvi b_k^2
        const double dgk = O[k * d + k];
        const double sum_awk_x2_plus_sq = G;   // = 2 * sum a_k w_k + (cvj a_k^2 + cvi b_k^2 sums)
        // diag_q = AW - sum_i a_k w_k + S2 (T - dg);  diag_s = (4 sum a_k w_k + 4 S2 dg + sq + 2 SVV dg) / 2
        // with sum a_k w_k =: X and sq =: Q (G = 2X + Q) we need X and Q separately:
        // (X: see above)
        const double Q = sum_awk_x2_plus_sq - 2. * X;
        const double diag_q = AW - X + S2 * (oc.T - dgk);
        const double diag_s = .5 * (4. * X + 4. * S2 * dgk + Q + 2. * SVV * dgk);
        out = inv_sig * (diag_q + diag_s - RS) + TRU / (2. * sig_u) + TRV / (2. * sig_v);
    }
    // entropy term: - N/2 (inv + inv' (1 - I))   (matrix_normal_exps_cy.pyx:482-485)
    const double ik = Oinv[k * d + l], ikt = Oinv[l * d + k];
    out -= q.N / 2. * (ik + (k == l ? 0. : ikt));
    gO[k * d + l] = out;
}

// per-user partial of  sum_t [ E[(u.v)^2] - 2 r E[u.v] + r^2 ]   (matrix_normal_exps_cy.pyx:200-211)
template <int DP>
__global__ __launch_bounds__(256) void k_kl_value_rows(const KlDims q, const int64_t *__restrict__ ptr,
                                                       const int32_t *__restrict__ idx,
                                                       const double *__restrict__ val,
                                                       const double *__restrict__ M,
                                                       const double *__restrict__ S,
                                                       const double *__restrict__ xoff,
                                                       const double *__restrict__ qrow,
                                                       const double *__restrict__ sdiag,
                                                       const double *__restrict__ dgv,
                                                       const OmegaConsts *__restrict__ ocp,
                                                       double *__restrict__ part /* (n, 3) */)
{
    constexpr int G = 64 / DP;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave_id_uniform();
    if (row >= q.n) return;
    const int d = q.d, N = q.N;
    const OmegaConsts oc = *ocp;
    const int k = lane % DP, g = lane / DP;
    const bool kok = k < d;
    const double ak = kok ? M[(size_t)row * d + k] : 0.;
    const double oak = kok ? xoff[(size_t)row * d + k] + dgv[k] * ak : 0.;   // (O' a)_k
    const double vi = sdiag[row], aOa = qrow[row];
    double bit = 0., un = 0.;
    if (g == 0) un = ak * ak;
    const int64_t beg = ptr[row], end = ptr[row + 1];
    for (int64_t t = beg + g; t < end; t += G) {
        const int j = q.n + idx[t];
        const double r = val[t];
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
        part[(size_t)row * 3] = bit;
        part[(size_t)row * 3 + 1] = un;
        part[(size_t)row * 3 + 2] = vi;
    }
}

// out[0] = sum bit, out[1] = ||M_users||^2, out[2] = tr(S_users), out[3] = ||M_items||^2
__global__ __launch_bounds__(256) void k_kl_value_finish(const KlDims q, const double *__restrict__ part,
                                                         const double *__restrict__ M,
                                                         double *__restrict__ out)
{
    __shared__ double sh[4][256];
    double a = 0, b = 0, c = 0, e = 0;
    for (int t = threadIdx.x; t < q.n; t += 256) {
        a += part[(size_t)t * 3]; b += part[(size_t)t * 3 + 1]; c += part[(size_t)t * 3 + 2];
    }
    for (size_t t = threadIdx.x; t < (size_t)q.m * q.d; t += 256) {
        const double v = M[(size_t)q.n * q.d + t];
        e += v * v;
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b; sh[2][threadIdx.x] = c; sh[3][threadIdx.x] = e;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int z = 0; z < 4; z++) sh[z][threadIdx.x] += sh[z][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int z = 0; z < 4; z++) out[z] = sh[z][0];
}

__global__ void k_extract_diag(int d, const double *__restrict__ O, double *__restrict__ dgv)
{
    const int k = threadIdx.x;
    if (k < d) dgv[k] = O[k * d + k];
}

int dp5_for(int d)
{
    int dp = 4;
    while (dp < d) dp <<= 1;
    return dp;
}

#define DISPATCH_DP5(dp, ...)                                                  \
    switch (dp) {                                                              \
    case 4: { constexpr int DP = 4; __VA_ARGS__; } break;                      \
    case 8: { constexpr int DP = 8; __VA_ARGS__; } break;                      \
    case 16: { constexpr int DP = 16; __VA_ARGS__; } break;                    \
    case 32: { constexpr int DP = 32; __VA_ARGS__; } break;                    \
    case 64: { constexpr int DP = 64; __VA_ARGS__; } break;                    \
    default: apmf_set_error("latent_d too large"); return APMF_EINVAL;         \
    }

struct Work {
    double *xoff, *crow, *qrow, *sdiag, *dgv, *racc_u, *racc_v, *part, *out4, *Oinv;
    OmegaConsts *oc;
};

int carve(apmf_ctx *ctx, int n, int m, int d, Work *w)
{
    const size_t N = (size_t)n + m;
    const size_t doubles = N * d + 3 * N + 64 + (size_t)n * (2 * d + 4) + (size_t)m * (d + 2) + (size_t)n * 3 +
                           16 + (size_t)d * d + 16;
    void *p;
    APMF_TRY(apmf_ws(ctx, 2, doubles * sizeof(double) + 256, &p));
    double *b = (double *)p;
    w->xoff = b; b += N * d;
    w->crow = b; b += N;
    w->qrow = b; b += N;
    w->sdiag = b; b += N;
    w->dgv = b; b += 64;
    w->racc_u = b; b += (size_t)n * (2 * d + 4);
    w->racc_v = b; b += (size_t)m * (d + 2);
    w->part = b; b += (size_t)n * 3;
    w->out4 = b; b += 16;
    w->Oinv = b; b += (size_t)d * d;
    w->oc = (OmegaConsts *)b;
    return APMF_OK;
}

int row_constants(apmf_ctx *ctx, const KlDims &q, const double *M, const double *S, const double *O,
                  const Work &w)
{
    hipLaunchKernelGGL(k_extract_diag, dim3(1), dim3(64), 0, ctx->stream, q.d, O, w.dgv);
    hipLaunchKernelGGL(k_kl_rowconst, dim3((q.N + 255) / 256), dim3(256), 0, ctx->stream, q.N, q.d, M, S, O,
                       w.xoff, w.crow, w.qrow, w.sdiag, w.oc);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

// out = x - lr * g with NumPy's two roundings (no fused multiply-add): the trial point of the
// learning-rate search
__global__ __launch_bounds__(256) void k_trial_point(size_t count, const double *__restrict__ x,
                                                     const double *__restrict__ g, double lr,
                                                     double *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    out[i] = __dsub_rn(x[i], __dmul_rn(lr, g[i]));
}

// ---- full-covariance variational fit: gradient of the KL w.r.t. (mean, cov) ----------------
// normal_exps_cy.pyx:140-303 (ActivePMF.fit_normal).  The latent vector is stacked, u[k, i] =
// i d + k, v[k, j] = n d + j d + k (active_pmf.py:141-142), so a rating (i, j) touches the
// d x d blocks UU = cov[U_i, U_i], UV = cov[U_i, V_j], VV = cov[V_j, V_j].  One wave per user
// row (SIDE 0) accumulates the mean gradient of U_i and the UU block over the user's ratings in
// storage order and writes the per-rating UV / VU blocks; one wave per item row (SIDE 1) does
// the same for V_j and VV.  No atomics.  The reference's arithmetic is kept, including the
// summed increments of _cov_4exp_grad_clcl (:188-197) and of sum_tripexpect_ccl (:239-249)
// that its own notebook flags as a bug.
constexpr int FG_D = 32;   // largest latent_d of the full-covariance path (K = (n+m) d is the limit anyway)

template <int SIDE>
__global__ __launch_bounds__(64) void k_full_grad_rows(int n, int m, int d, const int64_t *__restrict__ ptr,
                                                       const int32_t *__restrict__ idx,
                                                       const double *__restrict__ val,
                                                       const double *__restrict__ mean,
                                                       const double *__restrict__ cov, double inv_sig,
                                                       double *__restrict__ gmean, double *__restrict__ gcov)
{
    __shared__ double Cuu[FG_D][FG_D + 1], Cuv[FG_D][FG_D + 1], Cvv[FG_D][FG_D + 1], OWN[FG_D][FG_D + 1];
    __shared__ double sa[FG_D], sb[FG_D], se[FG_D], s1[FG_D], s2[FG_D], sA[FG_D], sB[FG_D], sC[FG_D], sD[FG_D];
    const int x = threadIdx.x;
    const int row = blockIdx.x;
    const size_t K = (size_t)(n + m) * d;
    const int64_t beg = ptr[row], end = ptr[row + 1];
    for (int e = x; e < d * d; e += 64) OWN[e / d][e % d] = 0.;
    double gm = 0.;
    for (int64_t t = beg; t < end; t++) {
        const int i = SIDE == 0 ? row : idx[t], j = SIDE == 0 ? idx[t] : row;
        const double r = val[t];
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
        // the own diagonal block (UU for users, VV for items), accumulated over the ratings
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
            // the cross blocks of this rating are touched by no other rating
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

// g_cov <- -(inv + inv' (1 - I)) / 2 (inv symmetric: -inv off the diagonal, -inv/2 on it), plus the
// prior terms on the diagonal; g_mean <- mean / prior   (normal_exps_cy.pyx:286-303)
__global__ __launch_bounds__(256) void k_full_grad_init(size_t K, size_t nd, double sig_u, double sig_v,
                                                        const double *__restrict__ mean,
                                                        double *__restrict__ gmean, double *__restrict__ gcov)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= K * K) return;
    const size_t i = t / K, j = t % K;
    const double inv = gcov[t], invt = gcov[j * K + i];
    if (i == j) {
        const double prior = i < nd ? sig_u : sig_v;
        gcov[t] = -inv / 2. + 1. / (2. * prior);
        gmean[i] = mean[i] / prior;
    } else if (i < j) {
        const double v = -(inv + invt) / 2.;
        // both halves are written by the thread of the upper one: no read-after-write hazard
        gcov[t] = v;
        gcov[j * K + i] = -(invt + inv) / 2.;
    }
}

// data and prior terms of ActivePMF.kl_divergence (active_pmf.py:202-246) from the swept grids:
// out[0] = sum_ratings (var + mean^2 - 2 r mean + r^2), out[1] = sum_u (mean^2 + cov_ii), out[2] = same over v
__global__ __launch_bounds__(256) void k_full_kl_terms(int n, int m, int d, int ld, const int64_t *__restrict__ ptr,
                                                       const int32_t *__restrict__ idx,
                                                       const double *__restrict__ val,
                                                       const double *__restrict__ pvar,
                                                       const double *__restrict__ pmean,
                                                       const double *__restrict__ mean,
                                                       const double *__restrict__ cov, double *__restrict__ out)
{
    __shared__ double sh[3][256];
    const size_t K = (size_t)(n + m) * d, nd = (size_t)n * d;
    double a = 0., bu = 0., bv = 0.;
    for (int i = threadIdx.x; i < n; i += 256)
        for (int64_t t = ptr[i]; t < ptr[i + 1]; t++) {
            const size_t o = (size_t)i * ld + idx[t];
            const double r = val[t], pm = pmean[o];
            a += pvar[o] + pm * pm - 2. * r * pm + r * r;
        }
    for (size_t k = threadIdx.x; k < K; k += 256) {
        const double v = mean[k] * mean[k] + cov[k * K + k];
        if (k < nd) bu += v;
        else bv += v;
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = bu; sh[2][threadIdx.x] = bv;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int z = 0; z < 3; z++) sh[z][threadIdx.x] += sh[z][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int z = 0; z < 3; z++) out[z] = sh[z][0];
}

} // namespace

extern "C" {

// d_factor (nullable): Cholesky factor of d_cov_useritems left by the KL evaluation of the same matrix
static int mn_kl_gradient_impl(apmf_ctx *ctx, const apmf_ratings *r, int d, const double *d_mean,
                               const double *d_cov_useritems, const double *d_cov_latents, double sigma_sq,
                               double sigma_u_sq, double sigma_v_sq, double *d_g_mean,
                               double *d_g_cov_useritems, double *d_g_cov_latents, const double *d_factor)
{
    APMF_REQUIRE(ctx && r && d_mean && d_cov_useritems && d_cov_latents && d_g_mean && d_g_cov_useritems &&
                     d_g_cov_latents, "NULL operand");
    APMF_REQUIRE(d >= 1 && d <= APMF_MAX_D, "latent_d=%d unsupported (1..%d)", d, APMF_MAX_D);
    APMF_REQUIRE(sigma_sq != 0. && sigma_u_sq != 0. && sigma_v_sq != 0., "zero variance");
    KlDims q = {r->n, r->m, d, r->n + r->m};
    Work w;
    APMF_TRY(carve(ctx, q.n, q.m, d, &w));
    APMF_TRY(row_constants(ctx, q, d_mean, d_cov_useritems, d_cov_latents, w));
    // entropy terms first: g_S <- inverse, then scaled; g_Omega's inverse goes to a scratch
    if (d_factor) APMF_TRY(apmf_dense_inverse_from_factor(ctx, d_factor, q.N, d_g_cov_useritems));
    else APMF_TRY(apmf_dense_inv_logdet(ctx, d_cov_useritems, q.N, d_g_cov_useritems, nullptr));
    APMF_TRY(apmf_dense_inv_logdet(ctx, d_cov_latents, d, w.Oinv, nullptr));
    const size_t nn = (size_t)q.N * q.N;
    hipLaunchKernelGGL(k_entropy_term, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_g_cov_useritems, q.N, d / 2.);
    APMF_CHECK_LAUNCH();
    const double inv_sig = 1. / sigma_sq;
    const int dp = dp5_for(d);
    DISPATCH_DP5(dp, hipLaunchKernelGGL((k_kl_rows<DP, 0>), dim3((q.n + 3) / 4), dim3(256), 0, ctx->stream, q,
                                        r->ptr[0], r->idx[0], r->val[0], d_mean, d_cov_useritems, w.xoff,
                                        w.crow, w.sdiag, w.dgv, w.oc, inv_sig, d_g_cov_useritems, w.racc_u));
    DISPATCH_DP5(dp, hipLaunchKernelGGL((k_kl_rows<DP, 1>), dim3((q.m + 3) / 4), dim3(256), 0, ctx->stream, q,
                                        r->ptr[1], r->idx[1], r->val[1], d_mean, d_cov_useritems, w.xoff,
                                        w.crow, w.sdiag, w.dgv, w.oc, inv_sig, d_g_cov_useritems, w.racc_v));
    APMF_CHECK_LAUNCH();
    // tr(cov_latents) for the prior term of the diagonal (matrix_normal_exps_cy.pyx:452-456)
    OmegaConsts hoc;
    APMF_TRY(apmf_d2h(ctx, &hoc, w.oc, sizeof(hoc)));
    hipLaunchKernelGGL(k_kl_finish_rows, dim3((q.N + 255) / 256), dim3(256), 0, ctx->stream, q, d_mean,
                       w.racc_u, w.racc_v, inv_sig, sigma_u_sq, sigma_v_sq, hoc.T, d_g_mean,
                       d_g_cov_useritems);
    hipLaunchKernelGGL(k_kl_omega, dim3(d * d), dim3(256), 0, ctx->stream, q, d_mean, w.racc_u, w.racc_v,
                       w.sdiag, d_cov_latents, w.Oinv, w.oc, inv_sig, sigma_u_sq, sigma_v_sq, d_g_cov_latents);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

// d_factor_out (nullable): receives the Cholesky factor of d_cov_useritems when *h_have_factor comes back 1
static int mn_kl_value_impl(apmf_ctx *ctx, const apmf_ratings *r, int d, const double *d_mean,
                            const double *d_cov_useritems, const double *d_cov_latents, double sigma_sq,
                            double sigma_u_sq, double sigma_v_sq, double *h_kl, double *d_factor_out,
                            int *h_have_factor, const double *ld_s_known = nullptr)
{
    APMF_REQUIRE(ctx && r && d_mean && d_cov_useritems && d_cov_latents && h_kl, "NULL operand");
    APMF_REQUIRE(d >= 1 && d <= APMF_MAX_D, "latent_d=%d unsupported (1..%d)", d, APMF_MAX_D);
    KlDims q = {r->n, r->m, d, r->n + r->m};
    Work w;
    APMF_TRY(carve(ctx, q.n, q.m, d, &w));
    APMF_TRY(row_constants(ctx, q, d_mean, d_cov_useritems, d_cov_latents, w));
    DISPATCH_DP5(dp5_for(d),
                 hipLaunchKernelGGL((k_kl_value_rows<DP>), dim3((q.n + 3) / 4), dim3(256), 0, ctx->stream, q,
                                    r->ptr[0], r->idx[0], r->val[0], d_mean, d_cov_useritems, w.xoff, w.qrow,
                                    w.sdiag, w.dgv, w.oc, w.part));
    hipLaunchKernelGGL(k_kl_value_finish, dim3(1), dim3(256), 0, ctx->stream, q, w.part, d_mean, w.out4);
    APMF_CHECK_LAUNCH();
    double h4[4];
    OmegaConsts hoc;
    APMF_TRY(apmf_d2h(ctx, h4, w.out4, sizeof(h4)));
    APMF_TRY(apmf_d2h(ctx, &hoc, w.oc, sizeof(hoc)));
    double ld_s = 0, ld_o = 0;
    if (ld_s_known) ld_s = *ld_s_known;     // the caller already factorised this matrix
    else if (d_factor_out) APMF_TRY(apmf_dense_logdet_keep_factor(ctx, d_cov_useritems, q.N, d_factor_out, &ld_s, h_have_factor));
    else APMF_TRY(apmf_dense_inv_logdet(ctx, d_cov_useritems, q.N, nullptr, &ld_s));
    APMF_TRY(apmf_dense_inv_logdet(ctx, d_cov_latents, d, nullptr, &ld_o));
    double kl = -(ld_s * d + ld_o * q.N) / 2.;
    // sic: the item trace term is dropped and the item prior uses sigma_u_sq (:176, :191-197)
    kl += (h4[1] + h4[2] * hoc.T) / (2. * sigma_u_sq);
    kl += (h4[3] + 0. * hoc.T) / (2. * sigma_u_sq);
    kl += h4[0] / (2. * sigma_sq);
    (void)sigma_v_sq;
    *h_kl = kl;
    return APMF_OK;
}

int apmf_mn_kl_gradient(apmf_ctx *ctx, const apmf_ratings *r, int d, const double *d_mean,
                        const double *d_cov_useritems, const double *d_cov_latents, double sigma_sq,
                        double sigma_u_sq, double sigma_v_sq, double *d_g_mean,
                        double *d_g_cov_useritems, double *d_g_cov_latents)
{
    return mn_kl_gradient_impl(ctx, r, d, d_mean, d_cov_useritems, d_cov_latents, sigma_sq, sigma_u_sq,
                               sigma_v_sq, d_g_mean, d_g_cov_useritems, d_g_cov_latents, nullptr);
}

int apmf_mn_kl_value(apmf_ctx *ctx, const apmf_ratings *r, int d, const double *d_mean,
                     const double *d_cov_useritems, const double *d_cov_latents, double sigma_sq,
                     double sigma_u_sq, double sigma_v_sq, double *h_kl)
{
    return mn_kl_value_impl(ctx, r, d, d_mean, d_cov_useritems, d_cov_latents, sigma_sq, sigma_u_sq, sigma_v_sq,
                            h_kl, nullptr, nullptr);
}

int apmf_full_kl_gradient(apmf_ctx *ctx, const apmf_ratings *r, int d, const double *d_mean,
                          const double *d_cov, double sigma_sq, double sigma_u_sq, double sigma_v_sq,
                          double *d_g_mean, double *d_g_cov)
{
    APMF_REQUIRE(ctx && r && d_mean && d_cov && d_g_mean && d_g_cov, "NULL operand");
    APMF_REQUIRE(d >= 1 && d <= FG_D, "latent_d=%d unsupported by the full-covariance gradient (1..%d)", d, FG_D);
    APMF_REQUIRE(sigma_sq != 0. && sigma_u_sq != 0. && sigma_v_sq != 0., "zero variance");
    const size_t K = ((size_t)r->n + r->m) * d;
    APMF_REQUIRE(K <= 46000, "(n+m) d = %zu is too large for a dense covariance", K);
    // entropy term: the inverse lands in g_cov and is rewritten in place
    APMF_TRY(apmf_dense_inv_logdet(ctx, d_cov, (int)K, d_g_cov, nullptr));
    hipLaunchKernelGGL(k_full_grad_init, dim3((unsigned)((K * K + 255) / 256)), dim3(256), 0, ctx->stream, K,
                       (size_t)r->n * d, sigma_u_sq, sigma_v_sq, d_mean, d_g_mean, d_g_cov);
    const double inv_sig = 1. / sigma_sq;
    hipLaunchKernelGGL((k_full_grad_rows<0>), dim3(r->n), dim3(64), 0, ctx->stream, r->n, r->m, d, r->ptr[0],
                       r->idx[0], r->val[0], d_mean, d_cov, inv_sig, d_g_mean, d_g_cov);
    hipLaunchKernelGGL((k_full_grad_rows<1>), dim3(r->m), dim3(64), 0, ctx->stream, r->n, r->m, d, r->ptr[1],
                       r->idx[1], r->val[1], d_mean, d_cov, inv_sig, d_g_mean, d_g_cov);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

// ActivePMF.kl_divergence (active_pmf.py:202-246): the K2-full sweep at the rated entries, the prior
// terms, and log|det cov| / 2
extern "C" int apmf_full_score_grid_f64(apmf_ctx *ctx, int n, int m, int d, int ld, const double *d_mean,
                                        const double *d_cov, const uint8_t *d_mask, double *d_var,
                                        double *d_pmean, double *d_sum, int want, double *d_best_val,
                                        int64_t *d_best_idx);
extern "C" int apmf_dense_project_psd(apmf_ctx *ctx, double *d_mat, int N, double min_eig, int *h_projected);

int apmf_full_kl_value(apmf_ctx *ctx, const apmf_ratings *r, int d, const double *d_mean, const double *d_cov,
                       double sigma_sq, double sigma_u_sq, double sigma_v_sq, double *h_kl)
{
    APMF_REQUIRE(ctx && r && d_mean && d_cov && h_kl, "NULL operand");
    APMF_REQUIRE(d >= 1 && d <= FG_D, "latent_d=%d unsupported by the full-covariance path (1..%d)", d, FG_D);
    const int n = r->n, m = r->m, ld = round_up(m, APMF_GRID_ALIGN);
    const size_t K = ((size_t)n + m) * d;
    double *grids = nullptr;
    APMF_TRY(apmf_malloc(ctx, (2 * (size_t)n * ld + 8) * sizeof(double), (void **)&grids));
    double *pvar = grids, *pmean = grids + (size_t)n * ld, *out3 = pmean + (size_t)n * ld;
    int rc = apmf_full_score_grid_f64(ctx, n, m, d, ld, d_mean, d_cov, nullptr, pvar, pmean, nullptr, 0, nullptr, nullptr);
    double t3[3] = {0., 0., 0.}, ld_c = 0.;
    if (rc == APMF_OK) {
        hipLaunchKernelGGL(k_full_kl_terms, dim3(1), dim3(256), 0, ctx->stream, n, m, d, ld, r->ptr[0], r->idx[0],
                           r->val[0], pvar, pmean, d_mean, d_cov, out3);
        rc = apmf_d2h(ctx, t3, out3, sizeof(t3));
    }
    (void)apmf_free(ctx, grids);
    if (rc == APMF_OK) rc = apmf_dense_inv_logdet(ctx, d_cov, (int)K, nullptr, &ld_c);
    if (rc != APMF_OK) return rc;
    *h_kl = t3[0] / (2. * sigma_sq) + t3[1] / (2. * sigma_u_sq) + t3[2] / (2. * sigma_v_sq) - ld_c / 2.;
    return APMF_OK;
}

// ActivePMF.fit_normal_kls / fit_normal (active_pmf.py:250-288) with the approximation resident on
// the device: gradient, trial point, projection, KL and the accept / reject decision of the
// learning-rate search without the K x K covariance crossing PCIe (K = (n+m) d).  Same calling
// convention as apmf_mn_fit_normal.
int apmf_full_fit_normal(apmf_ctx *ctx, const apmf_ratings *r, int d, double *d_mean, double *d_cov,
                         double sigma_sq, double sigma_u_sq, double sigma_v_sq, double min_eig, double *h_lr,
                         double *h_old_kl, int64_t max_steps, double *h_kl_trace, int64_t trace_cap,
                         int64_t *h_steps, int *h_converged)
{
    APMF_REQUIRE(ctx && r && d_mean && d_cov && h_lr && h_old_kl && h_steps && h_converged, "NULL operand");
    APMF_REQUIRE(trace_cap == 0 || h_kl_trace, "h_kl_trace is NULL");
    APMF_REQUIRE(d >= 1 && d <= FG_D, "latent_d=%d unsupported by the full-covariance path (1..%d)", d, FG_D);
    const size_t K = ((size_t)r->n + r->m) * d;
    const size_t cnt[2] = {K, K * K};
    double *cur[2] = {d_mean, d_cov}, *grad[2] = {nullptr, nullptr}, *cand[2] = {nullptr, nullptr};
    int rc = APMF_OK;
    for (int k = 0; k < 2 && rc == APMF_OK; k++) {
        rc = apmf_malloc(ctx, cnt[k] * sizeof(double), (void **)&grad[k]);
        if (rc == APMF_OK) rc = apmf_malloc(ctx, cnt[k] * sizeof(double), (void **)&cand[k]);
    }
    double lr = *h_lr, old_kl = *h_old_kl;
    int64_t steps = 0;
    bool converged = false;
    if (rc == APMF_OK && old_kl != old_kl)
        rc = apmf_full_kl_value(ctx, r, d, cur[0], cur[1], sigma_sq, sigma_u_sq, sigma_v_sq, &old_kl);
    while (rc == APMF_OK && !converged && (max_steps < 0 || steps < max_steps)) {
        rc = apmf_full_kl_gradient(ctx, r, d, cur[0], cur[1], sigma_sq, sigma_u_sq, sigma_v_sq, grad[0], grad[1]);
        while (rc == APMF_OK) {
            for (int k = 0; k < 2; k++)
                hipLaunchKernelGGL(k_trial_point, dim3((unsigned)((cnt[k] + 255) / 256)), dim3(256), 0, ctx->stream,
                                   cnt[k], cur[k], grad[k], lr, cand[k]);
            if (hipGetLastError() != hipSuccess) { apmf_set_error("kernel launch failed"); rc = APMF_EHIP; break; }
            if ((rc = apmf_dense_project_psd(ctx, cand[1], (int)K, min_eig, nullptr)) != APMF_OK) break;
            double new_kl = 0.;
            if ((rc = apmf_full_kl_value(ctx, r, d, cand[0], cand[1], sigma_sq, sigma_u_sq, sigma_v_sq, &new_kl)) != APMF_OK)
                break;
            if (new_kl < old_kl) {
                for (int k = 0; k < 2; k++)
                    (void)hipMemcpyAsync(cur[k], cand[k], cnt[k] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
                lr *= 1.25;
                if (old_kl - new_kl < .005) converged = true;
                if (steps < trace_cap) h_kl_trace[steps] = new_kl;
                steps++;
                old_kl = new_kl;
                break;
            }
            lr *= .5;
            if (lr < 1e-10) { converged = true; break; }
        }
    }
    if (rc == APMF_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {
        apmf_set_error("stream synchronisation failed");
        rc = APMF_EHIP;
    }
    for (int k = 0; k < 2; k++) {
        if (grad[k]) (void)apmf_free(ctx, grad[k]);
        if (cand[k]) (void)apmf_free(ctx, cand[k]);
    }
    *h_lr = lr;
    *h_old_kl = old_kl;
    *h_steps = steps;
    *h_converged = converged ? 1 : 0;
    return rc;
}

} // extern "C"
