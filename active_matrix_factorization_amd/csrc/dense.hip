// dense.hip - libapmf_dense.so: the (n+m) x (n+m) dense symmetric linear algebra of the
// variational fit, on the device:
//   project_psd  (mn_active_pmf.py:42-67: symmetrise, eigh, clip the spectrum, rebuild)
//   inverse + log|det| of a covariance (matrix_normal_exps_cy.pyx:180-182,477-485:
//   np.linalg.slogdet / np.linalg.inv, i.e. LAPACK getrf / getri)
// These are plain LAPACK-shaped operations, so they go to rocSOLVER / rocBLAS (syevd, getrf,
// getri, gemm); the element-wise glue (symmetrise, spectrum clip, mirror) is HIP.  Kept in a
// library of its own so that libapmf_hip.so has no BLAS dependency; loaded lazily.
#include <algorithm>

#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "common.h"
#include "chol.h"

namespace {

void destroy_handle(void *h) { (void)rocblas_destroy_handle((rocblas_handle)h); }

// one rocBLAS handle per context, created on first use, bound to the context's stream
int handle_for(apmf_ctx *ctx, rocblas_handle *out)
{
    if (!ctx->blas) {
        (void)hipSetDevice(ctx->device);
        rocblas_handle h = nullptr;
        if (rocblas_create_handle(&h) != rocblas_status_success) {
            apmf_set_error("rocblas_create_handle failed");
            return APMF_EHIP;
        }
        // bit-reproducible results: the accept / reject searches built on these routines
        // amplify a last-place difference into a different trajectory
        (void)rocblas_set_atomics_mode(h, rocblas_atomics_not_allowed);
        ctx->blas = h;
        ctx->blas_destroy = destroy_handle;
    }
    if (rocblas_set_stream((rocblas_handle)ctx->blas, ctx->stream) != rocblas_status_success) {
        apmf_set_error("rocblas_set_stream failed");
        return APMF_EHIP;
    }
    *out = (rocblas_handle)ctx->blas;
    return APMF_OK;
}

#define APMF_CHECK_ROC(expr)                                                                  \
    do {                                                                                      \
        rocblas_status _s = (expr);                                                           \
        if (_s != rocblas_status_success) {                                                   \
            apmf_set_error("%s failed: rocblas_status %d (%s:%d)", #expr, (int)_s, __FILE__,  \
                           __LINE__);                                                         \
            return APMF_EHIP;                                                                 \
        }                                                                                     \
    } while (0)

// A <- (A + A') / 2, done per (i, j > i) pair
__global__ __launch_bounds__(256) void k_symmetrize(double *__restrict__ A, int N)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)N * N) return;
    const int i = (int)(t / N), j = (int)(t % N);
    if (j <= i) return;
    const double v = (A[(size_t)i * N + j] + A[(size_t)j * N + i]) / 2;
    A[(size_t)i * N + j] = v;
    A[(size_t)j * N + i] = v;
}

// Vs[:, c] = V[:, c] * max(w[c], min_eig)   (column-major eigenvectors)
__global__ __launch_bounds__(256) void k_scale_cols(const double *__restrict__ V, const double *__restrict__ w,
                                                    double min_eig, double *__restrict__ Vs, int N)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)N * N) return;
    const int c = (int)(t / N);
    const double e = w[c];
    Vs[t] = V[t] * (e > min_eig ? e : min_eig);
}

// log |det| from the diagonal of an LU factor (numpy.linalg.slogdet's second value)
__global__ __launch_bounds__(256) void k_logabsdet_lu(const double *__restrict__ L, int N, double *__restrict__ out)
{
    __shared__ double sh[256];
    double s = 0;
    for (int i = threadIdx.x; i < N; i += 256) s += log(fabs(L[(size_t)i * N + i]));
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

int grid_for(size_t count) { return (int)((count + 255) / 256); }

// A[i][i] -= min_eig + 1e-10 * max(|A[i][i]|, |min_eig|)  (scratch[0..N) is free to use)
__global__ __launch_bounds__(256) void k_shift_diag(double *A, int N, double min_eig, double *scratch)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double a = A[(size_t)i * N + i];
    A[(size_t)i * N + i] = a - (min_eig + 1e-10 * fmax(fabs(a), fabs(min_eig)));
}

// ---- symmetric positive definite fast paths -------------------------------------------
// The covariances these routines see during a fit are symmetric (bitwise: project_psd
// symmetrises) and positive definite, so log|det| and the inverse come from a Cholesky
// factorisation: rocSOLVER's LU (getrf) is ~4x slower at these sizes and was 2/3 of a fit.
// Anything that is not exactly symmetric, or whose factorisation fails, takes the LU path.

// flag |= 1 if A is not bitwise symmetric
__global__ __launch_bounds__(256) void k_check_symmetric(const double *__restrict__ A, int N, int *flag)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)N * N) return;
    const int i = (int)(t / N), j = (int)(t % N);
    if (j > i && A[(size_t)i * N + j] != A[(size_t)j * N + i]) atomicOr(flag, 1);
}

// out = 2 sum log L[i][i] of a Cholesky factor
__global__ __launch_bounds__(256) void k_logdet_chol(const double *__restrict__ L, int N, double *__restrict__ out)
{
    __shared__ double sh[256];
    double s = 0;
    for (int i = threadIdx.x; i < N; i += 256) s += log(L[(size_t)i * N + i]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = 2. * sh[0];
}

// the row-major lower triangle (i > j at i * N + j) mirrored into the upper one
__global__ __launch_bounds__(256) void k_mirror_triangle(double *__restrict__ A, int N)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)N * N) return;
    const int i = (int)(t / N), j = (int)(t % N);
    if (i > j) A[(size_t)j * N + i] = A[t];
}

// N <= 32, one wave: Cholesky in LDS of A (shifted down on the diagonal when `shift` is set:
// the project_psd certificate), then optionally log|det| and the inverse.
// res[0] = 0 ok / 1 not symmetric or not positive definite, res[1] = log det.
constexpr int SMALL_N = 32;
__global__ __launch_bounds__(64) void k_small_spd(const double *__restrict__ A, int N, int shift, double min_eig,
                                                  double *__restrict__ inv_out, double *__restrict__ res)
{
    __shared__ double L[SMALL_N][SMALL_N + 1], X[SMALL_N][SMALL_N + 1];
    __shared__ int bad;
    const int t = threadIdx.x;
    if (t == 0) bad = 0;
    __syncthreads();
    for (int e = t; e < N * N; e += 64) {
        const int i = e / N, j = e % N;
        double v = A[e];
        if (A[(size_t)j * N + i] != v) bad = 1;
        if (shift && i == j) v -= min_eig + 1e-10 * fmax(fabs(v), fabs(min_eig));
        L[i][j] = v;
    }
    __syncthreads();
    // right-looking Cholesky, lane i owns row i
    for (int j = 0; j < N && !bad; j++) {
        __syncthreads();
        const double djj = L[j][j];
        if (!(djj > 0.)) {
            if (t == 0) bad = 1;
            __syncthreads();
            break;
        }
        const double rj = sqrt(djj);
        __syncthreads();
        if (t == j) L[j][j] = rj;
        if (t > j && t < N) L[t][j] /= rj;
        __syncthreads();
        if (t > j && t < N) {
            const double lij = L[t][j];
            for (int k = j + 1; k <= t; k++) L[t][k] -= lij * L[k][j];
        }
    }
    __syncthreads();
    if (bad) {
        if (t == 0) { res[0] = 1.; res[1] = 0.; }
        return;
    }
    if (t == 0) {
        double s = 0;
        for (int i = 0; i < N; i++) s += log(L[i][i]);
        res[0] = 0.;
        res[1] = 2. * s;
    }
    if (!inv_out) return;
    // X = L^-1 (lower), column t by forward substitution
    if (t < N) {
        for (int i = 0; i < N; i++) {
            double v = (i == t) ? 1. : 0.;
            for (int k = t; k < i; k++) v -= L[i][k] * X[k][t];
            X[i][t] = i < t ? 0. : v / L[i][i];
        }
    }
    __syncthreads();
    // A^-1 = X' X
    for (int e = t; e < N * N; e += 64) {
        const int i = e / N, j = e % N;
        double v = 0;
        for (int k = (i > j ? i : j); k < N; k++) v += X[k][i] * X[k][j];
        inv_out[e] = v;
    }
}

// ---- blocked Cholesky of our own (chol.h) ----------------------------------------------------
// Cholesky of the lower triangle of the row-major N x N matrix A0 (and, side by side in the same
// launches, of A1 when it is not NULL), in place; d_info (one device int per matrix, zeroed here)
// receives 0 or the 1-based index of the first non-positive pivot; d_linv: CH_NB^2 doubles per matrix.
int chol_lower(apmf_ctx *ctx, double *A0, int N, int *d_info, double *d_linv, double *A1 = nullptr)
{
    const int nmat = A1 ? 2 : 1;
    APMF_CHECK_HIP(hipMemsetAsync(d_info, 0, nmat * sizeof(int), ctx->stream));
    ChJobs J;
    J.j[0] = ChJob{A0, d_info, d_linv, 0};
    J.j[1] = ChJob{A1, d_info + 1, d_linv + CH_NB * CH_NB, 0};
    J.list = nullptr;
    return chol_lower_jobs(ctx, J, nmat, N);
}

} // namespace

extern "C" {

// the context's rocBLAS handle (bound to its stream), for k6_batch.hip's batched eigendecomposition
int apmf_dense_handle(apmf_ctx *ctx, void **out)
{
    rocblas_handle h;
    APMF_TRY(handle_for(ctx, &h));
    *out = h;
    return APMF_OK;
}

// handle of slow lane `lane` (0 .. 3) of this context, bound to `stream`
int apmf_dense_lane_handle(apmf_ctx *ctx, int lane, hipStream_t stream, void **out)
{
    APMF_REQUIRE(ctx && out && lane >= 0 && lane < 4, "bad lane");
    rocblas_handle h0;
    APMF_TRY(handle_for(ctx, &h0));          // makes sure blas_destroy is set (and the library initialised)
    if (!ctx->lane_blas[lane]) {
        rocblas_handle h = nullptr;
        if (rocblas_create_handle(&h) != rocblas_status_success) {
            apmf_set_error("rocblas_create_handle failed for a slow lane");
            return APMF_EHIP;
        }
        ctx->lane_blas[lane] = h;
    }
    if (rocblas_set_stream((rocblas_handle)ctx->lane_blas[lane], stream) != rocblas_status_success) {
        apmf_set_error("rocblas_set_stream failed");
        return APMF_EHIP;
    }
    *out = ctx->lane_blas[lane];
    return APMF_OK;
}

int apmf_dense_project_psd(apmf_ctx *ctx, double *d_mat, int N, double min_eig, int *h_projected)
{
    APMF_REQUIRE(ctx && d_mat, "NULL operand");
    APMF_REQUIRE(N >= 1, "bad size");
    rocblas_handle h;
    APMF_TRY(handle_for(ctx, &h));
    const size_t nn = (size_t)N * N;
    void *w;
    APMF_TRY(apmf_ws(ctx, 3, (2 * nn + 2 * (size_t)N + 8 + CH_NB * CH_NB) * sizeof(double) + 64, &w));
    double *V = (double *)w, *Vs = V + nn, *evals = Vs + nn, *E = evals + N;
    int *info = (int *)(E + N);
    double *linv = E + N + 8;      // scratch of the blocked Cholesky
    hipLaunchKernelGGL(k_symmetrize, dim3(grid_for(nn)), dim3(256), 0, ctx->stream, d_mat, N);
    APMF_CHECK_LAUNCH();
    double wmin;
    int hinfo;
    if (N <= SMALL_N) {
        // d x d latent covariance: the same certificate by one single-wave kernel
        hipLaunchKernelGGL(k_small_spd, dim3(1), dim3(64), 0, ctx->stream, d_mat, N, 1, min_eig,
                           (double *)nullptr, evals);
        APMF_CHECK_LAUNCH();
        double res[2];
        APMF_TRY(apmf_d2h(ctx, res, evals, sizeof(res)));
        if (res[0] == 0.) {
            if (h_projected) *h_projected = 0;
            return APMF_OK;
        }
    }
    // Fast path.  The reference only rebuilds the matrix when its smallest eigenvalue is below
    // min_eig; during a fit that is the exception.  "All eigenvalues > min_eig + delta" is
    // certified by a Cholesky factorisation of the shifted matrix (a few % of the cost of the
    // eigendecomposition); delta keeps the certificate clear of the rounding of either test.
    // On failure nothing is concluded and the eigendecomposition decides, as before.
    APMF_CHECK_HIP(hipMemcpyAsync(V, d_mat, nn * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_shift_diag, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, V, N, min_eig, evals);
    APMF_CHECK_LAUNCH();
    APMF_TRY(chol_lower(ctx, V, N, info, linv));
    APMF_TRY(apmf_d2h(ctx, &hinfo, info, sizeof(int)));
    if (hinfo == 0) {
        if (h_projected) *h_projected = 0;
        return APMF_OK;             // the symmetrised input is returned unchanged
    }
    APMF_CHECK_HIP(hipMemcpyAsync(V, d_mat, nn * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    APMF_CHECK_ROC(rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, N, V, N, evals, E, info));
    APMF_TRY(apmf_d2h(ctx, &wmin, evals, sizeof(double)));   // ascending: evals[0] is the minimum
    APMF_TRY(apmf_d2h(ctx, &hinfo, info, sizeof(int)));
    if (hinfo != 0) {
        apmf_set_error("rocsolver_dsyevd did not converge (info=%d)", hinfo);
        return APMF_EHIP;
    }
    const bool project = wmin < min_eig;
    if (h_projected) *h_projected = project ? 1 : 0;
    if (!project) return APMF_OK;   // the symmetrised input is returned unchanged
    hipLaunchKernelGGL(k_scale_cols, dim3(grid_for(nn)), dim3(256), 0, ctx->stream, V, evals, min_eig, Vs, N);
    APMF_CHECK_LAUNCH();
    const double one = 1., zero = 0.;
    APMF_CHECK_ROC(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, N, N, N, &one, Vs, N,
                                 V, N, &zero, d_mat, N));
    hipLaunchKernelGGL(k_symmetrize, dim3(grid_for(nn)), dim3(256), 0, ctx->stream, d_mat, N);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

int apmf_dense_inv_logdet(apmf_ctx *ctx, const double *d_mat, int N, double *d_inv, double *h_logdet)
{
    APMF_REQUIRE(ctx && d_mat, "NULL operand");
    APMF_REQUIRE(N >= 1, "bad size");
    if (!d_inv && !h_logdet) return APMF_OK;
    rocblas_handle h;
    APMF_TRY(handle_for(ctx, &h));
    const size_t nn = (size_t)N * N;
    void *w;
    const size_t linv_off = ((nn + 8) * sizeof(double) + ((size_t)N + 16) * sizeof(int) + 15) / 16 * 16;
    APMF_TRY(apmf_ws(ctx, 3, linv_off + CH_NB * CH_NB * sizeof(double) + 64, &w));
    double *linv = (double *)((char *)w + linv_off);   // scratch of the blocked Cholesky
    double *LU = d_inv ? d_inv : (double *)w;
    double *scal = (double *)w + nn;
    int *info = (int *)(scal + 4);
    int *ipiv = info + 8;
    // symmetric positive definite input: Cholesky instead of LU (see the kernels above)
    if (LU == d_mat) {
        // in place: the input must survive a failed Cholesky attempt, so straight to LU
    } else if (N <= SMALL_N) {
        hipLaunchKernelGGL(k_small_spd, dim3(1), dim3(64), 0, ctx->stream, d_mat, N, 0, 0., d_inv, scal);
        APMF_CHECK_LAUNCH();
        double res[2];
        APMF_TRY(apmf_d2h(ctx, res, scal, sizeof(res)));
        if (res[0] == 0.) {
            if (h_logdet) *h_logdet = res[1];
            return APMF_OK;
        }
    } else {
        APMF_CHECK_HIP(hipMemsetAsync(info + 1, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_check_symmetric, dim3(grid_for(nn)), dim3(256), 0, ctx->stream, d_mat, N, info + 1);
        APMF_CHECK_LAUNCH();
        APMF_CHECK_HIP(hipMemcpyAsync(LU, d_mat, nn * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        APMF_TRY(chol_lower(ctx, LU, N, info, linv));
        int hi[2];
        APMF_TRY(apmf_d2h(ctx, hi, info, sizeof(hi)));
        if (hi[0] == 0 && hi[1] == 0) {
            if (h_logdet) {
                hipLaunchKernelGGL(k_logdet_chol, dim3(1), dim3(256), 0, ctx->stream, LU, N, scal);
                APMF_CHECK_LAUNCH();
                APMF_TRY(apmf_d2h(ctx, h_logdet, scal, sizeof(double)));
            }
            if (d_inv) {
                // row-major lower L = column-major upper U with A = U'U
                APMF_CHECK_ROC(rocsolver_dpotri(h, rocblas_fill_upper, N, LU, N, info));
                hipLaunchKernelGGL(k_mirror_triangle, dim3(grid_for(nn)), dim3(256), 0, ctx->stream, LU, N);
                APMF_CHECK_LAUNCH();
            }
            return APMF_OK;
        }
    }
    if (LU != d_mat)
        APMF_CHECK_HIP(hipMemcpyAsync(LU, d_mat, nn * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    // LU of the transpose (row-major data seen column-major): same determinant, and the inverse
    // of the transpose is the transpose of the inverse - exactly what row-major storage wants
    APMF_CHECK_ROC(rocsolver_dgetrf(h, N, N, LU, N, ipiv, info));
    int hinfo;
    APMF_TRY(apmf_d2h(ctx, &hinfo, info, sizeof(int)));
    if (hinfo != 0) {
        apmf_set_error("Singular matrix (getrf info=%d)", hinfo);
        return APMF_ENOTPD;
    }
    if (h_logdet) {
        hipLaunchKernelGGL(k_logabsdet_lu, dim3(1), dim3(256), 0, ctx->stream, LU, N, scal);
        APMF_CHECK_LAUNCH();
        APMF_TRY(apmf_d2h(ctx, h_logdet, scal, sizeof(double)));
    }
    if (d_inv) APMF_CHECK_ROC(rocsolver_dgetri(h, N, LU, N, ipiv, info));
    return APMF_OK;
}

// log|det| of a (hopefully) symmetric positive definite matrix, keeping its Cholesky factor:
// on success *h_have_factor = 1 and d_L (N x N) holds L in its row-major lower triangle, ready for
// apmf_dense_inverse_from_factor - the variational fit evaluates the KL of a trial point and, if
// the trial is accepted, the gradient's inverse of the SAME matrix.  Otherwise (small, not
// symmetric, not positive definite) *h_have_factor = 0 and the general routine supplies the value.
int apmf_dense_logdet_keep_factor(apmf_ctx *ctx, const double *d_mat, int N, double *d_L, double *h_logdet,
                                  int *h_have_factor)
{
    APMF_REQUIRE(ctx && d_mat && d_L && h_logdet && h_have_factor, "NULL operand");
    *h_have_factor = 0;
    if (N > SMALL_N) {
        const size_t nn = (size_t)N * N;
        void *w;
        APMF_TRY(apmf_ws(ctx, 3, (CH_NB * CH_NB + 16) * sizeof(double) + 64, &w));
        double *linv = (double *)w, *scal = linv + CH_NB * CH_NB;
        int *info = (int *)(scal + 4);
        APMF_CHECK_HIP(hipMemsetAsync(info + 1, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_check_symmetric, dim3(grid_for(nn)), dim3(256), 0, ctx->stream, d_mat, N, info + 1);
        APMF_CHECK_LAUNCH();
        APMF_CHECK_HIP(hipMemcpyAsync(d_L, d_mat, nn * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        APMF_TRY(chol_lower(ctx, d_L, N, info, linv));
        hipLaunchKernelGGL(k_logdet_chol, dim3(1), dim3(256), 0, ctx->stream, d_L, N, scal);
        APMF_CHECK_LAUNCH();
        double back[6];      // logdet (only meaningful if the flags are clear) and the two flags
        APMF_TRY(apmf_d2h(ctx, back, scal, sizeof(back)));
        int hi[2];
        memcpy(hi, &back[4], sizeof(hi));
        if (hi[0] == 0 && hi[1] == 0) {
            *h_logdet = back[0];
            *h_have_factor = 1;
            return APMF_OK;
        }
    }
    return apmf_dense_inv_logdet(ctx, d_mat, N, nullptr, h_logdet);
}

// d_inv <- (L L')^-1 from the factor left by apmf_dense_logdet_keep_factor (d_L is not modified)
int apmf_dense_inverse_from_factor(apmf_ctx *ctx, const double *d_L, int N, double *d_inv)
{
    APMF_REQUIRE(ctx && d_L && d_inv && d_L != d_inv, "bad operand");
    rocblas_handle h;
    APMF_TRY(handle_for(ctx, &h));
    const size_t nn = (size_t)N * N;
    void *w;
    APMF_TRY(apmf_ws(ctx, 3, 64, &w));
    APMF_CHECK_HIP(hipMemcpyAsync(d_inv, d_L, nn * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    // row-major lower L = column-major upper U with A = U'U
    APMF_CHECK_ROC(rocsolver_dpotri(h, rocblas_fill_upper, N, d_inv, N, (int *)w));
    hipLaunchKernelGGL(k_mirror_triangle, dim3(grid_for(nn)), dim3(256), 0, ctx->stream, d_inv, N);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

// One trial of the variational fit needs, for the candidate covariance S: (a) project_psd's
// verdict "no eigenvalue below min_eig" and (b) log det S with the factor kept.  Both are
// Cholesky factorisations - of S - (min_eig + delta) I and of S - and they are run side by side in the
// same launches (the factorisation is a latency-bound chain of small kernels; two cost what one
// does).  d_mat is symmetrised in place, like project_psd does first.  *h_certified = 1: nothing
// to clip, and then *h_have_factor / *h_logdet / d_L are as in apmf_dense_logdet_keep_factor.
// *h_certified = 0: nothing is concluded (the caller runs the general project_psd).
int apmf_dense_certify_and_factor(apmf_ctx *ctx, double *d_mat, int N, double min_eig, double *d_L,
                                  double *h_logdet, int *h_certified, int *h_have_factor)
{
    APMF_REQUIRE(ctx && d_mat && d_L && h_logdet && h_certified && h_have_factor, "NULL operand");
    APMF_REQUIRE(N > SMALL_N, "small matrices take the single-kernel path");
    *h_certified = 0;
    *h_have_factor = 0;
    const size_t nn = (size_t)N * N;
    void *w;
    APMF_TRY(apmf_ws(ctx, 3, (nn + 2 * CH_NB * CH_NB + 16) * sizeof(double) + 64, &w));
    double *V = (double *)w, *linv = V + nn, *scal = linv + 2 * CH_NB * CH_NB;
    int *info = (int *)(scal + 4);
    hipLaunchKernelGGL(k_symmetrize, dim3(grid_for(nn)), dim3(256), 0, ctx->stream, d_mat, N);
    APMF_CHECK_LAUNCH();
    APMF_CHECK_HIP(hipMemcpyAsync(V, d_mat, nn * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    APMF_CHECK_HIP(hipMemcpyAsync(d_L, d_mat, nn * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_shift_diag, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, V, N, min_eig, scal);
    APMF_CHECK_LAUNCH();
    APMF_TRY(chol_lower(ctx, V, N, info, linv, d_L));
    hipLaunchKernelGGL(k_logdet_chol, dim3(1), dim3(256), 0, ctx->stream, d_L, N, scal);
    APMF_CHECK_LAUNCH();
    double back[6];
    APMF_TRY(apmf_d2h(ctx, back, scal, sizeof(back)));
    int hi[2];
    memcpy(hi, &back[4], sizeof(hi));
    if (hi[0] != 0) return APMF_OK;          // not certified
    *h_certified = 1;
    if (hi[1] == 0) {
        *h_logdet = back[0];
        *h_have_factor = 1;
        return APMF_OK;
    }
    return apmf_dense_inv_logdet(ctx, d_mat, N, nullptr, h_logdet);
}

} // extern "C"
