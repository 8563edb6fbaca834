// dense.hip - libapmf_dense.so: the (n+m) x (n+m) dense symmetric linear algebra of the
// variational fit, on the device:
//   project_psd  (mn_active_pmf.py:42-67: symmetrise, eigh, clip the spectrum, rebuild)
//   inverse + log|det| of a covariance (matrix_normal_exps_cy.pyx:180-182,477-485:
//   np.linalg.slogdet / np.linalg.inv, i.e. LAPACK getrf / getri)
// These are plain LAPACK-shaped operations, so they go to rocSOLVER / rocBLAS (syevd, getrf,
// getri, gemm); the element-wise glue (symmetrise, spectrum clip, mirror) is HIP.  Kept in a
// library of its own so that libapmf_hip.so has no BLAS dependency; loaded lazily.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "common.h"

namespace {

rocblas_handle g_handle = nullptr;

int handle_for(apmf_ctx *ctx, rocblas_handle *out)
{
    if (!g_handle) {
        if (rocblas_create_handle(&g_handle) != rocblas_status_success) {
            apmf_set_error("rocblas_create_handle failed");
            return APMF_EHIP;
        }
    }
    if (rocblas_set_stream(g_handle, ctx->stream) != rocblas_status_success) {
        apmf_set_error("rocblas_set_stream failed");
        return APMF_EHIP;
    }
    *out = g_handle;
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

// the triangle potri wrote (memory index c * N + r, r >= c) mirrored into the other one
__global__ __launch_bounds__(256) void k_mirror_triangle(double *__restrict__ A, int N)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)N * N) return;
    const int c = (int)(t / N), r = (int)(t % N);
    if (r > c) A[(size_t)r * N + c] = A[t];
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

} // namespace

extern "C" {

int apmf_dense_project_psd(apmf_ctx *ctx, double *d_mat, int N, double min_eig, int *h_projected)
{
    APMF_REQUIRE(ctx && d_mat, "NULL operand");
    APMF_REQUIRE(N >= 1, "bad size");
    rocblas_handle h;
    APMF_TRY(handle_for(ctx, &h));
    const size_t nn = (size_t)N * N;
    void *w;
    APMF_TRY(apmf_ws(ctx, 3, (2 * nn + 2 * (size_t)N) * sizeof(double) + 64, &w));
    double *V = (double *)w, *Vs = V + nn, *evals = Vs + nn, *E = evals + N;
    int *info = (int *)(E + N);
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
    APMF_CHECK_ROC(rocsolver_dpotrf(h, rocblas_fill_lower, N, V, N, info));
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
    APMF_TRY(apmf_ws(ctx, 3, (nn + 8) * sizeof(double) + ((size_t)N + 16) * sizeof(int) + 64, &w));
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
        APMF_CHECK_ROC(rocsolver_dpotrf(h, rocblas_fill_lower, N, LU, N, info));
        int hi[2];
        APMF_TRY(apmf_d2h(ctx, hi, info, sizeof(hi)));
        if (hi[0] == 0 && hi[1] == 0) {
            if (h_logdet) {
                hipLaunchKernelGGL(k_logdet_chol, dim3(1), dim3(256), 0, ctx->stream, LU, N, scal);
                APMF_CHECK_LAUNCH();
                APMF_TRY(apmf_d2h(ctx, h_logdet, scal, sizeof(double)));
            }
            if (d_inv) {
                APMF_CHECK_ROC(rocsolver_dpotri(h, rocblas_fill_lower, N, LU, N, info));
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

} // extern "C"
