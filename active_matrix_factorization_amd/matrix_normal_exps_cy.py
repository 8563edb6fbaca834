"""Drop-in for the reference's ``matrix_normal_exps_cy`` module
(python-pmf/matrix_normal_exps_cy.pyx): moments of a matrix-normal MN(mean, cov_rows,
cov_cols) and the KL objective / gradient of the variational fit.

* The scalar moment functions (tripexpect, quadexpect, exp_squared, exp_a2bc,
  exp_dotprod_sq) are one-candidate conveniences: plain arithmetic on three or four
  entries.  The hot loop that called exp_dotprod_sq once per candidate in the reference is
  replaced by the batched HIP kernels behind MNActivePMF (K2, csrc/k2_score.hip).
* mn_kl_divergence / matrixnormal_gradient are the variational-fit pair (SURVEY.md 8(f)
  rank 1).  Both run on the device: the per-rating sums as the K5 kernels
  (csrc/k5_klgrad.hip), the (n+m) x (n+m) inverse / log-determinant through rocSOLVER
  (libapmf_dense.so).  The reference's quirks are reproduced on purpose (see each comment).
"""
import numpy as np

DTYPE = float


def tripexpect(mean, cov_rows, cov_cols, a_i, a_j, b_i, b_j, c_i, c_j):
    '''E[a b c] for MN(mean, cov_rows, cov_cols)   (matrix_normal_exps_cy.pyx:9-23)'''
    ma, mb, mc = mean[a_i, a_j], mean[b_i, b_j], mean[c_i, c_j]
    return (ma * mb * mc
            + ma * cov_rows[b_i, c_i] * cov_cols[b_j, c_j]
            + mb * cov_rows[a_i, c_i] * cov_cols[a_j, c_j]
            + mc * cov_rows[a_i, b_i] * cov_cols[a_j, b_j])


def _cov(cov_rows, cov_cols, x, y):
    return cov_rows[x[0], y[0]] * cov_cols[x[1], y[1]]


def quadexpect(mean, cov_rows, cov_cols, a_i, a_j, b_i, b_j, c_i, c_j, d_i, d_j):
    '''E[a b c d] for distinct entries (Isserlis)   (matrix_normal_exps_cy.pyx:28-69)'''
    a, b, c, d = (a_i, a_j), (b_i, b_j), (c_i, c_j), (d_i, d_j)
    ma, mb, mc, md = (float(mean[x]) for x in (a, b, c, d))
    ab, ac, ad = (_cov(cov_rows, cov_cols, a, x) for x in (b, c, d))
    bc, bd = (_cov(cov_rows, cov_cols, b, x) for x in (c, d))
    cd = _cov(cov_rows, cov_cols, c, d)
    return (ma * mb * mc * md
            + ma * mb * cd + ma * mc * bd + ma * md * bc
            + mb * mc * ad + mb * md * ac + mc * md * ab
            + ab * cd + ac * bd + ad * bc)


def exp_squared(mean, cov_rows, cov_cols, a_i, a_j, b_i, b_j):
    '''E[a^2 b^2]   (matrix_normal_exps_cy.pyx:73-91)'''
    a, b = (a_i, a_j), (b_i, b_j)
    ma, mb = float(mean[a]), float(mean[b])
    ab = _cov(cov_rows, cov_cols, a, b)
    va, vb = _cov(cov_rows, cov_cols, a, a), _cov(cov_rows, cov_cols, b, b)
    return 4 * ma * mb * ab + 2 * ab ** 2 + (ma ** 2 + va) * (mb ** 2 + vb)


def exp_a2bc(mean, cov_rows, cov_cols, a_i, a_j, b_i, b_j, c_i, c_j):
    '''E[a^2 b c]   (matrix_normal_exps_cy.pyx:95-121)'''
    a, b, c = (a_i, a_j), (b_i, b_j), (c_i, c_j)
    ma, mb, mc = float(mean[a]), float(mean[b]), float(mean[c])
    va = _cov(cov_rows, cov_cols, a, a)
    ab, ac, bc = (_cov(cov_rows, cov_cols, *p) for p in ((a, b), (a, c), (b, c)))
    return (ma ** 2 + va) * (mb * mc + bc) + 2 * ma * mc * ab + 2 * ma * mb * ac + 2 * ab * ac


def exp_dotprod_sq(num_users, mean, cov_useritems, cov_latents, i, j):
    '''E[(U_i . V_j)^2] for ONE candidate   (matrix_normal_exps_cy.pyx:126-154).

    Same d(d+1)/2-term Isserlis sum and term order as the reference.  Do not loop over this
    for a pool of candidates: MNActivePMF scores pools and whole grids on the GPU.'''
    d = mean.shape[1]
    j_ = num_users + j
    e = 0.
    for k in range(d):
        e += exp_squared(mean, cov_useritems, cov_latents, i, k, j_, k)
        for l in range(k + 1, d):
            e += 2 * quadexpect(mean, cov_useritems, cov_latents, i, k, j_, k, i, l, j_, l)
    return e


def _ctx():
    try:
        from . import _lib
    except ImportError:
        import _lib
    return _lib.default_context()


HOST_DENSE_MAX = 16   # d x d matrices up to this size stay host LAPACK work (like the
                      # hyper-parameter draws); the (n+m) x (n+m) ones go to the device


def logabsdet(mat):
    """np.linalg.slogdet(mat)[1]; on the device (LU via rocSOLVER, libapmf_dense.so) beyond
    HOST_DENSE_MAX."""
    mat = np.ascontiguousarray(mat, np.float64)
    if mat.shape[0] <= HOST_DENSE_MAX:
        return float(np.linalg.slogdet(mat)[1])
    ctx = _ctx()
    return ctx.dense_inv_logdet(ctx.array(mat), mat.shape[0], None, want_logdet=True)


def inverse(mat):
    """np.linalg.inv(mat); on the device (getrf + getri via rocSOLVER) beyond HOST_DENSE_MAX."""
    mat = np.ascontiguousarray(mat, np.float64)
    if mat.shape[0] <= HOST_DENSE_MAX:
        return np.linalg.inv(mat)
    ctx = _ctx()
    out = ctx.empty(mat.shape)
    ctx.dense_inv_logdet(ctx.array(mat), mat.shape[0], out, want_logdet=False)
    return out.to_host()


def _ratings_on_device(ctx, ratings, num_users, num_useritems):
    return ctx.ratings(np.ascontiguousarray(ratings, np.float64), int(num_users),
                       int(num_useritems) - int(num_users))


def mn_kl_divergence(num_users, ratings, mean, cov_useritems, cov_latents,
                     sigma_sq, sigma_u_sq, sigma_v_sq, _device_ratings=None):
    '''KL( MN(mean, cov_useritems, cov_latents) || PMF posterior ) up to a constant
    (matrix_normal_exps_cy.pyx:159-213), evaluated on the device (K5 + LU log-determinants),
    quirks included:
      * `num_items = num_useritems - num_useritems` = 0, so the item-covariance trace term
        is dropped (:176, :191-192);
      * the item prior uses sigma_u_sq, not sigma_v_sq (:196-197).'''
    ctx = _ctx()
    mean = np.ascontiguousarray(mean, np.float64)
    rt = _device_ratings or _ratings_on_device(ctx, ratings, num_users, mean.shape[0])
    kl = ctx.mn_kl_value(rt, mean.shape[1], ctx.array(mean), ctx.array(cov_useritems, np.float64),
                         ctx.array(cov_latents, np.float64), float(sigma_sq), float(sigma_u_sq),
                         float(sigma_v_sq))
    if _device_ratings is None:
        rt.free()
    return kl


def matrixnormal_gradient(mn_apmf):
    '''Gradient of the KL w.r.t. (mean, cov_useritems, cov_latents) of an MNActivePMF
    (matrix_normal_exps_cy.pyx:216-244).  Raises TypeError if the approximation is unset.'''
    if mn_apmf is None or mn_apmf.mean is None or mn_apmf.cov_useritems is None \
            or mn_apmf.cov_latents is None:
        raise TypeError("mean, cov are None; run initialize_approx first")
    rt = mn_apmf._device_ratings()[1] if hasattr(mn_apmf, '_device_ratings') else None
    return _mnormal_grad(mn_apmf.mean, mn_apmf.cov_useritems, mn_apmf.cov_latents,
                         mn_apmf.ratings, mn_apmf.num_users, mn_apmf.latent_d,
                         mn_apmf.sigma_sq, mn_apmf.sigma_u_sq, mn_apmf.sigma_v_sq,
                         _device_ratings=rt)


def _mnormal_grad(mean, cov_useritems, cov_latents, ratings, num_users, latent_d,
                  sig_sq, sig_u_sq, sig_v_sq, _device_ratings=None):
    '''matrix_normal_exps_cy.pyx:395-485 (+ _quadexp_grad :250-340, _squareexp_grad :344-389)
    as the K5 kernels (csrc/k5_klgrad.hip): one wave per user / item row accumulates the
    collapsed pair sums, the (n+m) x (n+m) and d x d inverses of the entropy term come from
    rocSOLVER.  Returns (g_mean, g_cov_useritems, g_cov_latents) as NumPy arrays; like the
    reference, the derivative w.r.t. one triangular half is written into both halves.'''
    ctx = _ctx()
    mean = np.ascontiguousarray(mean, np.float64)
    N, d = mean.shape
    rt = _device_ratings or _ratings_on_device(ctx, ratings, num_users, N)
    g_mean, g_cu, g_cl = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
    ctx.mn_kl_gradient(rt, d, ctx.array(mean), ctx.array(cov_useritems, np.float64),
                       ctx.array(cov_latents, np.float64), float(sig_sq), float(sig_u_sq),
                       float(sig_v_sq), g_mean, g_cu, g_cl)
    out = g_mean.to_host(), g_cu.to_host(), g_cl.to_host()
    if _device_ratings is None:
        rt.free()
    return out
