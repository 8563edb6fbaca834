"""Drop-in for the reference's ``matrix_normal_exps_cy`` module
(python-pmf/matrix_normal_exps_cy.pyx): moments of a matrix-normal MN(mean, cov_rows,
cov_cols) and the KL objective / gradient of the variational fit.

* The scalar moment functions (tripexpect, quadexpect, exp_squared, exp_a2bc,
  exp_dotprod_sq) are one-candidate conveniences: plain arithmetic on three or four
  entries.  The hot loop that called exp_dotprod_sq once per candidate in the reference is
  replaced by the batched HIP kernels behind MNActivePMF (K2, csrc/k2_score.hip).
* mn_kl_divergence / matrixnormal_gradient are the variational-fit pair (SURVEY.md 8(f)
  rank 1, "next").  This round the per-rating sums of the gradient are vectorised over the
  ratings on the host; the per-rating E[(U_i.V_j)^2] terms of the KL come from the K2 pool
  kernel and the (n+m) x (n+m) inverse / log-determinant run on the device (rocSOLVER behind
  libapmf_dense.so).  The reference's quirks are reproduced on purpose (see each comment).
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


def _rating_operands(num_users, ratings, mean, cov_useritems):
    ii = ratings[:, 0].astype(np.intp)
    jj = ratings[:, 1].astype(np.intp) + num_users
    return (ii, jj, ratings[:, 2], mean[ii], mean[jj], cov_useritems[ii, jj],
            cov_useritems[ii, ii], cov_useritems[jj, jj])


def _upper_mirrored(cov_latents):
    # the reference only reads cov_latents[k, l] with k <= l
    up = np.triu(cov_latents)
    return up + np.triu(cov_latents, 1).T


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


def _e_dot_sq_on_device(num_users, ratings, mean, cov_useritems, cov_latents):
    """E[(U_i.V_j)^2] of every rated entry = Var + E^2 from the K2 pool kernel (no host path)."""
    ctx = _ctx()
    n = int(num_users)
    m, d = mean.shape[0] - n, mean.shape[1]
    cov_useritems = np.asarray(cov_useritems, np.float64)
    cross = ctx.upload_grid(cov_useritems[:n, n:], np.float64)
    P = ratings.shape[0]
    pm, pv = ctx.empty(P), ctx.empty(P)
    ctx.mn_score_pool(n, m, d, cross.shape[1], ctx.array(mean, np.float64), cross,
                      ctx.array(np.ascontiguousarray(np.diagonal(cov_useritems)), np.float64),
                      ctx.array(cov_latents, np.float64),
                      ctx.array(np.ascontiguousarray(ratings[:, 0], np.int32), np.int32),
                      ctx.array(np.ascontiguousarray(ratings[:, 1], np.int32), np.int32), P, pm, pv)
    pm, pv = pm.to_host(), pv.to_host()
    return pv + pm * pm


def mn_kl_divergence(num_users, ratings, mean, cov_useritems, cov_latents,
                     sigma_sq, sigma_u_sq, sigma_v_sq, e_dot_sq=None):
    '''KL( MN(mean, cov_useritems, cov_latents) || PMF posterior ) up to a constant
    (matrix_normal_exps_cy.pyx:159-213), quirks included:
      * `num_items = num_useritems - num_useritems` = 0, so the item-covariance trace term
        is dropped (:176, :191-192);
      * the item prior uses sigma_u_sq, not sigma_v_sq (:196-197).
    e_dot_sq: optional precomputed E[(U_i.V_j)^2] for every rating (from the K2 pool kernel).'''
    num_useritems, latent_d = mean.shape
    log_det_useritems = logabsdet(cov_useritems)
    log_det_latents = logabsdet(cov_latents)
    kl = -(log_det_useritems * latent_d + log_det_latents * num_useritems) / 2.

    tr_cov_latents = cov_latents.trace()
    tr_cov_users = np.diagonal(cov_useritems)[:num_users].sum()
    tr_cov_items = 0.   # sic
    kl += ((mean[:num_users, :] ** 2).sum() + tr_cov_users * tr_cov_latents) / (2 * sigma_u_sq)
    kl += ((mean[num_users:, :] ** 2).sum() + tr_cov_items * tr_cov_latents) / (2 * sigma_u_sq)

    ii, jj, r, a, b, s, vi, vj = _rating_operands(num_users, ratings, mean, cov_useritems)
    pred = (a * b).sum(1) + s * tr_cov_latents
    if e_dot_sq is None:
        e_dot_sq = _e_dot_sq_on_device(num_users, ratings, mean, cov_useritems, cov_latents)
    bit = (e_dot_sq - 2 * r * pred + r * r).sum()
    return kl + bit / (2 * sigma_sq)


def matrixnormal_gradient(mn_apmf):
    '''Gradient of the KL w.r.t. (mean, cov_useritems, cov_latents) of an MNActivePMF
    (matrix_normal_exps_cy.pyx:216-244).  Raises TypeError if the approximation is unset.'''
    if mn_apmf is None or mn_apmf.mean is None or mn_apmf.cov_useritems is None \
            or mn_apmf.cov_latents is None:
        raise TypeError("mean, cov are None; run initialize_approx first")
    return _mnormal_grad(mn_apmf.mean, mn_apmf.cov_useritems, mn_apmf.cov_latents,
                         mn_apmf.ratings, mn_apmf.num_users, mn_apmf.latent_d,
                         mn_apmf.sigma_sq, mn_apmf.sigma_u_sq, mn_apmf.sigma_v_sq)


def _mnormal_grad(mean, cov_useritems, cov_latents, ratings, num_users, latent_d,
                  sig_sq, sig_u_sq, sig_v_sq):
    '''matrix_normal_exps_cy.pyx:395-485 with its helpers _quadexp_grad (:250-340) and
    _squareexp_grad (:344-389), restated with the sums over latent pairs (k < l) in closed
    matrix form and the sum over ratings as scatter-adds.  Like the reference, the result is
    the derivative w.r.t. one triangular half written into both halves.'''
    num_useritems = cov_useritems.shape[0]
    d = latent_d
    g_mean = np.zeros_like(mean)
    g_cu = np.zeros_like(cov_useritems)
    g_cl = np.zeros_like(cov_latents)

    ii, jj, r, a, b, s, vi, vj = _rating_operands(num_users, ratings, mean, cov_useritems)
    s_, vi_, vj_ = s[:, None], vi[:, None], vj[:, None]
    om = _upper_mirrored(cov_latents)
    dg = np.diagonal(om).copy()
    off = om - np.diag(dg)
    T = dg.sum()
    p = a * b
    P = p.sum(1, keepdims=True)
    inv = 1. / sig_sq

    # pairs k < l, each counted once with weight 1/sigma^2: terms of _quadexp_grad
    gm_i = b * (P - p) + s_ * (b @ off) + vj_ * (a @ off) + b * s_ * (T - dg)
    gm_j = a * (P - p) + vi_ * (b @ off) + s_ * (a @ off) + a * s_ * (T - dg)
    off2 = (off * off).sum() / 2.                       # sum_{k<l} Omega_kl^2
    bob = np.einsum('tk,kl,tl->t', b, off, b) / 2.      # sum_{k<l} b_k b_l Omega_kl
    aoa = np.einsum('tk,kl,tl->t', a, off, a) / 2.
    aob = np.einsum('tk,kl,tl->t', a, off, b)           # sum_{k<l} (a_k b_l + a_l b_k) Omega_kl
    sdd = (T * T - (dg * dg).sum()) / 2.                # sum_{k<l} Omega_kk Omega_ll
    pdd = (p * (T - dg)).sum(1)                         # sum_{k<l} (p_l O_kk + p_k O_ll)
    gs_ii = bob + vj * off2
    gs_jj = aoa + vi * off2
    gs_ij = pdd + aob + 2 * s * sdd + 2 * s * off2
    # squares, weight 1/(2 sigma^2): terms of _squareexp_grad
    e_i = a * a + vi_ * dg
    e_j = b * b + vj_ * dg
    gm_i += .5 * (4 * b * s_ * dg + 2 * a * e_j)
    gm_j += .5 * (4 * a * s_ * dg + 2 * b * e_i)
    gs_ii += .5 * (e_j * dg).sum(1)
    gs_jj += .5 * (e_i * dg).sum(1)
    gs_ij += .5 * (4 * (p + s_ * dg) * dg).sum(1)
    # linear term  -R_ij E[U_ik V_jk] / sigma^2
    gm_i -= r[:, None] * b
    gm_j -= r[:, None] * a
    gs_ij -= r * T

    np.add.at(g_mean, ii, inv * gm_i)
    np.add.at(g_mean, jj, inv * gm_j)
    np.add.at(g_cu, (ii, ii), inv * gs_ii)
    np.add.at(g_cu, (jj, jj), inv * gs_jj)
    np.add.at(g_cu, (ii, jj), inv * gs_ij)
    np.add.at(g_cu, (jj, ii), inv * gs_ij)

    # cov_latents: off-diagonal (both halves), then the diagonal
    G = ((b * vi_).T @ b + (b * s_).T @ a + (a * s_).T @ b + (a * vj_).T @ a
         + 2 * om * ((vi * vj).sum() + (s * s).sum()))
    g_cl += inv * (G - np.diag(np.diagonal(G)))
    diag_q = (s_ * (P - p) + (s_ * s_) * (T - dg)).sum(0)
    diag_s = .5 * (4 * p * s_ + 4 * s_ * s_ * dg + vi_ * e_j + e_i * vj_).sum(0)
    diag_l = -(r * s).sum()
    g_cl[np.arange(d), np.arange(d)] += inv * (diag_q + diag_s + diag_l)

    # priors  sum E[U_ik^2] / (2 sigma_u^2)  (+ same for V)
    g_mean[:num_users, :] += mean[:num_users, :] / sig_u_sq
    g_mean[num_users:, :] += mean[num_users:, :] / sig_v_sq
    tr_latents = cov_latents.trace()
    user_idx = np.arange(num_users)
    item_idx = np.arange(num_users, num_useritems)
    g_cu[user_idx, user_idx] += tr_latents / (2 * sig_u_sq)
    g_cu[item_idx, item_idx] += tr_latents / (2 * sig_v_sq)
    latent_idx = np.arange(d)
    g_cl[latent_idx, latent_idx] += cov_useritems[user_idx, user_idx].sum() / (2 * sig_u_sq)
    g_cl[latent_idx, latent_idx] += cov_useritems[item_idx, item_idx].sum() / (2 * sig_v_sq)

    # entropy  d/2 ln det(Sigma) + (n+m)/2 ln det(Omega), w.r.t. a triangular half
    inv_cu = inverse(cov_useritems)
    g_cu -= d / 2. * (inv_cu + inv_cu.T * (1 - np.eye(num_useritems)))
    inv_cl = inverse(cov_latents)
    g_cl -= num_useritems / 2. * (inv_cl + inv_cl.T * (1 - np.eye(d)))
    return g_mean, g_cu, g_cl
