"""Independent NumPy restatement of the matrix-normal KL gradient (matrix_normal_exps_cy.pyx:
395-485), vectorised over the ratings; test-only cross-check of the K5 kernels."""
import numpy as np


def _upper_mirrored(cov_latents):
    up = np.triu(cov_latents)
    return up + np.triu(cov_latents, 1).T


def inverse(mat):
    return np.linalg.inv(mat)


def _rating_operands(num_users, ratings, mean, cov_useritems):
    ii = ratings[:, 0].astype(np.intp)
    jj = ratings[:, 1].astype(np.intp) + num_users
    return (ii, jj, ratings[:, 2], mean[ii], mean[jj], cov_useritems[ii, jj],
            cov_useritems[ii, ii], cov_useritems[jj, jj])


def _upper_mirrored(cov_latents):
    # the reference only reads cov_latents[k, l] with k <= l
    up = np.triu(cov_latents)
    return up + np.triu(cov_latents, 1).T


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
