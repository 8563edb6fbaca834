"""NumPy restatement of the reference's full-covariance KL gradient
(normal_exps_cy.pyx:217-303), per rating on the local 2d x 2d block and batched over the
ratings.  TEST INFRASTRUCTURE ONLY: the product computes this on the device
(csrc/k5_klgrad.hip, apmf_full_kl_gradient); this file is the independent check."""
import numpy as np


def normal_grad_numpy(mean, cov, ratings, d, u, v, sig, sig_u, sig_v):
    """normal_exps_cy.pyx:217-303 on the (N, 2d, 2d) stack of local covariance blocks: for the
    rating (i, j) the only entries touched are those of [u_0..u_{d-1}, v_0..v_{d-1}]."""
    N = ratings.shape[0]
    grad_mean = np.zeros_like(mean)
    grad_cov = np.zeros_like(cov)
    ii = ratings[:, 0].astype(np.intp)
    jj = ratings[:, 1].astype(np.intp)
    r = ratings[:, 2]
    idx = np.concatenate([u[:, ii].T, v[:, jj].T], axis=1)           # (N, 2d)
    mu = mean[idx]
    blk = cov[idx[:, :, None], idx[:, None, :]]                        # (N, 2d, 2d)
    a, b = mu[:, :d], mu[:, d:]
    Cuu, Cuv, Cvv = blk[:, :d, :d], blk[:, :d, d:], blk[:, d:, d:]
    cuu = np.einsum('tkk->tk', Cuu)
    cvv = np.einsum('tkk->tk', Cvv)
    cuv = np.einsum('tkk->tk', Cuv)
    p = a * b
    e = p + cuv                                                        # E[U_k V_k]
    Eo = e.sum(1, keepdims=True) - e                                   # sum over y != x
    off = 1. - np.eye(d)
    up = np.triu(np.ones((d, d)), 1)                                   # [k, l] = 1 for l > k

    g_mu = np.empty_like(mu)
    # pairs k < l (normal_exps_cy.pyx:239-249).  The k-side entries get their own sums over l;
    # the l-side entries all receive the SAME scalar - sum_tripexpect_ccl() sums over l before
    # `grad_mean[uli] += ...` broadcasts it - so u_x / v_x collect sum_{k < x} S(k).
    bx, ax = b[:, :, None], a[:, :, None]
    by, ay = b[:, None, :], a[:, None, :]
    ey = e[:, None, :]
    Ku = ((bx * ey + ay * Cvv + by * np.swapaxes(Cuv, 1, 2)) * up).sum(2)
    Kv = ((ax * ey + ay * Cuv + by * Cuu) * up).sum(2)
    S1 = ((p[:, :, None] * by + ax * Cvv + bx * Cuv + by * cuv[:, :, None]) * up).sum(2)
    S2 = ((p[:, :, None] * ay + ax * np.swapaxes(Cuv, 1, 2) + bx * Cuu + ay * cuv[:, :, None]) * up).sum(2)
    Lu = np.cumsum(S1, axis=1) - S1          # sum over k < x
    Lv = np.cumsum(S2, axis=1) - S2
    g_mu[:, :d] = Ku + Lu
    g_mu[:, d:] = Kv + Lv
    # squares and the linear term (:266-280)
    g_mu[:, :d] += 2 * b * cuv + a * (b * b + cvv) - b * r[:, None]
    g_mu[:, d:] += 2 * a * cuv + b * (a * a + cuu) - a * r[:, None]

    g_blk = np.zeros_like(blk)
    UU, UV, VU, VV = g_blk[:, :d, :d], g_blk[:, :d, d:], g_blk[:, d:, :d], g_blk[:, d:, d:]
    # the summed increments of _cov_4exp_grad_clcl (:251-256), one scalar per k
    S_vv = ((b[:, :, None] * b[:, None, :] + Cvv) * up).sum(2)
    S_uu = ((a[:, :, None] * a[:, None, :] + Cuu) * up).sum(2)
    S_vu = ((b[:, :, None] * a[:, None, :] + np.swapaxes(Cuv, 1, 2)) * up).sum(2)
    S_uv = ((a[:, :, None] * b[:, None, :] + Cuv) * up).sum(2)
    t_vv, t_uu = S_vv[:, :, None] * up, S_uu[:, :, None] * up
    t_vu, t_uv = S_vu[:, :, None] * up, S_uv[:, :, None] * up
    UU += t_vv + np.swapaxes(t_vv, 1, 2)
    VV += t_uu + np.swapaxes(t_uu, 1, 2)
    UV += t_vu + np.swapaxes(t_uv, 1, 2)
    VU += np.swapaxes(t_vu, 1, 2) + t_uv
    dk = np.arange(d)
    UV[:, dk, dk] += Eo + 2 * e - r[:, None]
    VU[:, dk, dk] += Eo + 2 * e - r[:, None]
    UU[:, dk, dk] += (b * b + cvv) / 2
    VV[:, dk, dk] += (a * a + cuu) / 2

    np.add.at(grad_mean, idx, g_mu / sig)
    np.add.at(grad_cov, (idx[:, :, None], idx[:, None, :]), g_blk / sig)

    # priors (:286-291) and the entropy term (:302-303)
    us, vs = u.reshape(-1), v.reshape(-1)
    grad_mean[us] += mean[us] / sig_u
    grad_mean[vs] += mean[vs] / sig_v
    grad_cov[us, us] += 1 / (2 * sig_u)
    grad_cov[vs, vs] += 1 / (2 * sig_v)
    inv = np.linalg.inv(cov)
    grad_cov -= (inv + inv.T * (1 - np.eye(cov.shape[0]))) / 2
    return grad_mean, grad_cov
