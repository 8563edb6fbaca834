"""Drop-in for the reference's ``normal_exps_cy`` module (python-pmf/normal_exps_cy.pyx):
moments of a multivariate normal N(mean, cov) over the stacked latent vector and the KL
gradient used by ActivePMF.fit_normal.

The scalar moment functions are one-candidate conveniences; pools and grids of candidates
are scored by the HIP kernel behind ActivePMF (K2-full, csrc/k2_full.hip).
normal_gradient is the variational-fit gradient: a HIP kernel over the
d x d blocks of the covariance each rating touches (csrc/k5_klgrad.hip).  It reproduces the
reference's arithmetic, including the summed increment of `_cov_4exp_grad_clcl`
(normal_exps_cy.pyx:188-197) that the reference's own notebook flags as "has to be a bug" -
parity is with the code, not with the maths.
"""
import numpy as np

DTYPE = float


def tripexpect(mean, cov, a, b, c):
    '''E[X_a X_b X_c]   (normal_exps_cy.pyx:9-14)'''
    return mean[a] * mean[b] * mean[c] + mean[a] * cov[b, c] + mean[b] * cov[a, c] \
        + mean[c] * cov[a, b]


def quadexpect(mean, cov, a, b, c, d):
    '''E[X_a X_b X_c X_d], distinct indices   (normal_exps_cy.pyx:40-72)'''
    ma, mb, mc, md = mean[a], mean[b], mean[c], mean[d]
    return (ma * mb * mc * md
            + ma * mb * cov[c, d] + ma * mc * cov[b, d] + ma * md * cov[b, c]
            + mb * mc * cov[a, d] + mb * md * cov[a, c] + mc * md * cov[a, b]
            + cov[a, b] * cov[c, d] + cov[a, c] * cov[b, d] + cov[a, d] * cov[b, c])


def exp_squared(mean, cov, a, b):
    '''E[X_a^2 X_b^2]   (normal_exps_cy.pyx:76-85)'''
    return 4 * mean[a] * mean[b] * cov[a, b] + 2 * cov[a, b] ** 2 + \
        (mean[a] ** 2 + cov[a, a]) * (mean[b] ** 2 + cov[b, b])


def exp_a2bc(mean, cov, a, b, c):
    '''E[X_a^2 X_b X_c]   (normal_exps_cy.pyx:89-107)'''
    ma, mb, mc = mean[a], mean[b], mean[c]
    return ((ma ** 2 + cov[a, a]) * (mb * mc + cov[b, c])
            + 2 * ma * mc * cov[a, b] + 2 * ma * mb * cov[a, c] + 2 * cov[a, b] * cov[a, c])


def exp_dotprod_sq(u, v, mean, cov, i, j):
    '''E[(U_i . V_j)^2] for ONE candidate   (normal_exps_cy.pyx:111-135); u, v are the
    (d, n) / (d, m) index maps of ActivePMF.'''
    d = u.shape[0]
    e = 0.
    for k in range(d):
        uki, vkj = u[k, i], v[k, j]
        e += exp_squared(mean, cov, uki, vkj)
        for l in range(k + 1, d):
            e += 2 * quadexpect(mean, cov, uki, vkj, u[l, i], v[l, j])
    return e


def normal_gradient(apmf):
    '''Gradient of the KL w.r.t. (mean, cov) of an ActivePMF (normal_exps_cy.pyx:140-168).'''
    if apmf is None or apmf.mean is None or apmf.cov is None:
        raise TypeError("mean, cov are None; run initialize_approx first")
    rt = apmf._device_ratings()[1] if hasattr(apmf, '_device_ratings') else None
    return _normal_grad(apmf.mean, apmf.cov, apmf.ratings, apmf.latent_d, apmf.u, apmf.v,
                        apmf.sigma_sq, apmf.sigma_u_sq, apmf.sigma_v_sq, _device_ratings=rt)


def _normal_grad(mean, cov, ratings, d, u, v, sig, sig_u, sig_v, _device_ratings=None):
    """normal_exps_cy.pyx:217-303 on the device (K5-full, csrc/k5_klgrad.hip: one wave per user /
    item row over the d x d blocks of the covariance a rating touches; the entropy term's inverse
    by Cholesky / LU in libapmf_dense.so).  u, v are the index maps of ActivePMF
    (u[k, i] = i d + k, v[k, j] = n d + j d + k), which the kernel assumes."""
    try:
        from . import _lib
    except ImportError:
        import _lib
    n, m = u.shape[1], v.shape[1]
    if not (np.array_equal(u, np.arange(n * d).reshape(n, d).T) and
            np.array_equal(v, n * d + np.arange(m * d).reshape(m, d).T)):
        raise ValueError("u / v are not the stacked index maps of ActivePMF")
    ctx = _lib.default_context()
    rt = _device_ratings or ctx.ratings(np.ascontiguousarray(ratings, np.float64), n, m)
    K = (n + m) * d
    g_mean, g_cov = ctx.empty(K), ctx.empty((K, K))
    ctx.full_kl_gradient(rt, d, ctx.array(mean, np.float64), ctx.array(cov, np.float64), float(sig),
                         float(sig_u), float(sig_v), g_mean, g_cov)
    out = g_mean.to_host(), g_cov.to_host()
    if _device_ratings is None:
        rt.free()
    return out
