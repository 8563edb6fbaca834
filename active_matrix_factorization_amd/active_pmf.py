#!/usr/bin/env python3
"""Drop-in for the reference's ``active_pmf`` module (python-pmf/active_pmf.py): active
learning on a PMF model with a full-covariance Gaussian N(mean, cov) over the stacked
(n+m)*d latent vector.

Same class surface, KEY_FUNCS, compare(), full_test(), make_fake_data(), get_ratings().
Candidate scoring (approx_pred_means_vars / pred_variance / the criteria built on them) is
the HIP kernel K2-full (csrc/k2_full.hip) instead of a Python loop over
normal_exps_cy.exp_dotprod_sq.
"""
import itertools
import numbers
import os
import random
import warnings

import numpy as np

try:
    from . import _lib
    from ._active_base import (_ActiveBase, _Evaluator, _criterion, minimize, make_key_funcs,
                               query_loop, run_compare, strictmap, project_psd as _project_psd)
    from .matrix_normal_exps_cy import logabsdet, inverse
    from .normal_exps_cy import quadexpect, exp_a2bc, exp_dotprod_sq, normal_gradient
    from .pmf_cy import ProbabilisticMatrixFactorization, parse_fit_type  # noqa: F401
except ImportError:      # top-level import (package directory on sys.path, like python-pmf/)
    import _lib
    from _active_base import (_ActiveBase, _Evaluator, _criterion, minimize, make_key_funcs,
                              query_loop, run_compare, strictmap, project_psd as _project_psd)
    from matrix_normal_exps_cy import logabsdet, inverse
    from normal_exps_cy import quadexpect, exp_a2bc, exp_dotprod_sq, normal_gradient
    from pmf_cy import ProbabilisticMatrixFactorization, parse_fit_type  # noqa: F401

ActivePMFEvaluator = _Evaluator


def project_psd(mat, min_eig=0):
    '''Symmetrise and clip the spectrum at min_eig (active_pmf.py:36-50), on the device.'''
    return _project_psd(mat, min_eig=min_eig)


class ActivePMF(_ActiveBase):
    _approx_init_uses_rng = True      # initialize_approx() draws its covariance from np.random

    def __init__(self, rating_tuples, latent_d=1, rating_values=None,
                 discrete_expectations=False, refit_lookahead=False, knowable=None,
                 fit_type=('batch',)):
        super(ActivePMF, self).__init__(rating_tuples, latent_d, rating_values,
                                        discrete_expectations, refit_lookahead, knowable, fit_type)
        self.mean = None
        self.cov = None
        n, m, d = self.num_users, self.num_items, self.latent_d
        k = self.approx_dim
        self.num_params = k + k * (k + 1) / 2
        # mean[u[k, i]] is the mean of U_ki, mean[v[k, j]] of V_kj (active_pmf.py:139-142)
        self.u = np.arange(0, n * d).reshape(n, d).T
        self.v = np.arange(n * d, (n + m) * d).reshape(m, d).T

    # ------------------------------------------------------------------ the approximation
    def initialize_approx(self):
        '''Mean at the MAP estimate, random positive-definite covariance (active_pmf.py:189-200).'''
        self.mean = np.hstack((self.users.reshape(-1), self.items.reshape(-1)))
        s = np.random.normal(0, 2, (self.approx_dim, self.approx_dim))
        self.cov = project_psd(s, min_eig=self.min_eig)

    def kl_divergence(self, mean=None, cov=None):
        '''KL up to an additive constant (active_pmf.py:202-246).'''
        mean = self.mean if mean is None else mean
        cov = self.cov if cov is None else cov
        if mean is None or cov is None:
            raise ValueError("run initialize_approx first")
        us, vs = self.u.reshape(-1), self.v.reshape(-1)
        pi = np.ascontiguousarray(self.ratings[:, 0], np.int32)
        pj = np.ascontiguousarray(self.ratings[:, 1], np.int32)
        pm, pv = self._score_pool(pi, pj, (mean, cov))
        r = self.ratings[:, 2]
        div = ((pv + pm * pm - 2 * r * pm).sum() + (r ** 2).sum()) / (2 * self.sigma_sq)
        div += ((mean[us] ** 2).sum() + cov[us, us].sum()) / (2 * self.sigma_u_sq)
        div += ((mean[vs] ** 2).sum() + cov[vs, vs].sum()) / (2 * self.sigma_v_sq)
        return div - logabsdet(cov) / 2

    def _fit_normal_device(self, max_steps):
        """fit_normal_kls with the approximation resident on the device: a batch of ONE in the engine that also
        fits the lookahead's hypotheses (csrc/k8_fullbatch.hip) - gradient, trial point, projection, KL and the
        accept / reject decision of the search (active_pmf.py:250-288) without the K x K covariance crossing PCIe
        per trial; the host arrays are re-bound after every call, like the reference re-binds them per accepted
        step."""
        if self.mean is None or self.cov is None:
            raise ValueError("run initialize_approx first")
        ctx, rt = self._device_ratings()
        batch = ctx.full_batch(rt, self.latent_d, 1)
        try:
            dm, dc = ctx.array(self.mean, np.float64), ctx.array(self.cov, np.float64)
            batch.set_state(dm, dc)
            sig = (float(self.sigma_sq), float(self.sigma_u_sq), float(self.sigma_v_sq))
            lr, old_kl, converged = self.normal_learning_rate, None, False
            cap = 1 if max_steps == 1 else 1 << 14
            while not converged:
                lrs, kls, steps, conv, trace = batch.fit(sig[0], sig[1], sig[2], self.min_eig, lr, old_kl,
                                                         max_steps=max_steps, trace_cap=cap)
                lr, old_kl, converged = float(lrs[0]), float(kls[0]), bool(conv[0])
                if steps[0]:
                    batch.get_state(0, dm, dc)
                    self.mean, self.cov = dm.to_host(), dc.to_host()
                yield trace[0, :min(int(steps[0]), cap)], converged
        finally:
            batch.free()

    def fit_normal_kls(self):
        '''Projected gradient descent on the KL, yielding it after each accepted step
        (active_pmf.py:250-288).'''
        for trace, _ in self._fit_normal_device(max_steps=1):
            for kl in trace:
                yield float(kl)

    def fit_normal(self):
        '''The whole fit in one device call (same steps as exhausting fit_normal_kls).'''
        for _ in self._fit_normal_device(max_steps=-1):
            pass

    def mean_meandiff(self):
        p = np.hstack((self.users.reshape(-1), self.items.reshape(-1)))
        return np.abs(self.mean - p).mean()

    def _approx_entropy(self):
        return logabsdet(self.cov)

    # ------------------------------------------------------------------ GPU scoring (K2-full)
    def _score_grid(self, state=None, total_only=False):
        mean, cov = state if state is not None else (self.mean, self.cov)
        if mean is None or cov is None:
            raise ValueError("run initialize_approx first")
        ctx = self._device().ctx
        n, m, d = self.num_users, self.num_items, self.latent_d
        ld = _lib.grid_ld(m)
        dm, dc = ctx.array(mean, np.float64), ctx.array(cov, np.float64)
        if total_only:
            tot = ctx.empty(1)
            ctx.full_score_grid(n, m, d, ld, dm, dc, dsum=tot)
            return float(tot.to_host()[0])
        var, pm = ctx.empty((n, ld)), ctx.empty((n, ld))
        ctx.full_score_grid(n, m, d, ld, dm, dc, var=var, pmean=pm)
        return ctx.download_grid(pm, m), ctx.download_grid(var, m)

    def _score_pool(self, pi, pj, state=None):
        if pi.size == 0:
            return np.zeros(0), np.zeros(0)
        pm, pv = self._score_grid(state)
        return pm[pi, pj], pv[pi, pj]

    def approx_pred_means_vars(self):
        '''Mean and variance of the predicted R matrix (active_pmf.py:301-322).'''
        return self._score_grid()

    def _total_variance(self):
        return self._score_grid(total_only=True)

    # ------------------------------------------------------------------ batched one-step lookahead
    def _lookahead_batch_ok(self):
        # with refit_lookahead every hypothesis draws a fresh random covariance (initialize_approx) between its own
        # MAP re-fit and its variational fit: kept on the one-at-a-time path, in the reference's RNG order
        return not self.refit_lookahead

    def _lookahead_batch(self, hyps, fn, pass_v):
        """fn(model refitted with R_ij = v) for every hypothesis (i, j, v), all fitted TOGETHER on the device
        (active_pmf.py:660-672 for the whole pool at once): H approximations starting from this model's, each with one
        extra rating, advanced by one chain of launches (apmf_full_batch_fit); _total_variance and _approx_entropy then
        come from the resident states, any other `fn` gets a real model object per hypothesis filled from the batch.
        The same engine with H = 1 is this class' fit_normal: the values equal the one-at-a-time evaluation bit for
        bit."""
        if self.mean is None or self.cov is None:
            raise ValueError("run initialize_approx first")
        ctx, rt = self._device_ratings()
        K = self.approx_dim
        sig = (float(self.sigma_sq), float(self.sigma_u_sq), float(self.sigma_v_sq))
        cls = type(self)
        special = None
        if not pass_v and fn is getattr(cls, '_total_variance', None) and cls._total_variance is ActivePMF._total_variance:
            special = 'tv'
        elif not pass_v and fn is getattr(cls, '_approx_entropy', None) and cls._approx_entropy is ActivePMF._approx_entropy:
            special = 'ent'
        per_hyp = (6 * K * K + 4 * K + 64 * 64 * (K // 64 + 3)) * 8
        budget = int(float(os.environ.get('APMF_LOOKAHEAD_GB', '96')) * (1 << 30))
        chunk = max(1, min(len(hyps), budget // per_hyp, 16384))
        dm, dc = ctx.array(self.mean, np.float64), ctx.array(self.cov, np.float64)
        out, batch = [], None
        try:
            for lo in range(0, len(hyps), chunk):
                part = hyps[lo:lo + chunk]
                H = len(part)
                if batch is None or batch.H != H:
                    if batch is not None:
                        batch.free()
                    batch = ctx.full_batch(rt, self.latent_d, H)
                batch.set_state(dm, dc)
                batch.set_extra([h[0] for h in part], [h[1] for h in part], [h[2] for h in part])
                batch.fit(sig[0], sig[1], sig[2], self.min_eig, self.normal_learning_rate, None, max_steps=-1)
                if special == 'tv':
                    out.extend(batch.total_variance().tolist())
                elif special == 'ent':
                    out.extend(batch.logdet().tolist())
                else:
                    hm, hc = ctx.empty(K), ctx.empty((K, K))
                    for k, (i, j, v) in enumerate(part):
                        apmf = self._hypothesis_model(i, j, v, keep_rng=True)
                        batch.get_state(k, hm, hc)
                        apmf.mean, apmf.cov = hm.to_host(), hc.to_host()
                        out.append(fn(apmf, v=v) if pass_v else fn(apmf))
        finally:
            if batch is not None:
                batch.free()
        return out

    # ------------------------------------------------------------------ entropy-bound criteria
    def approx_pred_covs(self):
        '''Covariance between all pairs of predicted entries, (n*m, n*m)
        (active_pmf.py:324-390).  O((n m)^2 d^2) scalar work: tiny matrices only; not on the
        accelerated path.'''
        n, m, D = self.num_users, self.num_items, self.latent_d
        mean, cov, u, v = self.mean, self.cov, self.u, self.v
        ijs = list(itertools.product(range(n), range(m)))
        e1 = {ij: (mean[u[:, ij[0]]] * mean[v[:, ij[1]]] + cov[u[:, ij[0]], v[:, ij[1]]]).sum()
              for ij in ijs}
        out = np.zeros((n * m, n * m))
        for p, (i, j) in enumerate(ijs):
            ui, vj = u[:, i], v[:, j]
            out[p, p] = exp_dotprod_sq(u, v, mean, cov, i, j) - e1[i, j] ** 2
            for q in range(p + 1, len(ijs)):
                a, b = ijs[q]
                ua, vb = u[:, a], v[:, b]
                cv = sum(quadexpect(mean, cov, ui[k], vj[k], ua[l], vb[l])
                         for k in range(D) for l in range(D) if l != k)
                if i == a:
                    cv += sum(exp_a2bc(mean, cov, ui[k], vj[k], vb[k]) for k in range(D))
                elif j == b:
                    cv += sum(exp_a2bc(mean, cov, vj[k], ui[k], ua[k]) for k in range(D))
                else:
                    cv += sum(quadexpect(mean, cov, ui[k], vj[k], ua[k], vb[k]) for k in range(D))
                cv -= e1[i, j] * e1[a, b]
                out[p, q] = out[q, p] = cv
        return out

    def _pred_entropy_bound(self):
        s, logdet = np.linalg.slogdet(self.approx_pred_covs())
        if s != 1:
            if s == -1 and logdet < -50:
                return -1000
            raise ValueError("prediction cov has det with sign {}, log {}".format(s, logdet))
        return logdet

    @_criterion("E[Pred Entropy Bound] (MAP)", True, True, chooser=minimize, lookahead=('_pred_entropy_bound', True))
    def exp_pred_entropy_bound(self, ij):
        return self._exp_with_rij(ij, ActivePMF._pred_entropy_bound, use_map=True)

    @_criterion("E[Pred Entropy Bound] (Approx)", True, True, chooser=minimize, lookahead=('_pred_entropy_bound', False))
    def exp_pred_entropy_bound_byapprox(self, ij):
        return self._exp_with_rij(ij, ActivePMF._pred_entropy_bound, use_map=False)


################################################################################

def full_test(apmf, real, picker_key=ActivePMF.pred_variance, fit_normal=True, fit_sigmas=False,
              processes=None):
    return query_loop(apmf, real, picker_key, fit_normal, fit_sigmas, with_pred=False,
                      initial_fit=True, refit_with_lls=True)


def _full_test_threaded(apmf, real, picker_key, fit_normal, fit_sigmas, worker_pool=None):
    return query_loop(apmf, real, picker_key, fit_normal, fit_sigmas, with_pred=False,
                      matrix_vals=True)


KEY_FUNCS = make_key_funcs(ActivePMF, with_entropy_bound=True)


def make_fake_data(noise=.25, num_users=10, num_items=10, mask_type=0, data_type='float', rank=5,
                   u_mean=0, u_std=2, v_mean=0, v_std=2):
    '''Low-rank synthetic matrix + observed ratings (active_pmf.py:926-960); same RNG calls.'''
    u = np.random.normal(u_mean, u_std, (num_users, rank))
    v = np.random.normal(v_mean, v_std, (num_items, rank))
    real = np.dot(u, v.T)
    if noise:
        real += np.random.normal(0, noise, (num_users, num_items))
    vals = None
    if data_type == 'float':
        pass
    elif data_type == 'int':
        real = np.round(real).astype(int)
    elif data_type == 'int-bounds':
        real = np.round(real).astype(int)
        lo, hi = real.min(), real.max()
        vals = range(int(np.floor(lo * 1.2 if lo < 0 else lo * .8)),
                     int(np.ceil(hi * 1.2 if hi > 0 else hi * .8)))
    elif data_type == 'binary':
        real = (real > .5).astype(int)
        vals = {0, 1}
    elif isinstance(data_type, numbers.Integral):
        real = np.minimum(np.maximum(np.round(real), 0), data_type).astype(int)
        vals = range(data_type + 1)
    else:
        raise ValueError("Don't know how to interpret data_type '{}'".format(data_type))
    return real, get_ratings(real, mask_type), vals


def get_ratings(real, mask_type=0):
    '''Observed-entry mask -> (N, 3) ratings, >= 1 per row and column (active_pmf.py:963-1000).'''
    num_users, num_items = real.shape
    if isinstance(mask_type, numbers.Real):
        mask = np.random.binomial(1, mask_type, real.shape)
    elif mask_type in {'diag', 'diagonal', 'diag-plus', 'diag-block'}:
        mask = np.zeros_like(real)
        np.fill_diagonal(mask, 1)
        square = num_users == num_items
        if mask_type == 'diag-plus':
            if not square:
                warnings.warn("can't do diag-plus for non-square; doing diag")
            else:
                n = num_users
                mask[-1, 1] = 1
                mask[range(1, n - 1), range(2, n)] = 1
        elif mask_type == 'diag-block':
            if not square:
                warnings.warn("can't do diag-block for non-square; doing diag")
            else:
                mask[:num_users // 2, :num_items // 2] = 1
    else:
        raise ValueError("Don't know how to interpret mask_type '{}'".format(mask_type))
    for zero_col in np.logical_not(mask.sum(axis=0)).nonzero()[0]:
        mask[random.randrange(num_users), zero_col] = 1
    for zero_row in np.logical_not(mask.sum(axis=1)).nonzero()[0]:
        mask[zero_row, random.randrange(num_items)] = 1
    assert np.all(mask.sum(axis=0) > 0)
    assert np.all(mask.sum(axis=1) > 0)
    ratings = np.zeros((int(mask.sum()), 3))
    for idx, (i, j) in enumerate(np.transpose(mask.nonzero())):
        ratings[idx] = [i, j, real[i, j]]
    return ratings


def compare(key_names, latent_d=5, processes=None, do_threading=True, steps=None,
            discrete_exp=False, refit_lookahead=False, fit_sigmas=False, real_ratings_vals=None,
            apmf=None, knowable=None, sig_u_mean=0, sig_u_var=-1, sig_v_mean=0, sig_v_var=-1,
            fit_type=('batch',), **kwargs):
    if real_ratings_vals is None:
        real, ratings, rating_vals = make_fake_data(**kwargs)
    else:
        real, ratings, rating_vals = real_ratings_vals
    return run_compare(ActivePMF, KEY_FUNCS, key_names, real, ratings, rating_vals, latent_d,
                       knowable, None, steps, discrete_exp, refit_lookahead, fit_sigmas, apmf,
                       (sig_u_mean, sig_u_var, sig_v_mean, sig_v_var), fit_type, do_threading,
                       with_pred=False)


def main():
    try:
        from ._cli import run_cli
    except ImportError:
        from _cli import run_cli
    run_cli(KEY_FUNCS, compare, kind='full')


if __name__ == '__main__':
    main()
