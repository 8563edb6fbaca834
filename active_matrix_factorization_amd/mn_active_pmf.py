#!/usr/bin/env python3
"""Drop-in for the reference's ``mn_active_pmf`` module (python-pmf/mn_active_pmf.py):
active learning on a PMF model with a matrix-normal variational posterior
MN(mean, cov_useritems, cov_latents).

Same class name, constructor, criteria (with their do_normal_fit / spawn_processes /
nice_name / chooser attributes), KEY_FUNCS, compare(), full_test() and CLI flags.  What
changed is where the time went: scoring every unobserved (i, j) - the per-candidate Python
loop through exp_dotprod_sq in the reference - is one batched HIP launch here (K2,
csrc/k2_score.hip), in float64 so that the selected indices match the reference.
"""
import numpy as np

try:
    from . import _lib
    from ._active_base import (_ActiveBase, _Evaluator, project_psd, strictmap, do_normal_fit,
                               spawn_processes, nice_name, minimize, maximize, make_key_funcs,
                               query_loop, run_compare)
    from .matrix_normal_exps_cy import (quadexpect, exp_a2bc, exp_dotprod_sq, mn_kl_divergence,
                                        matrixnormal_gradient, logabsdet)
    from .pmf_cy import ProbabilisticMatrixFactorization, parse_fit_type  # noqa: F401
except ImportError:      # top-level import (package directory on sys.path, like python-pmf/)
    import _lib
    from _active_base import (_ActiveBase, _Evaluator, project_psd, strictmap, do_normal_fit,
                              spawn_processes, nice_name, minimize, maximize, make_key_funcs,
                              query_loop, run_compare)
    from matrix_normal_exps_cy import (quadexpect, exp_a2bc, exp_dotprod_sq, mn_kl_divergence,
                                       matrixnormal_gradient, logabsdet)
    from pmf_cy import ProbabilisticMatrixFactorization, parse_fit_type  # noqa: F401

MNActivePMFEvaluator = _Evaluator


class MNActivePMF(_ActiveBase):
    def __init__(self, rating_tuples, latent_d=1, rating_values=None,
                 discrete_expectations=False, refit_lookahead=False, knowable=None,
                 fit_type=('batch',)):
        super(MNActivePMF, self).__init__(rating_tuples, latent_d, rating_values,
                                          discrete_expectations, refit_lookahead, knowable,
                                          fit_type)
        # parameters of the matrix-normal approximation
        self.mean = None
        self.cov_useritems = None
        self.cov_latents = None

    # ------------------------------------------------------------------ the approximation
    def initialize_approx(self, random_cov=False):
        '''Mean at the MAP estimate; identity (or random PSD) covariances
        (mn_active_pmf.py:202-219).'''
        self.mean = np.vstack((self.users, self.items))
        n_useritems = self.num_users + self.num_items
        if random_cov:
            a = np.random.normal(size=(n_useritems, n_useritems))
            b = np.random.normal(size=(self.latent_d, self.latent_d))
            self.cov_useritems = np.dot(a, a.T)
            self.cov_latents = np.dot(b, b.T)
        else:
            self.cov_useritems = np.eye(n_useritems)
            self.cov_latents = np.eye(self.latent_d)

    def _require_approx(self, *parts):
        if any(p is None for p in parts):
            raise ValueError("run initialize_approx first")

    def kl_divergence(self, mean=None, cov_useritems=None, cov_latents=None):
        '''KL(approximation || PMF model), up to an additive constant.'''
        mean = self.mean if mean is None else mean
        cov_useritems = self.cov_useritems if cov_useritems is None else cov_useritems
        cov_latents = self.cov_latents if cov_latents is None else cov_latents
        self._require_approx(mean, cov_useritems, cov_latents)
        return mn_kl_divergence(self.num_users, self.ratings, mean, cov_useritems, cov_latents,
                                self.sigma_sq, self.sigma_u_sq, self.sigma_v_sq,
                                _device_ratings=self._device_ratings()[1])

    def _fit_normal_device(self, max_steps):
        """fit_normal_kls with the approximation resident on the device (apmf_mn_fit_normal):
        gradient, trial point, both projections, KL and the accept / reject decision of the
        learning-rate search run without the (n+m)^2 covariance crossing PCIe; yields
        (kl_trace, converged) per call, the host arrays re-bound after every call like the
        reference re-binds them after every accepted step (mn_active_pmf.py:242-288)."""
        self._require_approx(self.mean, self.cov_useritems, self.cov_latents)
        ctx, rt = self._device_ratings()
        dm = ctx.array(self.mean, np.float64)
        dcu = ctx.array(self.cov_useritems, np.float64)
        dcl = ctx.array(self.cov_latents, np.float64)
        lr, old_kl, converged = self.normal_learning_rate, None, False
        while not converged:
            trace, steps, lr, converged, old_kl = ctx.mn_fit_normal(
                rt, self.latent_d, dm, dcu, dcl, float(self.sigma_sq), float(self.sigma_u_sq),
                float(self.sigma_v_sq), self.min_eig, lr, old_kl, max_steps=max_steps)
            if steps:
                self.mean, self.cov_useritems, self.cov_latents = dm.to_host(), dcu.to_host(), dcl.to_host()
            yield trace, converged

    def fit_normal_kls(self):
        '''Projected gradient descent on the KL, yielding it after each accepted step
        (mn_active_pmf.py:242-288).'''
        for trace, _ in self._fit_normal_device(max_steps=1):
            for kl in trace:
                yield kl

    def fit_normal(self):
        '''The whole fit in one device call (same steps as exhausting fit_normal_kls).'''
        for _ in self._fit_normal_device(max_steps=-1):
            pass

    def mean_meandiff(self):
        '''Mean |approximation mean - MAP value|.'''
        return np.abs(self.mean - np.vstack((self.users, self.items))).mean()

    def _approx_entropy(self):
        '''Entropy of the approximation, up to an additive constant.'''
        ui_logdet = logabsdet(self.cov_useritems)
        l_logdet = logabsdet(self.cov_latents)
        return 0.5 * (self.latent_d * ui_logdet
                      + (self.num_users + self.num_items) * l_logdet)

    # ------------------------------------------------------------------ GPU scoring (K2)
    def _upload_state(self, state=None):
        mean, cov_ui, cov_lat = state if state is not None else \
            (self.mean, self.cov_useritems, self.cov_latents)
        self._require_approx(mean, cov_ui, cov_lat)
        ctx = self._device().ctx
        n, m = self.num_users, self.num_items
        cov_ui = np.asarray(cov_ui, dtype=np.float64)
        cross = ctx.upload_grid(cov_ui[:n, n:], np.float64)
        return (ctx, ctx.array(mean, np.float64), cross,
                ctx.array(np.ascontiguousarray(np.diagonal(cov_ui)), np.float64),
                ctx.array(cov_lat, np.float64), cross.shape[1])

    def _score_grid(self, state=None, want_mean=True):
        """(mean, var) of U_i.V_j over the whole n x m grid: one launch of the f64 sweep."""
        ctx, dm, dc, dd, do, ld = self._upload_state(state)
        n, m, d = self.num_users, self.num_items, self.latent_d
        var = ctx.empty((n, ld))
        pm = ctx.empty((n, ld)) if want_mean else None
        ctx.mn_score_grid(np.float64, 1, n, m, d, ld, dm, dc, dd, do, var=var, pmean=pm)
        return (ctx.download_grid(pm, m) if want_mean else None), ctx.download_grid(var, m)

    def _score_pool(self, pi, pj, state=None):
        n, m, d = self.num_users, self.num_items, self.latent_d
        P = int(pi.size)
        if P == 0:
            return np.zeros(0), np.zeros(0)
        if (pi.min() < 0 or pi.max() >= n or pj.min() < 0 or pj.max() >= m):
            raise IndexError("candidate index out of bounds")
        if P * 4 >= n * m:
            # most of the grid is wanted: sweep it all (coalesced) and pick
            pm, pv = self._score_grid(state)
            return pm[pi, pj], pv[pi, pj]
        ctx, dm, dc, dd, do, ld = self._upload_state(state)
        out_m, out_v = ctx.empty(P), ctx.empty(P)
        ctx.mn_score_pool(n, m, d, ld, dm, dc, dd, do, ctx.array(pi, np.int32),
                          ctx.array(pj, np.int32), P, out_m, out_v)
        return out_m.to_host(), out_v.to_host()

    def approx_pred_means_vars(self):
        '''Mean and variance of the predicted R matrix under the approximation, as two
        (n, m) arrays (mn_active_pmf.py:317-330).'''
        return self._score_grid()

    def _total_variance(self):
        # only the fused sum leaves the device (mn_active_pmf.py:597-598)
        ctx, dm, dc, dd, do, ld = self._upload_state()
        total = ctx.empty(1)
        ctx.mn_score_grid(np.float64, 1, self.num_users, self.num_items, self.latent_d, ld, dm, dc,
                          dd, do, dsum=total)
        return float(total.to_host()[0])


################################################################################

def _mean_info(apmf):
    return "Mean diff of means: %g; mean useritems cov %g, latents cov %g" % (
        apmf.mean_meandiff(), np.abs(apmf.cov_useritems.mean()), np.abs(apmf.cov_latents.mean()))


def full_test(apmf, real, picker_key=MNActivePMF.pred_variance, fit_normal=True, fit_sigmas=False,
              processes=None, test_on=None):
    return query_loop(apmf, real, picker_key, fit_normal, fit_sigmas, test_on=test_on,
                      with_pred=True, initial_fit=True, refit_with_lls=True)


def _full_test_threaded(apmf, real, picker_key, fit_normal, fit_sigmas, worker_pool=None,
                        test_on=None):
    return query_loop(apmf, real, picker_key, fit_normal, fit_sigmas, test_on=test_on,
                      with_pred=True, matrix_vals=True)


KEY_FUNCS = make_key_funcs(MNActivePMF)


def compare(key_names, real, ratings, rating_vals=None, latent_d=5, knowable=None, test_on=None,
            processes=None, do_threading=True, steps=None, discrete_exp=False,
            refit_lookahead=False, fit_sigmas=False, apmf=None,
            sig_u_mean=0, sig_u_var=-1, sig_v_mean=0, sig_v_var=-1, fit_type=('batch',)):
    return run_compare(MNActivePMF, KEY_FUNCS, key_names, real, ratings, rating_vals, latent_d,
                       knowable, test_on, steps, discrete_exp, refit_lookahead, fit_sigmas, apmf,
                       (sig_u_mean, sig_u_var, sig_v_mean, sig_v_var), fit_type, do_threading,
                       with_pred=True)


def main():
    try:
        from ._cli import run_cli
    except ImportError:
        from _cli import run_cli
    run_cli(KEY_FUNCS, compare, kind='mn')


if __name__ == '__main__':
    main()
