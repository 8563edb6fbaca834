#!/usr/bin/env python3
"""Drop-in for the reference's ``bayes_pmf`` module (python-pmf/bayes_pmf.py): Bayesian PMF
by Gibbs sampling (Salakhutdinov & Mnih 2008) and sample-based active-learning criteria.

Same class surface, KEYS, full_test(), compare_active() and CLI flags.  On the GPU:
  * samples(): every half-sweep of `num_users` / `num_items` independent sample_feature()
    calls is ONE launch of the batched normal-equation build/solve kernel (K3,
    csrc/k3_gibbs.hip).  All random draws stay on the host with NumPy's global RNG and are
    made in the reference's order (hyper-parameters, then one N(0,1)^d per row in row
    order), so a seeded chain follows the reference's.
  * predict / pred_variance / prob_ge_cutoff / total_variance: running statistics over
    the S sampled (U, V) pairs per candidate in one kernel (K4, csrc/k4_samples.hip); the S
    dense (n, m) products of the reference are never materialised.
Hyper-parameter sampling (sample_hyperparam / sample_wishart) is d x d host work and keeps
the reference's arithmetic, quirks included.
"""
from collections import namedtuple
from copy import deepcopy
from itertools import islice, repeat
import os
import random
import warnings

import numpy as np
from scipy import stats, integrate

try:
    from . import _lib
    from .pmf_cy import ProbabilisticMatrixFactorization, rmse, parse_fit_type  # noqa: F401
except ImportError:      # top-level import (package directory on sys.path, like python-pmf/)
    import _lib
    from pmf_cy import ProbabilisticMatrixFactorization, rmse, parse_fit_type  # noqa: F401


def _wishart_variates(n, dof):
    """The random numbers sample_wishart consumes, drawn with the reference's calls in its order
    (they do not depend on the scale matrix, so a whole chain's draws can be made ahead of time)."""
    if dof <= 81 + n and dof == round(dof):
        return ('direct', np.random.normal(size=(n, int(dof))))
    chi = np.random.chisquare(dof - np.arange(0, n), size=n)
    return ('bartlett', chi, np.random.normal(size=(n * (n - 1) // 2)))


def _pack_variates(v):
    """(Wishart variates, z) of one sample_hyperparam call -> (flat array in k7_hyper.hip's layout, direct_dof):
    Bartlett [chi | strict lower triangle, row-major | z] (direct_dof 0) or the small-dof scheme [G (d x dof) | z]."""
    w, z = v
    if w[0] == 'direct':
        return np.concatenate([np.ascontiguousarray(w[1]).ravel(), z]), int(w[1].shape[1])
    return np.concatenate([w[1], w[2], z]), 0


def _wishart_from_variates(sigma, variates):
    n = sigma.shape[0]
    chol = np.linalg.cholesky(sigma)
    if variates[0] == 'direct':
        X = np.dot(chol, variates[1])
    else:
        A = np.diag(np.sqrt(variates[1]))
        A[np.tri(n, k=-1, dtype=bool)] = variates[2]
        X = np.dot(chol, A)
    return np.dot(X, X.T)


def sample_wishart(sigma, dof):
    '''A draw from the Wishart distribution (bayes_pmf.py:41-59, after M. J. Johnson); the
    direct scheme for small integer dof, Bartlett's otherwise - same RNG call sequence.'''
    return _wishart_from_variates(sigma, _wishart_variates(sigma.shape[0], dof))


def iter_mean(iterable):
    i = iter(iterable)
    total = next(i)
    count = -1
    for count, x in enumerate(i):
        total += x
    return total / (count + 2)


class BayesianPMF(ProbabilisticMatrixFactorization):
    def __init__(self, rating_tuples, latent_d=5, subtract_mean=True, rating_values=None,
                 discrete_expectations=True, num_integration_pts=50, knowable=None,
                 fit_type=('batch',)):
        super(BayesianPMF, self).__init__(rating_tuples, latent_d=latent_d,
                                          subtract_mean=subtract_mean, knowable=knowable,
                                          fit_type=fit_type)
        if rating_values is not None:
            rating_values = set(map(float, rating_values))
            if not rating_values.issuperset(self.ratings[:, 2]):
                raise ValueError("got ratings not in rating_values")
        self.rating_values = rating_values
        self.discrete_expectations = discrete_expectations
        self.num_integration_pts = num_integration_pts
        self.beta = 2  # observation noise precision
        # Opt-in: draw the per-row N(0,1)^d of the Gibbs sweeps on the device
        # (Philox) instead of np.random.normal on the host.  Changes the random stream, so results no
        # longer match a reference run with the same seed; the hyper-parameter draws stay on the host.
        self.device_rng = bool(int(os.environ.get('APMF_DEVICE_RNG', '0')))
        # Gaussian-Wishart hyper-priors: (scale matrix, b0, degrees of freedom, mu0)
        self.u_hyperparams = (np.eye(latent_d), 2, latent_d, np.zeros(latent_d))
        self.v_hyperparams = (np.eye(latent_d), 2, latent_d, np.zeros(latent_d))

    def __copy__(self):
        res = BayesianPMF(self.ratings, self.latent_d)
        res.__setstate__(self.__getstate__())
        return res

    def __deepcopy__(self, memodict):
        rec = self._recipe_clone()
        res = BayesianPMF(self.ratings, self.latent_d, **({} if rec is None else {'knowable': rec}))
        res.__setstate__(deepcopy(self.__getstate__(_pairs=rec is None), memodict))
        return res

    def __getstate__(self, _pairs=True):
        state = super(BayesianPMF, self).__getstate__(_pairs)
        state['__dict__'] = self._extra_dict()
        return state

    def _set_rating_values(self, vals):
        if vals:
            vals = tuple(sorted(vals))
            self._rating_values = vals
            edges = np.empty(len(vals) + 2)
            edges[0], edges[1:-1], edges[-1] = -np.inf, vals, np.inf
            self._rating_bounds = (edges[1:] + edges[:-1]) / 2
        else:
            self._rating_values = None
            self._rating_bounds = None

    rating_values = property(lambda self: self._rating_values, _set_rating_values)
    rating_bounds = property(lambda self: self._rating_bounds)

    # ------------------------------------------------------------------ Gibbs sampler
    def sample_hyperparam(self, feats, do_users):
        '''(mu, alpha) ~ Gaussian-Wishart posterior given the feature matrix
        (bayes_pmf.py:157-186).  Host d x d work, arithmetic as in the reference - including
        np.dot of the two 1-D vectors mu0 - x_bar, which is their INNER product (:171-176).'''
        wi, b0, df, mu0 = self.u_hyperparams if do_users else self.v_hyperparams
        return self._hyper_from_variates(feats, do_users, self._hyper_variates(feats.shape[0], do_users))

    def _hyper_variates(self, N, do_users):
        """What one sample_hyperparam call draws, in its order: the Wishart's variates, then N(0,1)^d."""
        df = (self.u_hyperparams if do_users else self.v_hyperparams)[2]
        return _wishart_variates(self.latent_d, df + N), np.random.normal(0, 1, self.latent_d)

    def _hyper_from_variates(self, feats, do_users, variates):
        wi, b0, df, mu0 = self.u_hyperparams if do_users else self.v_hyperparams
        N = feats.shape[0]
        x_bar = np.mean(feats, axis=0).T
        S_bar = np.cov(feats, rowvar=0)
        mu0_xbar = mu0 - x_bar
        WI_post = np.linalg.inv(np.linalg.inv(wi) + N * S_bar
                                + (b0 * N) / (b0 + N) * np.dot(mu0_xbar, mu0_xbar.T))
        WI_post /= 2
        WI_post = WI_post + WI_post.T
        alpha = _wishart_from_variates(WI_post, variates[0])
        mu_temp = (b0 * mu0 + N * x_bar) / (b0 + N)
        lam = np.linalg.cholesky(np.linalg.inv((b0 + N) * alpha))
        mu = np.dot(lam, variates[1]) + mu_temp
        return mu, alpha

    def _mean_sub(self):
        return float(self.mean_rating) if self.subtract_mean else 0.

    def sample_feature(self, n, is_user, mu, alpha, oth_feats, rated_indices, ratings):
        '''One user/item feature vector given the other side's features
        (bayes_pmf.py:189-216): the K3 kernel on a single row; the N(0,1)^d draw is made here
        with np.random, like the reference.'''
        ctx = self._device().ctx
        oth_feats = np.ascontiguousarray(oth_feats, np.float64)
        rated_indices = np.asarray(rated_indices)
        d = self.latent_d
        rows = np.column_stack([np.zeros(rated_indices.size), rated_indices,
                                np.asarray(ratings, np.float64)])
        rt = ctx.ratings(rows, 1, oth_feats.shape[0])
        z = np.random.normal(0, 1, d)
        out = ctx.empty((1, d))
        ctx.gibbs_half_sweep(rt, 0, ctx.array(oth_feats), d, ctx.array(np.asarray(alpha, np.float64)),
                             ctx.array(np.asarray(mu, np.float64)), float(self.beta), self._mean_sub(),
                             ctx.array(z.reshape(1, d)), out)
        rt.free()
        return out.to_host()[0]

    def samples(self, num_gibbs=2, fit_first=False):
        '''Markov chain from the current MAP estimate; yields (user_sample, item_sample)
        forever (bayes_pmf.py:227-302).  Ratings added after the iterator started are not
        seen.  Each row needs >= 1 rating (the reference raises KeyError otherwise; here the
        kernel call raises ValueError).'''
        ctx = self._device().ctx
        n, m, d = self.num_users, self.num_items, self.latent_d
        rt = ctx.ratings(self.ratings, n, m)     # snapshot of the ratings, like the index dicts
        if fit_first:
            self.do_fit()
        user_sample = self.users.copy()
        item_sample = self.items.copy()
        beta, sub = float(self.beta), self._mean_sub()
        d_users, d_items = ctx.array(user_sample), ctx.array(item_sample)
        d_z_u, d_z_v = ctx.empty((n, d)), ctx.empty((m, d))
        device_rng = getattr(self, 'device_rng', False)
        if device_rng:      # seeded from NumPy's generator: np.random.seed still fixes the chain
            rng_seed, rng_ctr = int(np.random.randint(0, 2 ** 31 - 1)) * 2654435761 + 1, 0
        device_hyper = device_rng and d <= 32
        if device_hyper:
            # hyper-priors and outputs resident: the whole yield is enqueued without a host hop
            hp_u = tuple(ctx.array(np.asarray(x, np.float64)) for x in (self.u_hyperparams[0], self.u_hyperparams[3]))
            hp_v = tuple(ctx.array(np.asarray(x, np.float64)) for x in (self.v_hyperparams[0], self.v_hyperparams[3]))
            d_mu_u, d_al_u, d_mu_v, d_al_v = ctx.empty(d), ctx.empty((d, d)), ctx.empty(d), ctx.empty((d, d))
            d_status = ctx.zeros(1, np.int32)
        while True:
            if device_hyper:
                # sample_hyperparam (bayes_pmf.py:157-186) on the device, from the resident samples
                ctx.gibbs_sample_hyperparam(d_users, n, d, hp_u[0], self.u_hyperparams[1], self.u_hyperparams[2],
                                            hp_u[1], d_mu_u, d_al_u, seed=rng_seed ^ 0x5DEECE66D, offset=rng_ctr,
                                            status=d_status)
                ctx.gibbs_sample_hyperparam(d_items, m, d, hp_v[0], self.v_hyperparams[1], self.v_hyperparams[2],
                                            hp_v[1], d_mu_v, d_al_v, seed=rng_seed ^ 0x5DEECE66D,
                                            offset=rng_ctr + 1024, status=d_status)
                rng_ctr += 2048
            else:
                mu_u, alpha_u = self.sample_hyperparam(user_sample, True)
                mu_v, alpha_v = self.sample_hyperparam(item_sample, False)
                d_mu_u, d_al_u = ctx.array(mu_u, np.float64), ctx.array(alpha_u, np.float64)
                d_mu_v, d_al_v = ctx.array(mu_v, np.float64), ctx.array(alpha_v, np.float64)
            for gibbs in range(num_gibbs):
                # one N(0,1)^d per user in user order == one (n, d) draw
                if device_rng:
                    ctx.fill_normal(d_z_u, rng_seed, rng_ctr)
                    rng_ctr += (n * d + 1) // 2
                else:
                    d_z_u.upload(np.random.normal(0, 1, (n, d)))
                new_users = ctx.empty((n, d))
                ctx.gibbs_half_sweep(rt, 0, d_items, d, d_al_u, d_mu_u, beta, sub, d_z_u, new_users)
                d_users = new_users
                # items against the NEW users
                if device_rng:
                    ctx.fill_normal(d_z_v, rng_seed, rng_ctr)
                    rng_ctr += (m * d + 1) // 2
                else:
                    d_z_v.upload(np.random.normal(0, 1, (m, d)))
                new_items = ctx.empty((m, d))
                ctx.gibbs_half_sweep(rt, 1, d_users, d, d_al_v, d_mu_v, beta, sub, d_z_v, new_items)
                d_items = new_items
            user_sample, item_sample = d_users.to_host(), d_items.to_host()
            if device_hyper and int(d_status.to_host()[0]):
                raise np.linalg.LinAlgError("a matrix of the hyper-parameter step is not positive definite")
            yield user_sample, item_sample

    def samples_parallel(self, num_gibbs=2, pool=None, multiproc_mode=None, fit_first=False):
        '''bayes_pmf.py:306-424 parallelised rows over a process pool; the batched kernel
        already updates all rows at once, so this is samples() (pool ignored: a HIP context
        cannot be used from forked workers).'''
        if multiproc_mode == 'force' and pool is None:
            raise ValueError("need a process pool if multiproc is forced")
        return self.samples(num_gibbs=num_gibbs, fit_first=fit_first)

    # ------------------------------------------------------------------ sample statistics (K4)
    def _stats(self, samples_iter, cutoff=0., want=('mean',)):
        samps = list(samples_iter)
        if not samps:
            raise ValueError("no samples")
        Us = np.ascontiguousarray([u for u, v in samps], np.float64)
        Vs = np.ascontiguousarray([v for u, v in samps], np.float64)
        S, n, d = Us.shape
        m = Vs.shape[1]
        ctx = self._device().ctx
        ld = _lib.grid_ld(m)
        out = {k: ctx.empty((n, ld)) for k in want}
        ctx.sample_stats(S, n, m, d, ld, ctx.array(Us), ctx.array(Vs),
                         float(self.mean_rating) if self.subtract_mean else 0., float(cutoff),
                         mean=out.get('mean'), var=out.get('var'), prob=out.get('prob'))
        return {k: ctx.download_grid(g, m) for k, g in out.items()}

    def matrix_results(self, vals, which):
        res = np.empty((self.num_users, self.num_items))
        res.fill(np.nan)
        res[which] = vals
        return res

    def predict(self, samples_iter, which=Ellipsis):
        '''Mean reconstruction over the samples (bayes_pmf.py:433-438).'''
        return self._stats(samples_iter, want=('mean',))['mean'][which]

    def pred_variance(self, samples_iter, which=Ellipsis):
        '''Variance (ddof 0) of each prediction over the samples (bayes_pmf.py:440-448).'''
        if which is None:
            which = Ellipsis
        return self._stats(samples_iter, want=('var',))['var'][which]

    def total_variance(self, samples_iter, which=Ellipsis):
        return self.pred_variance(samples_iter, which=which).sum()

    def prob_ge_cutoff(self, samples_iter, cutoff, which=Ellipsis):
        '''Fraction of samples in which each entry is >= cutoff (bayes_pmf.py:528-538).'''
        return self._stats(samples_iter, cutoff=cutoff, want=('prob',))['prob'][which]

    def random(self, samples_iter, which=Ellipsis):
        shape = np.empty((self.num_users, self.num_items))[which].shape
        return np.random.rand(*shape)

    def bayes_rmse(self, samples_iter, true_r, which=Ellipsis):
        return rmse(self.predict(samples_iter, which), true_r[which])

    # ------------------------------------------------------------------ lookahead
    def exp_variance(self, samples_iter, which=Ellipsis, pool=None, fit_first=True, num_samps=30):
        '''Expected total variance if R_ij were known, for each ij in which
        (bayes_pmf.py:457-525).  Each hypothesis re-runs the sampler (K3) and reduces with K4;
        evaluated in this process (`pool` ignored).'''
        return self._distribute(_exp_variance_helper, samples_iter, which, None, fit_first, num_samps)

    def _distribute(self, fn, samples_iter, which, pool, fit_first, num_samps):
        n, m = self.num_users, self.num_items
        ii, jj = np.meshgrid(np.arange(n), np.arange(m), indexing='ij')
        i_indices, j_indices = ii[which], jj[which]
        samps = list(samples_iter)
        vals = np.asarray([self.predicted_matrix(u, v)[which] for u, v in samps])
        if self.discrete_expectations and self.rating_values is not None:
            discrete = True
            alpha = .1
            prev_samps = vals.shape[0]
            denom = prev_samps + alpha * len(self.rating_values)
            params = [(np.histogram(v, bins=self.rating_bounds)[0] + alpha) / denom
                      for v in vals.reshape(prev_samps, -1).T]
        else:
            if self.discrete_expectations and self.rating_values is None:
                warnings.warn("have no rating_values; doing continuous")
            discrete = False
            params = zip(np.mean(vals, 0).flat, np.var(vals, 0).flat)
        res = np.empty(np.shape(vals)[1:])
        res.fill(np.nan)
        if discrete and fn is _exp_variance_helper and self.fit_type == ('batch',) and \
                os.environ.get('APMF_LOOKAHEAD_BATCH', '1') != '0':
            # every hypothesis (candidate x rating value) of the pool in one batch on the device
            exps = self._exp_variance_batched(list(zip(i_indices.flat, j_indices.flat)), list(params),
                                              fit_first, num_samps)
        else:
            exps = map(fn, zip(repeat(self), i_indices.flat, j_indices.flat, repeat(discrete), params,
                               repeat(fit_first), repeat(num_samps)))
        for idx, exp in enumerate(exps):
            res.flat[idx] = exp
        return res

    def _exp_variance_batched(self, cands, params, fit_first, num_samps, num_gibbs=2):
        """E[total variance | R_ij] for every candidate (bayes_pmf.py:457-525, 560-598), all
        hypotheses "R_ij = v" sampled TOGETHER: their Gibbs chains advance by batched half-sweeps
        (K3 with the hypothesis as a grid dimension; each hypothesis is the shared ratings container
        plus one rating), their statistics come from K4 on the device-resident chain history.
        The random numbers are NumPy's, drawn per hypothesis in the order the reference's one-at-a-
        time evaluation consumes them (deep copy's constructor, then per yield: both hyper-parameter
        draws, then the sweeps' normals) - they do not depend on the state, so a chain's draws are
        made ahead of time and a seeded run gives what the sequential evaluation gives."""
        ctx = self._device().ctx
        n, m, d = self.num_users, self.num_items, self.latent_d
        values = list(self.rating_values)
        beta = float(self.beta)
        base = np.ascontiguousarray(self.ratings, np.float64)
        rt = ctx.ratings(base, n, m)
        out = [np.nan] * len(cands)
        hyps = []                      # (candidate index, value index, i, j, v)
        for c, (i, j) in enumerate(cands):
            i, j = int(i), int(j)
            if (i, j) in self.rated:
                warnings.warn("Asked to check a known entry; returning NaN")
                continue
            hyps.extend((c, k, i, j, float(v)) for k, v in enumerate(values))
        evals = np.zeros((len(cands), len(values)))
        per_hyp = num_samps * (num_gibbs * (n + m) * d * 8 + 4096) + (n + m) * d * 8 * (num_samps + 4)
        chunk = max(1, min(len(hyps), int(float(os.environ.get('APMF_LOOKAHEAD_HOST_GB', '4')) * 2 ** 30) // per_hyp,
                           4096))
        ld = _lib.grid_ld(m)
        sig = (float(self.sigma_sq), float(self.sigma_u_sq), float(self.sigma_v_sq))
        for lo in range(0, len(hyps), chunk):
            part = hyps[lo:lo + chunk]
            H = len(part)
            # ---- all random numbers of these chains, hypothesis by hypothesis (the reference's order)
            draws = []
            for _ in part:
                np.random.random((n, d))      # deepcopy -> BayesianPMF(...) draws its initial factors
                np.random.random((m, d))
                chain = []
                for s in range(num_samps):
                    hu = self._hyper_variates(n, True)
                    hv = self._hyper_variates(m, False)
                    zs = [(np.random.normal(0, 1, (n, d)), np.random.normal(0, 1, (m, d))) for _g in range(num_gibbs)]
                    chain.append((hu, hv, zs))
                draws.append(chain)
            # ---- starting points: the model's MAP factors, re-fitted with the extra rating if fit_first
            mean_sub = np.zeros(H)
            means_plus = np.zeros(H)
            for h, (_c, _k, i, j, v) in enumerate(part):
                # the mean add_rating() leaves behind (pmf_cy.pyx:152-153): np.mean of the extended column
                means_plus[h] = float(np.mean(np.append(base[:, 2], v)))
                mean_sub[h] = means_plus[h] if self.subtract_mean else 0.
            U0 = np.ascontiguousarray(np.broadcast_to(self.users, (H, n, d)), np.float64)
            V0 = np.ascontiguousarray(np.broadcast_to(self.items, (H, m, d)), np.float64)
            d_users, d_items = ctx.array(U0), ctx.array(V0)
            if fit_first:
                # the MAP re-fit of every hypothesis (bayes_pmf.py:240 under samples(fit_first=True)): all H fit_lls
                # searches as ONE chain of launches over the shared container (apmf_pmf_batch_fit_lls: bit-identical
                # to H fits on containers rebuilt with the hypothesis' rating)
                ctx.pmf_batch_fit_lls(rt, d_users, d_items, d, H, [p[2] for p in part], [p[3] for p in part],
                                      [p[4] for p in part], sig[0], sig[1], sig[2], self.subtract_mean, means_plus,
                                      float(self.learning_rate), float(self.min_learning_rate), float(self.stop_thresh))
                U0, V0 = d_users.to_host(), d_items.to_host()
            d_ms = ctx.array(mean_sub)
            d_ei = ctx.array(np.array([p[2] for p in part], np.int32), np.int32)
            d_ej = ctx.array(np.array([p[3] for p in part], np.int32), np.int32)
            d_ev = ctx.array(np.array([p[4] for p in part], np.float64))
            hist_U, hist_V = ctx.empty((H, num_samps, n, d)), ctx.empty((H, num_samps, m, d))
            users_h, items_h = U0, V0
            # the hyper-parameter step of all H chains on the device (k7_hyper.hip, one block per chain) from the
            # variates NumPy drew above - no factor leaves the device between the sweeps; APMF_BATCH_HYPER=0 (or
            # d > 32, or fewer than two rows) keeps the host arithmetic (values agree to ~1e-10, not to the bit)
            dev_hyper = d <= 32 and n >= 2 and m >= 2 and os.environ.get('APMF_BATCH_HYPER', '1') != '0'
            if dev_hyper:
                hp_u = tuple(ctx.array(np.asarray(x, np.float64)) for x in (self.u_hyperparams[0], self.u_hyperparams[3]))
                hp_v = tuple(ctx.array(np.asarray(x, np.float64)) for x in (self.v_hyperparams[0], self.v_hyperparams[3]))
                d_mu_u, d_al_u, d_mu_v, d_al_v = ctx.empty((H, d)), ctx.empty((H, d, d)), ctx.empty((H, d)), ctx.empty((H, d, d))
                d_status = ctx.zeros(1, np.int32)
            for s in range(num_samps):
                if dev_hyper:
                    for side, (d_feats, rows, hp, hyp, d_mu, d_al) in enumerate((
                            (d_users, n, hp_u, self.u_hyperparams, d_mu_u, d_al_u),
                            (d_items, m, hp_v, self.v_hyperparams, d_mu_v, d_al_v))):
                        packed = [_pack_variates(draws[h][s][side]) for h in range(H)]
                        ctx.gibbs_sample_hyperparam_batched(H, d_feats, rows, d, hp[0], hyp[1], hyp[2], hp[1],
                                                            ctx.array(np.stack([p[0] for p in packed])), packed[0][1],
                                                            d_mu, d_al, d_status)
                else:
                    if s > 0:
                        users_h, items_h = d_users.to_host(), d_items.to_host()
                    mus_u, als_u, mus_v, als_v = np.empty((H, d)), np.empty((H, d, d)), np.empty((H, d)), np.empty((H, d, d))
                    for h in range(H):
                        mus_u[h], als_u[h] = self._hyper_from_variates(users_h[h], True, draws[h][s][0])
                        mus_v[h], als_v[h] = self._hyper_from_variates(items_h[h], False, draws[h][s][1])
                    d_mu_u, d_al_u, d_mu_v, d_al_v = ctx.array(mus_u), ctx.array(als_u), ctx.array(mus_v), ctx.array(als_v)
                for g in range(num_gibbs):
                    zu = ctx.array(np.stack([draws[h][s][2][g][0] for h in range(H)]))
                    new_users = ctx.empty((H, n, d))
                    ctx.gibbs_half_sweep_batched(rt, 0, H, d_items, d, d_al_u, d_mu_u, beta, d_ms, zu, new_users,
                                                 d_ei, d_ej, d_ev)
                    d_users = new_users
                    zv = ctx.array(np.stack([draws[h][s][2][g][1] for h in range(H)]))
                    new_items = ctx.empty((H, m, d))
                    ctx.gibbs_half_sweep_batched(rt, 1, H, d_users, d, d_al_v, d_mu_v, beta, d_ms, zv, new_items,
                                                 d_ei, d_ej, d_ev)
                    d_items = new_items
                for h in range(H):
                    ctx.copy(hist_U, d_users, n * d * 8, dst_off=((h * num_samps + s) * n * d) * 8, src_off=h * n * d * 8)
                    ctx.copy(hist_V, d_items, m * d * 8, dst_off=((h * num_samps + s) * m * d) * 8, src_off=h * m * d * 8)
            if dev_hyper and int(d_status.to_host()[0]):
                raise np.linalg.LinAlgError('a matrix of a hyper-parameter step is not positive definite')
            # ---- total variance over the samples of every hypothesis (K4), from the device history
            var = ctx.empty((n, ld))
            for h, (c, k, i, j, v) in enumerate(part):
                ctx.sample_stats(num_samps, n, m, d, ld, hist_U.ptr + h * num_samps * n * d * 8,
                                 hist_V.ptr + h * num_samps * m * d * 8, float(mean_sub[h]), 0., var=var)
                evals[c, k] = ctx.download_grid(var, m).sum()
        rt.free()
        for c, (i, j) in enumerate(cands):
            if (int(i), int(j)) in self.rated:
                continue
            est = (evals[c] * params[c]).sum()
            print("\t{:>20}({},{}) {}: {: 10.2f}".format('total_variance', int(i), int(j), "summed", est))
            out[c] = est
        return out


def _integrate_lookahead(fn, bpmf, i, j, discrete, params, fit_first, num_samps):
    if (i, j) in bpmf.rated:
        warnings.warn("Asked to check a known entry; returning NaN")
        return np.nan

    def calculate_fn(v):
        b = deepcopy(bpmf)
        b.add_rating(i, j, v)
        samps = b.samples(fit_first=fit_first)
        return fn(b, islice(samps, num_samps))

    if discrete:
        evals = np.array([calculate_fn(v) for v in bpmf.rating_values])
        est = (evals * params).sum()
        s = "summed"
    else:
        mean, var = params
        dist = stats.norm(loc=mean, scale=np.sqrt(var))
        pts = dist.ppf(np.linspace(.001, .999, bpmf.num_integration_pts))
        evals = np.fromiter(map(calculate_fn, pts), float, pts.size)
        trapz = getattr(integrate, 'trapezoid', None) or integrate.trapz
        est = trapz(evals * dist.pdf(pts), pts)
        s = "from {:5.1f} to {:5.1f} in {}".format(pts[0], pts[-1], pts.size)
    name = getattr(fn, '__name__', '')
    print("\t{:>20}({},{}) {}: {: 10.2f}".format(name, i, j, s, est))
    return est


def _exp_variance_helper(args):
    return _integrate_lookahead(BayesianPMF.total_variance, *args)


################################################################################

Key = namedtuple('Key', ['nice_name', 'key_fn', 'choose_max', 'wants_pool', 'args'])

KEYS = {
    'random': Key("Random", 'random', True, False, ()),
    'pred-variance': Key("Var[R_ij]", 'pred_variance', True, False, ()),
    'exp-variance': Key("E[Var[R]]", 'exp_variance', False, True, ()),
    'pred': Key("Pred", 'predict', True, False, ()),
    'prob-ge-3.5': Key("Prob >= 3.5", 'prob_ge_cutoff', True, False, (3.5,)),
    'prob-ge-.5': Key("Prob >= .5", 'prob_ge_cutoff', True, False, (.5,)),
    'prob-ge-0': Key("Prob >= 0", 'prob_ge_cutoff', True, False, (0,)),
}


def fetch_samples(bpmf, num, *args, **kwargs):
    samps = list(islice(bpmf.samples(*args, **kwargs), num))
    return samps, bpmf.predict(samps)


def full_test(bpmf, samples, real, key_name, num_samps=128, lookahead_fit='batch',
              lookahead_samps=128, pool=None, multieval=False, init_rmse=None, test_on=Ellipsis):
    '''select -> query -> resample loop (bayes_pmf.py:682-729): np.argmax / np.argmin of the
    criterion over np.array(list(unrated)).T picks the query.'''
    key = KEYS[key_name]
    total = real.size
    picker_fn = getattr(bpmf, key.key_fn)
    chooser = np.argmax if key.choose_max else np.argmin
    if init_rmse is None:
        init_rmse = bpmf.bayes_rmse(samples, real, which=test_on)
    yield (len(bpmf.rated), init_rmse, None, None)
    while bpmf.unrated:
        print("{:<40} Picking query point {}...".format(key.nice_name, len(bpmf.rated) + 1))
        if len(bpmf.unrated) == 1:
            vals = None
            i, j = next(iter(bpmf.unrated))
        else:
            unrated = np.array(list(bpmf.unrated)).T
            which = tuple(unrated)
            evals = picker_fn(samples, *key.args, which=which)
            i, j = unrated[:, chooser(evals)]
            vals = bpmf.matrix_results(evals, which)
        bpmf.add_rating(i, j, real[i, j])
        print("{:<40} Queried ({}, {}); {}/{} known".format(key.nice_name, i, j, len(bpmf.rated), total))
        samples, pred = fetch_samples(bpmf, num_samps, fit_first=True)
        err = rmse(pred[test_on], real[test_on])
        print("{:<40} RMSE {}: {:.5}".format(key.nice_name, len(bpmf.rated), err))
        yield len(bpmf.rated), err, (i, j), vals


def compare_active(key_names, latent_d, real, ratings, rating_vals=None, discrete=True,
                   subtract_mean=True, num_steps=None, procs=None, threaded=False,
                   fit_type=('batch',), num_samps=128, test_set='all', **kwargs):
    '''bayes_pmf.py:733-825.  Keys are evaluated one after the other in this process.'''
    knowable = np.isfinite(real)
    knowable[real == 0] = 0
    pickable = knowable.copy()
    pickable[ratings[:, 0].astype(int), ratings[:, 1].astype(int)] = 0
    try:
        test_set = float(test_set)
    except ValueError:
        if test_set != 'all':
            warnings.warn("dunno what to do with test_set {}".format(test_set))
            test_set = 'all'
    if test_set == 'all':
        test_on, query_on = knowable, pickable
    else:
        if test_set % 1 == 0 and test_set != 1:
            avail_pts = list(zip(*pickable.nonzero()))
            picked_indices = random.sample(avail_pts, int(test_set))
            picker = np.zeros(pickable.shape, bool)
            picker[tuple(np.transpose(picked_indices))] = 1
        else:
            picker = np.random.binomial(1, test_set, size=pickable.shape)
        test_on = picker * pickable
        query_on = (1 - picker) * pickable
    query_set = set(zip(*query_on.nonzero()))
    print("{} points known, {} to query, testing on {}, {} knowable, {} total".format(
        ratings.shape[0], query_on.sum(), test_on.sum(), knowable.sum(), real.size))

    bpmf_init = BayesianPMF(ratings, latent_d, subtract_mean=subtract_mean,
                            rating_values=rating_vals, discrete_expectations=discrete,
                            knowable=query_set, fit_type=fit_type)
    print("Doing initial MAP fit...")
    bpmf_init.fit()
    print("Getting initial MCMC samples...")
    samples = list(islice(bpmf_init.samples(fit_first=fit_type), num_samps))
    init_rmse = bpmf_init.bayes_rmse(samples, real, test_on)
    print("Initial RMSE: {}".format(init_rmse))

    results = {
        '_real': real,
        '_ratings': ratings,
        '_rating_vals': rating_vals,
        '_initial_bpmf': deepcopy(bpmf_init),
    }
    kwargs.pop('lookahead_fit', None)
    kwargs.pop('lookahead_samps', None)
    for key_name in key_names:
        res = full_test(deepcopy(bpmf_init), samples, real, key_name, num_samps=num_samps,
                        init_rmse=init_rmse, test_on=test_on, **kwargs)
        results[key_name] = list(islice(res, num_steps))
    return results


def main():
    try:
        from ._cli import run_bayes_cli
    except ImportError:
        from _cli import run_bayes_cli
    run_bayes_cli(KEYS, compare_active)


if __name__ == '__main__':
    main()
