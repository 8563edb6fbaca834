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
import os
import threading
import time

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


class _Resident(object):
    """The approximation of one model, resident on the device: an apmf_mn_batch of one
    (csrc/k6_batch.hip).  fit_normal advances it in place and the K2 sweeps read it where it
    lies; the host arrays behind MNActivePMF.mean / cov_useritems / cov_latents are refreshed
    only when somebody looks at them."""
    __slots__ = ('batch', 'rt', 'host_fresh', 'dev_fresh')

    def __init__(self, batch, rt):
        self.batch, self.rt = batch, rt
        self.host_fresh = True      # the host arrays hold the current point
        self.dev_fresh = False      # the device holds the current point


def lookahead_budget_bytes():
    """Device memory the batched lookahead may use for hypothesis states (APMF_LOOKAHEAD_GB,
    default 96 of the 288 GB)."""
    return int(float(os.environ.get('APMF_LOOKAHEAD_GB', '96')) * (1 << 30))


# counters of the most recent batched lookahead (summed over its chunks and stream groups): MNBatch.fit_stats() keys
LAST_BATCH_STATS = {}
_stats_lock = threading.Lock()
_MIN_STREAM_GROUP = 16          # hypotheses: smaller pools stay one batch


def lookahead_streams():
    """Streams (groups of hypotheses advancing side by side) of a batched lookahead: APMF_LOOKAHEAD_STREAMS,
    default 3; 1 = one batch on the model's own stream.  (Profiled and unprofiled runs execute the same code: to
    take a `rocprofv3 --kernel-trace` of a lookahead set APMF_LOOKAHEAD_STREAMS=1 on its command line - ROCm 7.2's
    tracing tool faults in its own packet interceptor when several host threads submit at once, DESIGN.md section 4,
    K6.)"""
    if 'APMF_LOOKAHEAD_STREAMS' in os.environ:
        return max(1, min(8, int(os.environ['APMF_LOOKAHEAD_STREAMS'])))
    return 3


_workers = {}


def _stream_workers(count):
    """The worker threads of this process' stream groups (kept: each owns the context _lib.default_context gives
    its thread; per pid: threads do not survive a fork)."""
    key = os.getpid()
    pool = _workers.get(key)
    if pool is None or pool._max_workers < count:
        from concurrent.futures import ThreadPoolExecutor
        pool = _workers[key] = ThreadPoolExecutor(max_workers=8, thread_name_prefix='apmf-lookahead')
    return pool


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

    # ------------------------------------------------------------------ state: host views of a device resident
    def _part(name):
        key = '_' + name

        def get(self):
            res = self._resident_or_none()
            if res is not None and not res.host_fresh:
                self._sync_host()
            return self.__dict__.get(key)

        def set_(self, value):
            self._sync_host()                 # the other parts must not be lost
            if value is not None:
                # a private, read-only copy: an in-place edit would leave the device copy stale
                # without anybody noticing (re-bind the attribute instead, like the reference does)
                value = np.array(value, dtype=np.float64, order='C')
                value.flags.writeable = False
            self.__dict__[key] = value
            res = self._resident_or_none()
            if res is not None:
                res.dev_fresh = False
        return property(get, set_)

    mean = _part('mean')
    cov_useritems = _part('cov_useritems')
    cov_latents = _part('cov_latents')
    del _part

    def _extra_dict(self):
        # pickles keep the reference's attribute names (mn_active_pmf.py:174-178)
        self._sync_host()
        d = super(MNActivePMF, self)._extra_dict()
        for name in ('mean', 'cov_useritems', 'cov_latents'):
            d[name] = d.pop('_' + name, None)
        return d

    def _resident_or_none(self):
        dev = self.__dict__.get('_dev')
        return dev.resident if dev is not None else None

    def _sync_host(self):
        """Bring the host arrays up to date with the device resident (if it is ahead)."""
        res = self._resident_or_none()
        if res is None or res.host_fresh:
            return
        ctx = res.batch.ctx
        n_all, d = self.num_users + self.num_items, self.latent_d
        dm, ds, do = ctx.empty((n_all, d)), ctx.empty((n_all, n_all)), ctx.empty((d, d))
        res.batch.get_state(0, dm, ds, do)
        for name, arr in (('_mean', dm), ('_cov_useritems', ds), ('_cov_latents', do)):
            host = arr.to_host()
            host.flags.writeable = False
            self.__dict__[name] = host
        res.host_fresh = True

    def _resident(self):
        """The device resident, created / re-uploaded as needed."""
        self._require_approx(self.__dict__.get('_mean'), self.__dict__.get('_cov_useritems'),
                             self.__dict__.get('_cov_latents'))
        ctx, rt = self._device_ratings()
        dev = self._device()
        res = dev.resident
        if res is not None and res.rt is not rt:
            # the ratings container was rebuilt (add_rating): the batch borrowed the old one
            self._sync_host()
            res.batch.free()
            res = dev.resident = None
        if res is None:
            res = dev.resident = _Resident(ctx.mn_batch(rt, self.latent_d, 1, trace_cap=1), rt)
        if not res.dev_fresh:
            res.batch.set_state(ctx.array(self.__dict__['_mean']), ctx.array(self.__dict__['_cov_useritems']),
                                ctx.array(self.__dict__['_cov_latents']))
            res.dev_fresh = True
            res.host_fresh = True
        return res

    def add_ratings(self, extra):
        self._sync_host()
        super(MNActivePMF, self).add_ratings(extra)

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

    def _sigmas(self):
        return float(self.sigma_sq), float(self.sigma_u_sq), float(self.sigma_v_sq)

    def kl_divergence(self, mean=None, cov_useritems=None, cov_latents=None):
        '''KL(approximation || PMF model), up to an additive constant.'''
        if mean is None and cov_useritems is None and cov_latents is None:
            return float(self._resident().batch.kl(*self._sigmas())[0])
        mean = self.mean if mean is None else mean
        cov_useritems = self.cov_useritems if cov_useritems is None else cov_useritems
        cov_latents = self.cov_latents if cov_latents is None else cov_latents
        self._require_approx(mean, cov_useritems, cov_latents)
        return mn_kl_divergence(self.num_users, self.ratings, mean, cov_useritems, cov_latents,
                                self.sigma_sq, self.sigma_u_sq, self.sigma_v_sq,
                                _device_ratings=self._device_ratings()[1])

    def _fit_normal_device(self, max_steps):
        """fit_normal_kls on the device resident (apmf_mn_batch_fit, a batch of one): gradient,
        trial point, both projections, KL and the accept / reject decision of the learning-rate
        search run without the approximation ever crossing PCIe; yields (kl_trace, converged)
        per call (mn_active_pmf.py:242-288)."""
        lr, old_kl, converged = self.normal_learning_rate, None, False
        sig = self._sigmas()
        while not converged:
            res = self._resident()
            lrs, kls, steps, conv, trace = res.batch.fit(sig[0], sig[1], sig[2], self.min_eig, lr, old_kl,
                                                         max_steps=max_steps)
            lr, old_kl, converged = float(lrs[0]), float(kls[0]), bool(conv[0])
            if steps[0]:
                res.host_fresh = False
            yield trace[0, :min(int(steps[0]), trace.shape[1])], converged

    def fit_normal_kls(self):
        '''Projected gradient descent on the KL, yielding it after each accepted step
        (mn_active_pmf.py:242-288).'''
        for trace, _ in self._fit_normal_device(max_steps=1):
            for kl in trace:
                yield float(kl)

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
    def _upload_state(self, state):
        mean, cov_ui, cov_lat = state
        self._require_approx(mean, cov_ui, cov_lat)
        ctx = self._device().ctx
        n = self.num_users
        cov_ui = np.asarray(cov_ui, dtype=np.float64)
        cross = ctx.upload_grid(cov_ui[:n, n:], np.float64)
        return (ctx, ctx.array(mean, np.float64), cross,
                ctx.array(np.ascontiguousarray(np.diagonal(cov_ui)), np.float64),
                ctx.array(cov_lat, np.float64), cross.shape[1])

    def _score_grid(self, state=None, want_mean=True):
        """(mean, var) of U_i.V_j over the whole n x m grid: one launch of the f64 sweep,
        from the device resident (or from an explicitly given state)."""
        n, m, d = self.num_users, self.num_items, self.latent_d
        if state is not None:
            ctx, dm, dc, dd, do, ld = self._upload_state(state)
            var = ctx.empty((n, ld))
            pm = ctx.empty((n, ld)) if want_mean else None
            ctx.mn_score_grid(np.float64, 1, n, m, d, ld, dm, dc, dd, do, var=var, pmean=pm)
        else:
            res = self._resident()
            ctx, ld = res.batch.ctx, _lib.grid_ld(m)
            var = ctx.empty((n, ld))
            pm = ctx.empty((n, ld)) if want_mean else None
            res.batch.score_grid(0, 1, var=var, pmean=pm, ld=ld)
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
        if state is not None:
            ctx, dm, dc, dd, do, ld = self._upload_state(state)
            out_m, out_v = ctx.empty(P), ctx.empty(P)
            ctx.mn_score_pool(n, m, d, ld, dm, dc, dd, do, ctx.array(pi, np.int32),
                              ctx.array(pj, np.int32), P, out_m, out_v)
        else:
            res = self._resident()
            ctx = res.batch.ctx
            out_m, out_v = ctx.empty(P), ctx.empty(P)
            res.batch.score_pool(0, ctx.array(pi, np.int32), ctx.array(pj, np.int32), P, out_m, out_v)
        return out_m.to_host(), out_v.to_host()

    def approx_pred_means_vars(self):
        '''Mean and variance of the predicted R matrix under the approximation, as two
        (n, m) arrays (mn_active_pmf.py:317-330).'''
        return self._score_grid()

    def _total_variance(self):
        # only the fused sum leaves the device (mn_active_pmf.py:597-598)
        return float(self._resident().batch.total_variance()[0])

    def _pick_unrated_device(self, key, chooser):
        """pick_query_point over the whole unrated set for the variance criterion
        (mn_active_pmf.py:701-729 with the default pool): one masked float64 sweep with the
        arg-extremum fused, straight from the device resident; exact ties (rare) are collected on
        the device and broken by the reference's rule, first in the set's iteration order."""
        if getattr(key, 'batched', None) != '_pred_variance_pool':
            return None
        res = self._resident()
        ctx = res.batch.ctx
        n, m = self.num_users, self.num_items
        cand = self._candidates()
        ld = cand.ld
        var = ctx.empty((n, ld))
        best_val, best_idx = ctx.empty(1), ctx.empty(1, np.int64)
        want = 1 if chooser is max else -1
        res.batch.score_grid(0, 1, mask=cand.dev, var=var, want=want, best_val=best_val, best_idx=best_idx,
                             ld=ld)
        flat = int(best_idx.to_host()[0])
        if flat < 0:
            return None
        best = float(best_val.to_host()[0])
        if best != best:
            return None                  # NaN scores: let the host path reproduce python's max()
        cap = 4096
        pi, pj = ctx.empty(cap, np.int32), ctx.empty(cap, np.int32)
        count = ctx.grid_collect_eq(n, m, ld, var, cand.dev, best, pi, pj, cap)
        if count <= 1:
            return (flat // m, flat % m)
        if count > cap:
            return None                  # a massive tie: the general path
        return self._first_in_pool_order(pi.to_host()[:count], pj.to_host()[:count])

    # ------------------------------------------------------------------ batched one-step lookahead
    # Multi-GPU (BASELINE.json configs[3] "candidates sharded 8 x MI355X"): with a
    # sharding.Comm of world > 1 attached (one process per GPU, every rank holding the same model)
    # the hypotheses of a lookahead are dealt to the ranks in contiguous blocks and the values
    # all-gathered - RCCL on device buffers when the Comm has a communicator.
    lookahead_comm = None

    def _lookahead_batch(self, hyps, fn, pass_v):
        comm = self.lookahead_comm
        if comm is None or comm.world == 1 or len(hyps) < comm.world:
            return self._lookahead_batch_local(hyps, fn, pass_v)
        try:
            from . import sharding
        except ImportError:
            import sharding
        bs = sharding.block_size(len(hyps), comm.world)
        lo, hi = sharding.block_range(len(hyps), comm.rank, comm.world)
        mine = np.full(bs, np.nan)
        if hi > lo:
            mine[:hi - lo] = self._lookahead_batch_local(hyps[lo:hi], fn, pass_v)
        ctx = self._device().ctx
        if comm.ctx is ctx:
            out = comm.all_gather_blocks(ctx.array(mine)).to_host()      # RCCL (or staged in a rehearsal)
        else:
            out = comm.all_gather_blocks(mine)
        return out.reshape(-1)[:len(hyps)].tolist()

    def _lookahead_batch_local(self, hyps, fn, pass_v):
        """fn(model refitted with R_ij = v) for every hypothesis (i, j, v), all fitted together
        on the device (mn_active_pmf.py:660-672 for the whole pool at once): the hypotheses are
        H approximations starting from this model's, each with one extra rating, advanced by one
        chain of launches (apmf_mn_batch_fit); _total_variance is then one K2 sweep with
        B = H over the resident states and _approx_entropy comes from the factorisations the
        fit already made.  Other `fn`s get a real model object per hypothesis, filled from the
        batch.  With refit_lookahead every hypothesis' MAP re-fit (do_fit) runs first, on its
        own deep copy, and the approximation restarts from it (initialize_approx)."""
        n_all, d = self.num_users + self.num_items, self.latent_d
        sig = self._sigmas()
        cls = type(self)
        special = None
        if not pass_v and fn is getattr(cls, '_total_variance', None) and \
                cls._total_variance is MNActivePMF._total_variance:
            special = 'tv'
        elif not pass_v and fn is getattr(cls, '_approx_entropy', None) and \
                cls._approx_entropy is MNActivePMF._approx_entropy:
            special = 'ent'
        base = None
        if not self.refit_lookahead:
            res = self._resident()
            ctx = res.batch.ctx
            base = (ctx.empty((n_all, d)), ctx.empty((n_all, n_all)), ctx.empty((d, d)))
            res.batch.get_state(0, *base)
        ctx, rt = self._device_ratings()
        LAST_BATCH_STATS.clear()
        # Several streams: below a few hundred active hypotheses a launch of the fit's chain holds less than one
        # wave of workgroups and lasts as long as ONE of them does (0.6 ms per search iteration whatever the count),
        # so the pool is cut into groups that advance side by side, each a batch of its own on a stream (a context,
        # a host thread) of its own.  A hypothesis' arithmetic never depends on who shares its batch: same bits.
        groups = lookahead_streams()
        if groups > 1 and special is not None and len(hyps) >= 2 * _MIN_STREAM_GROUP:
            groups = min(groups, len(hyps) // _MIN_STREAM_GROUP)
            ctx.sync()                                  # the base state is complete before another stream reads it
            bounds = [len(hyps) * g // groups for g in range(groups + 1)]
            parts = [hyps[bounds[g]:bounds[g + 1]] for g in range(groups)]
            device = ctx.device
            budget = lookahead_budget_bytes() // groups

            def run(part):
                wctx = _lib.default_context(device)     # this worker thread's own context (kept with the thread)
                wrt = wctx.ratings(self.ratings, self.num_users, self.num_items)
                try:
                    return self._lookahead_run(wctx, wrt, base, part, fn, pass_v, special, budget)
                finally:
                    wrt.free()
                    wctx.trim()         # (sync + the worker context's block cache back to the driver)
            return [v for vals in _stream_workers(groups).map(run, parts) for v in vals]
        return self._lookahead_run(ctx, rt, base, hyps, fn, pass_v, special, lookahead_budget_bytes())

    def _lookahead_run(self, ctx, rt, base, hyps, fn, pass_v, special, budget):
        """One group of the batched lookahead on one context (stream): see _lookahead_batch_local."""
        n_all, d = self.num_users + self.num_items, self.latent_d
        sig = self._sigmas()
        per_hyp = (5 * n_all * n_all + (4 + d) * n_all * d + 64 * 64 * (n_all // 64 + 3)) * 8
        chunk = max(1, min(len(hyps), budget // per_hyp, 16384))
        out = []
        batch = None
        for lo in range(0, len(hyps), chunk):
            part = hyps[lo:lo + chunk]
            H = len(part)
            if batch is None or batch.H != H:
                if batch is not None:
                    batch.free()
                batch = ctx.mn_batch(rt, d, H)
            models = None
            t_refit = time.perf_counter()
            if self.refit_lookahead:
                # MAP re-fit per hypothesis (do_fit), then initialize_approx(): mean = MAP, identity
                # covariances.  The batch fit needs no model objects: fit_lls runs straight on a
                # container with the hypothesis' rating appended (other fit types, or a criterion
                # that wants the models, take the deep copies the reference makes)
                if self.fit_type == ('batch',) and special is not None:
                    # all H re-fits as one chain of launches over the SHARED container (apmf_pmf_batch_fit_lls:
                    # bit-identical to H fits on rebuilt containers); the fitted factors never leave the device
                    dd_, s_, su_, sv_, sub_, _mr = self._hyper()
                    n_u = self.num_users
                    dU = ctx.array(np.broadcast_to(self.users, (H,) + self.users.shape), np.float64)
                    dV = ctx.array(np.broadcast_to(self.items, (H,) + self.items.shape), np.float64)
                    hv = np.array([h[2] for h in part], np.float64)
                    # the mean add_rating() would leave behind (pmf_cy.pyx:152-153): np.mean of the extended column
                    means_r = [float(np.mean(np.append(self.ratings[:, 2], v))) for v in hv] if sub_ else None
                    ctx.pmf_batch_fit_lls(rt, dU, dV, d, H, [h[0] for h in part], [h[1] for h in part], hv, s_, su_,
                                          sv_, sub_, means_r, float(self.learning_rate), float(self.min_learning_rate),
                                          float(self.stop_thresh))
                    d_means = ctx.empty((H, n_all, d))
                    _lib._check(ctx._lib.apmf_d2d_2d(ctx._h, d_means.ptr, n_all * d * 8, dU.ptr, n_u * d * 8,
                                                     n_u * d * 8, H))
                    _lib._check(ctx._lib.apmf_d2d_2d(ctx._h, d_means.ptr + n_u * d * 8, n_all * d * 8, dV.ptr,
                                                     (n_all - n_u) * d * 8, (n_all - n_u) * d * 8, H))
                    means = d_means
                else:
                    models = [self._hypothesis_model(i, j, v, keep_rng=True) for i, j, v in part]
                    for apmf in models:
                        apmf.do_fit()
                    means = np.stack([np.vstack((apmf.users, apmf.items)) for apmf in models])
                batch.set_state(means if isinstance(means, _lib.DeviceArray) else ctx.array(means),
                                ctx.array(np.eye(n_all)), ctx.array(np.eye(d)), per_hypothesis=1)
                ctx.sync()
                with _stats_lock:
                    LAST_BATCH_STATS['refit_s'] = LAST_BATCH_STATS.get('refit_s', 0.) + time.perf_counter() - t_refit
            else:
                batch.set_state(*base)
            batch.set_extra([h[0] for h in part], [h[1] for h in part], [h[2] for h in part])
            batch.fit(sig[0], sig[1], sig[2], self.min_eig, self.normal_learning_rate, None, max_steps=-1)
            with _stats_lock:
                for k, v in batch.fit_stats().items():
                    LAST_BATCH_STATS[k] = LAST_BATCH_STATS.get(k, 0) + v
            if special == 'tv':
                out.extend(batch.total_variance().tolist())
            elif special == 'ent':
                out.extend(batch.entropy().tolist())
            else:
                dm, ds, do = ctx.empty((n_all, d)), ctx.empty((n_all, n_all)), ctx.empty((d, d))
                for k, (i, j, v) in enumerate(part):
                    apmf = models[k] if models is not None else self._hypothesis_model(i, j, v, keep_rng=True)
                    batch.get_state(k, dm, ds, do)
                    apmf.mean, apmf.cov_useritems, apmf.cov_latents = dm.to_host(), ds.to_host(), do.to_host()
                    out.append(fn(apmf, v=v) if pass_v else fn(apmf))
        if batch is not None:
            batch.free()
        return out


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
