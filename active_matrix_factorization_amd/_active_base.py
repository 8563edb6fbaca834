"""Machinery shared by the two variational active-PMF classes (MNActivePMF: matrix-normal
posterior, ActivePMF: full-covariance Gaussian).

The reference keeps two near-identical 1 100-line modules (mn_active_pmf.py, active_pmf.py).
Here everything that does not depend on the shape of the posterior lives once in
`_ActiveBase`: the criterion decorators, the selection surface (pick_query_point,
_get_key_vals, get_key_evals), the one-step lookahead integrator and the experiment loop.
What differs - how a pool of candidates is scored on the GPU and how the approximation is
fitted - is supplied by the subclass.

GPU note ("fork + HIP"): criteria are evaluated in the calling process -
a HIP context does not survive fork().  Pool-scoring criteria are one batched kernel launch;
the lookahead criteria (the ones the reference hands to its process pool) are mapped over a
small pool of threads, each with its own Context: `procs` is the number of threads (default
APMF_LOOKAHEAD_THREADS or 4; 0 or 1 = sequential), `worker_pool` is accepted and ignored.
"""
import concurrent.futures
import itertools
import math
import operator
import os
import random
import threading
import warnings
from copy import deepcopy

import numpy as np
from scipy import stats

try:
    from . import _lib
    from .pmf_cy import ProbabilisticMatrixFactorization
except ImportError:      # top-level import, like the reference's sibling modules
    import _lib
    from pmf_cy import ProbabilisticMatrixFactorization


# ---------------------------------------------------------------------------- helpers
def project_psd(mat, min_eig=0, destroy=False):
    '''Symmetrise, then clip the spectrum at min_eig (mn_active_pmf.py:42-67,
    active_pmf.py:36-50), on the device: the (n+m) x (n+m) eigendecomposition is the single
    most expensive step of the variational fit (rocSOLVER syevd + rocBLAS gemm behind
    libapmf_dense.so; 74 ms instead of 1.8 s of host LAPACK at 2625 x 2625).'''
    mat = np.ascontiguousarray(mat, np.float64)
    if mat.shape[0] <= 16:       # the d x d latent covariance: host LAPACK, like the reference
        mat = (mat + mat.T) / 2
        vals, vecs = np.linalg.eigh(mat)
        if vals.min() < min_eig:
            mat = np.dot(vecs, np.dot(np.diag(np.maximum(vals, min_eig)), vecs.T))
            mat = (mat + mat.T) / 2
        return mat
    ctx = _lib.default_context()
    dmat = ctx.array(mat)
    ctx.dense_project_psd(dmat, mat.shape[0], min_eig)
    return dmat.to_host()


def strictmap(*args, **kwargs):
    return list(map(*args, **kwargs))


# -- concurrent lookahead ---------------------------------------------------------------------
# The reference farms the lookahead criteria (spawn_processes = True) out to a forked process
# pool.  A HIP context does not survive fork(), but every hypothesis (candidate, rating value)
# is an independent deep copy whose re-fit is one long C-ABI call that releases the GIL, and a
# single re-fit is a latency-bound chain of small launches that keeps only a few CUs busy: the
# map runs on a small pool of THREADS, each with its own Context (= own stream, workspaces,
# rocBLAS handle).  Results come back in input order and are bit-identical to the sequential
# ones (tested; it is why the Cholesky factorisation is our own).  Measured on one MI355X, 20
# hypotheses at 94 x 425 (tools/probe_lookahead_threads.py): 37.5 s sequential, 14.7 s with 4
# threads - provided the runtime has a hardware queue per stream (_lib.load sets
# GPU_MAX_HW_QUEUES=8; with the default 4 two worker streams share a queue: 1.7x instead of 2.5x).
# More than 4 threads do not help (tools/probe_scaling.py).  `procs` / APMF_LOOKAHEAD_THREADS
# set the width
# (default 4; 0 or 1 = sequential).
_workers = None
_worker_state = threading.local()


def lookahead_threads(procs=None):
    if procs is not None and procs >= 0:
        return int(procs)
    override = getattr(_worker_state, 'width', None)      # a `procs` given further up the call chain
    if override is not None:
        return override
    return int(os.environ.get('APMF_LOOKAHEAD_THREADS', '4'))


def lookahead_batched():
    """APMF_LOOKAHEAD_BATCH=0 switches the one-batch-on-the-device evaluation of the lookahead
    hypotheses off (every hypothesis is then a deep copy re-fitted on its own; used by the tests
    that check the two give the same values bit for bit)."""
    return os.environ.get('APMF_LOOKAHEAD_BATCH', '1') != '0'


def _parallel_map(fn, items, procs=None):
    """[fn(x) for x in items], concurrently when worthwhile; never nested."""
    global _workers
    items = list(items)
    width = min(lookahead_threads(procs), len(items), 16)
    if width <= 1 or getattr(_worker_state, 'inside', False):
        return [fn(x) for x in items]
    if _workers is None:
        # as many threads as the widest map ever asked for (each worker thread keeps one Context:
        # a stream, workspaces, a rocBLAS handle and its block cache - _lib.release_thread_contexts()
        # gives them back)
        _workers = concurrent.futures.ThreadPoolExecutor(max_workers=max(width, 4),
                                                         thread_name_prefix='apmf-lookahead')
    elif _workers._max_workers < width:
        _workers.shutdown(wait=True)
        _workers = concurrent.futures.ThreadPoolExecutor(max_workers=width, thread_name_prefix='apmf-lookahead')

    def run(x):
        _worker_state.inside = True
        try:
            return fn(x)
        finally:
            _worker_state.inside = False

    gate = threading.Semaphore(width)

    def gated(x):
        with gate:
            return run(x)
    return list(_workers.map(gated, items))


# 21-point Gauss-Kronrod abscissae of QUADPACK's dqk21, in the order the routine evaluates its integrand:
# the centre, then +- the 10-point Gauss nodes' neighbours (xgk(2), xgk(4), ...), then +- xgk(1), xgk(3), ...
_XGK21 = (0.995657163025808080735527280689003, 0.973906528517171720077964012084452,
          0.930157491355708226001207180059508, 0.865063366688984510732096688423493,
          0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
          0.562757134668604683339000099272694, 0.433395394129247190799265943165784,
          0.294392862701460198131126603103866, 0.148874338981631210884826001129720)
_quad_patch_lock = threading.Lock()


def _gk21_points(a, b):
    """The 21 points dqk21 evaluates on [a, b], with its arithmetic (centr +- hlgth * xgk)."""
    centr, hlgth = 0.5 * (a + b), 0.5 * (b - a)
    pts = [centr]
    for idx in (1, 3, 5, 7, 9, 0, 2, 4, 6, 8):
        absc = hlgth * _XGK21[idx]
        pts += [centr - absc, centr + absc]
    return pts


class _Quadrature(object):
    """What stats.norm.expect(f, loc=mean, scale=std, lb=left, ub=right, epsrel=.02) - the call of the reference's
    continuous lookahead (mn_active_pmf.py:686-691) - will evaluate f at, known BEFORE f is: the first Gauss-Kronrod
    pass over each of its intervals does not depend on the integrand (found by running that very call on a constant
    integrand: zero error estimate, no interval is bisected - so it is right for whichever SciPy is installed), and
    when the adaptive routine later bisects an interval it evaluates the 21 points of each half, which follow from
    the interval's end points (captured from the integrate.quad calls of the dry run; QUADPACK's dqagse / dqk21
    arithmetic, checked against the dry run's own first pass and switched off if it does not reproduce it)."""

    def __init__(self, mean, std, left, right):
        import scipy.integrate
        seen, bounds = [], []
        me = threading.get_ident()

        def constant(v):
            seen.append(float(v))
            return 1.

        with _quad_patch_lock:
            orig = scipy.integrate.quad

            def spy(func, a, b, *args, **kw):
                if threading.get_ident() == me:
                    bounds.append((float(a), float(b)))
                return orig(func, a, b, *args, **kw)
            scipy.integrate.quad = spy
            try:
                stats.norm.expect(constant, loc=mean, scale=std, lb=left, ub=right, epsrel=.02)
            finally:
                scipy.integrate.quad = orig
        self.first_pass = seen
        predicted = [v for a, b in bounds for v in _gk21_points(a, b)]
        self.intervals = list(bounds) if predicted == seen else None

    def bisection_points(self, v):
        """v was asked for and is not known: if it is the centre of the left half of a tracked interval, that
        interval is being bisected - the 42 points of its two halves; else None."""
        if self.intervals is None:
            return None
        for idx, (a, b) in enumerate(self.intervals):
            mid = 0.5 * (a + b)
            if 0.5 * (a + mid) == v:
                self.intervals[idx:idx + 1] = [(a, mid), (mid, b)]
                return _gk21_points(a, mid) + _gk21_points(mid, b)
        return None


class _Evaluator(object):
    """Picklable `key(ij)` callable (the reference's *Evaluator classes)."""

    def __init__(self, model, key):
        self.model = model
        self.key_name = key.__name__

    def __call__(self, ij):
        fn = getattr(self.model, self.key_name)
        try:
            return fn(ij)
        except Exception:
            import traceback
            traceback.print_exc()
            raise


# criterion metadata (mn_active_pmf.py:90-113)
def do_normal_fit(val):
    def decorator(f):
        f.do_normal_fit = val
        return f
    return decorator


def spawn_processes(val):
    def decorator(f):
        f.spawn_processes = val
        return f
    return decorator


def nice_name(name):
    def decorator(f):
        f.nice_name = name
        return f
    return decorator


def minimize(f):
    f.chooser = min
    return f


def maximize(f):
    f.chooser = max
    return f


def batched(method_name):
    """Marks a criterion whose values for a whole pool come from one GPU launch:
    `method_name(self, pi, pj) -> float64 array`."""
    def decorator(f):
        f.batched = method_name
        return f
    return decorator


def _criterion(name, fit, spawn, chooser=maximize, batch=None, grid=None, lookahead=None):
    def wrapper(f):
        do_normal_fit(fit)(f)
        spawn_processes(spawn)(f)
        chooser(f)
        nice_name(name)(f)
        if batch:
            batched(batch)(f)
        if grid:
            # `grid(self) -> (n, m) float64 array` of the criterion over the whole matrix: the
            # full-pool selections never build the list of candidate tuples
            f.grid = grid
        if lookahead:
            # (fn name, use_map): E[fn(model with R_ij known)], evaluated for a whole pool by one
            # batched device fit when the class supports it (_lookahead_batch)
            f.lookahead = lookahead
        return f
    return wrapper


class _Candidates(object):
    """The `unrated` set as a mask: host bool (n, m) and device uint8 (n, ld), 1 = candidate.
    Kept in step with add_ratings(); rebuilt if the set was replaced or edited behind our back."""
    __slots__ = ('host', 'dev', 'count', 'set_id', 'ld')

    def __init__(self, model, ctx):
        n, m = model.num_users, model.num_items
        rec = model.__dict__.get('_pairs_recipe')
        if rec is not None:
            # model built from dense arrays (from_dense): the mask is already there
            self.host = rec.cand_mask.copy()
            self.ld = _lib.grid_ld(m)
            self.dev = ctx.upload_grid(self.host.view(np.uint8), np.uint8, ld=self.ld)
            self.count = rec.n_unrated
            self.set_id = None
            return
        unrated, rated = model.unrated, model.rated
        host = None
        if len(unrated) + len(rated) == n * m:
            # unrated = knowable - rated is disjoint from rated, so with these sizes it is the
            # complement of rated: no need to walk the 10^6-tuple set
            host = np.ones((n, m), bool)
            if rated:
                r = np.array(list(rated), dtype=np.int64).reshape(-1, 2)
                host[r[:, 0], r[:, 1]] = False
        if host is None:
            host = np.zeros((n, m), bool)
            if unrated:
                u = np.fromiter(itertools.chain.from_iterable(unrated), dtype=np.int64,
                                count=2 * len(unrated)).reshape(-1, 2)
                host[u[:, 0], u[:, 1]] = True
        self.host = host
        self.ld = _lib.grid_ld(m)
        self.dev = ctx.upload_grid(host.view(np.uint8), np.uint8, ld=self.ld)
        self.count = len(unrated)
        self.set_id = id(unrated)

    def remove(self, ctx, pairs):
        zero = np.zeros(1, np.uint8)
        for i, j in pairs:
            if self.host[i, j]:
                self.host[i, j] = False
                ctx.write_bytes(self.dev, i * self.ld + j, zero)


def _pool_arrays(pool):
    pool = pool if isinstance(pool, list) else list(pool)
    if not pool:
        return pool, np.zeros(0, np.int32), np.zeros(0, np.int32)
    try:        # 1.6x np.asarray on a list of 10^6 tuples; anything that is not flat pairs of ints goes the slow way
        arr = np.fromiter(itertools.chain.from_iterable(pool), dtype=np.int64, count=2 * len(pool)).reshape(-1, 2)
    except (TypeError, ValueError):
        arr = np.asarray(pool)
    return pool, np.ascontiguousarray(arr[:, 0], np.int32), np.ascontiguousarray(arr[:, 1], np.int32)


# ---------------------------------------------------------------------------- the base class
class _ActiveBase(ProbabilisticMatrixFactorization):
    """Active-learning surface of the reference's (MN)ActivePMF, posterior-agnostic."""

    def __init__(self, rating_tuples, latent_d=1, rating_values=None,
                 discrete_expectations=False, refit_lookahead=False, knowable=None,
                 fit_type=('batch',)):
        super(_ActiveBase, self).__init__(rating_tuples, latent_d=latent_d, subtract_mean=False,
                                          knowable=knowable, fit_type=fit_type)
        self.ratings = np.asarray(self.ratings, dtype=float)
        if rating_values is not None:
            rating_values = set(map(float, rating_values))
            if not rating_values.issuperset(self.ratings[:, 2]):
                raise ValueError("got ratings not in rating_values")
        self.rating_values = rating_values
        self.discrete_expectations = discrete_expectations
        self.refit_lookahead = refit_lookahead
        self.approx_dim = (self.num_users + self.num_items) * self.latent_d
        self.normal_learning_rate = 1e-4
        self.min_eig = 1e-5

    # -- copy / pickle: subclass attributes travel in state['__dict__'] like the reference
    def __copy__(self):
        res = type(self)(self.ratings, self.latent_d, self.rating_values,
                         self.discrete_expectations)
        res.__setstate__(self.__getstate__())
        return res

    def __deepcopy__(self, memodict):
        rec = self._recipe_clone()
        res = type(self)(self.ratings, self.latent_d, self.rating_values,
                         self.discrete_expectations, **({} if rec is None else {'knowable': rec}))
        res.__setstate__(deepcopy(self.__getstate__(_pairs=rec is None), memodict))
        return res

    def __getstate__(self, _pairs=True):
        state = super(_ActiveBase, self).__getstate__(_pairs)
        state['__dict__'] = self._extra_dict()
        state['__dict__'].pop('lookahead_comm', None)       # a communicator is not model state
        return state

    rating_values = property(lambda self: self._rating_values)
    rating_bounds = property(lambda self: self._rating_bounds)

    @rating_values.setter
    def rating_values(self, vals):
        if vals:
            vals = tuple(sorted(vals))
            self._rating_values = vals
            edges = np.empty(len(vals) + 2)
            edges[0], edges[1:-1], edges[-1] = -np.inf, vals, np.inf
            self._rating_bounds = (edges[1:] + edges[:-1]) / 2
        else:
            self._rating_values = None
            self._rating_bounds = None

    # -- candidate mask ---------------------------------------------------------------------
    def add_ratings(self, extra):
        super(_ActiveBase, self).add_ratings(extra)
        dev = self.__dict__.get('_dev')
        if dev is not None and dev.cand is not None and dev.ctx is not None:
            extra = np.array(extra, ndmin=2)
            dev.cand.remove(dev.ctx, [(int(i), int(j)) for i, j in extra[:, :2]])
            dev.cand.count = self._n_unrated()

    def _is_rated(self, i, j):
        rec = self.__dict__.get('_pairs_recipe')
        if rec is not None:
            return bool(rec.rated_mask[i, j])
        return (i, j) in self.rated

    def _candidates(self):
        dev = self._device()
        c = dev.cand
        lazy = self.__dict__.get('_pairs_recipe') is not None
        stale = c is None or c.count != self._n_unrated()
        if not stale and not lazy:
            if c.set_id is None:
                c.set_id = id(self.unrated)      # built from the masks; the sets exist by now
            elif c.set_id != id(self.unrated):
                stale = True
        if stale:
            c = dev.cand = _Candidates(self, dev.ctx)
        return c

    def _pick_from_grid(self, vals, chooser):
        """chooser(self.unrated, key=vals.__getitem__) without the Python loop over the set:
        the first extremum in the set's iteration order (mn_active_pmf.py:886)."""
        cand = self._candidates().host
        masked = np.where(cand, vals, -np.inf if chooser is max else np.inf)
        if np.isnan(masked).any():
            return chooser(self.unrated, key=vals.__getitem__)
        best = masked.max() if chooser is max else masked.min()
        ti, tj = np.nonzero((masked == best) & cand)
        if ti.size == 0:
            return chooser(self.unrated, key=vals.__getitem__)
        return self._first_in_pool_order(ti, tj)

    def _first_in_pool_order(self, ti, tj):
        if ti.size == 1:
            return (int(ti[0]), int(tj[0]))
        ties = set(zip(ti.tolist(), tj.tolist()))
        for ij in self.unrated:          # the reference's tie rule: set iteration order
            if ij in ties:
                return ij
        raise RuntimeError("candidate mask out of step with the unrated set")

    # -- supplied by the subclass ---------------------------------------------------------
    def _score_pool(self, pi, pj):
        """(mean, var) float64 arrays of E / Var[U_i.V_j] for the pool, from the GPU."""
        raise NotImplementedError

    def fit_normal_kls(self):
        raise NotImplementedError

    def fit_normal(self):
        '''Fit the approximation by projected gradient descent on the KL.'''
        for kl in self.fit_normal_kls():
            pass

    def _descend(self, current, gradient, step, project, objective):
        """The reference's learning-rate search for the variational fit
        (mn_active_pmf.py:242-288 / active_pmf.py:250-288): x1.25 on an improving step,
        x0.5 otherwise, converged when the KL improves by < .005 or lr < 1e-10."""
        lr = self.normal_learning_rate
        old_kl = objective(*current())
        converged = False
        while not converged:
            grads = gradient()
            while True:
                cand = tuple(project(k, x - lr * g)
                             for k, (x, g) in enumerate(zip(current(), grads)))
                new_kl = objective(*cand)
                if new_kl < old_kl:
                    step(cand)
                    lr *= 1.25
                    if old_kl - new_kl < .005:
                        converged = True
                    yield new_kl
                    old_kl = new_kl
                    break
                else:
                    lr *= .5
                    if lr < 1e-10:
                        converged = True
                        break

    # -- one-candidate views -----------------------------------------------------------------
    def approx_pred_mean_var(self, i, j):
        mean, var = self._score_pool(np.array([i], np.int32), np.array([j], np.int32))
        return float(mean[0]), float(var[0])

    # -- criteria (mn_active_pmf.py:404-625) ------------------------------------------------
    @_criterion("Random", False, False)
    def random_weighting(self, ij):
        return random.random()

    def _pred_pool(self, pi, pj):
        return self.predicted_matrix()[pi, pj]

    def _pred_grid(self):
        return self.predicted_matrix()

    @_criterion("Pred Mag", False, False, batch='_pred_pool', grid='_pred_grid')
    def pred(self, ij):
        '''The MAP estimate of R_ij.'''
        i, j = ij
        return np.dot(self.users[i, :], self.items[j, :])

    def _prob_ge_pool(self, pi, pj, cutoff):
        mean, var = self._score_pool(pi, pj)
        # sic: the variance is passed as `scale` (mn_active_pmf.py:438)
        return stats.norm.sf(cutoff, loc=mean, scale=var)

    def _prob_ge_cutoff(self, ij, cutoff):
        mean, var = self.approx_pred_mean_var(*ij)
        return stats.norm.sf(cutoff, loc=mean, scale=var)

    def _prob_ge_3_5_pool(self, pi, pj):
        return self._prob_ge_pool(pi, pj, 3.5)

    def _prob_ge_half_pool(self, pi, pj):
        return self._prob_ge_pool(pi, pj, .5)

    def _prob_ge_grid(self, cutoff):
        mean, var = self.approx_pred_means_vars()
        return stats.norm.sf(cutoff, loc=mean, scale=var)      # sic, as in _prob_ge_pool

    def _prob_ge_3_5_grid(self):
        return self._prob_ge_grid(3.5)

    def _prob_ge_half_grid(self):
        return self._prob_ge_grid(.5)

    @_criterion("Prob >= 3.5", True, False, batch='_prob_ge_3_5_pool', grid='_prob_ge_3_5_grid')
    def prob_ge_3_5(self, ij):
        return self._prob_ge_cutoff(ij, 3.5)

    @_criterion("Prob >= .5", True, False, batch='_prob_ge_half_pool', grid='_prob_ge_half_grid')
    def prob_ge_half(self, ij):
        return self._prob_ge_cutoff(ij, .5)

    def _pred_variance_pool(self, pi, pj):
        return self._score_pool(pi, pj)[1]

    def _pred_variance_grid(self):
        return self.approx_pred_means_vars()[1]

    @_criterion("Pred Variance", True, False, batch='_pred_variance_pool', grid='_pred_variance_grid')
    def pred_variance(self, ij):
        '''Variance of the prediction for R_ij under the current approximation.'''
        return self.approx_pred_mean_var(*ij)[1]

    def _onestep_ge_cutoff(self, ij, cutoff, use_map):
        '''One-step lookahead utility (Garnett et al., KDD-MLG 2011) for the class
        "value >= cutoff".'''
        def fn(apmf, v):
            return apmf._last_step_lookahead_helper(cutoff=cutoff, v=v)
        fn.__name__ = '_1step_{}'.format(cutoff)
        return self._exp_with_rij(ij, fn, use_map=use_map, discretize=True, pass_v=True)

    @_criterion("1 step >= 3.5 (MAP)", True, True, lookahead=(('onestep', 3.5), True))
    def onestep_ge_3_5(self, ij):
        return self._onestep_ge_cutoff(ij, 3.5, True)

    @_criterion("1 step >= 3.5 (Approx)", True, True, lookahead=(('onestep', 3.5), False))
    def onestep_ge_3_5_approx(self, ij):
        return self._onestep_ge_cutoff(ij, 3.5, False)

    @_criterion("1 step >= .5 (MAP)", True, True, lookahead=(('onestep', .5), True))
    def onestep_ge_half(self, ij):
        return self._onestep_ge_cutoff(ij, .5, True)

    @_criterion("1 step >= .5 (Approx)", True, True, lookahead=(('onestep', .5), False))
    def onestep_ge_half_approx(self, ij):
        return self._onestep_ge_cutoff(ij, .5, False)

    def _last_step_lookahead_helper(self, cutoff, v):
        utility = int(v >= cutoff)
        pool, pi, pj = _pool_arrays(self.unrated)
        return utility + max(self._prob_ge_pool(pi, pj, cutoff))

    def _total_variance(self):
        return self.approx_pred_means_vars()[1].sum()

    @_criterion("E[U/V Entropy] (MAP)", True, True, chooser=minimize, lookahead=('_approx_entropy', True))
    def exp_approx_entropy(self, ij):
        return self._exp_with_rij(ij, type(self)._approx_entropy, use_map=True)

    @_criterion("E[U/V Entropy] (Approx)", True, True, chooser=minimize, lookahead=('_approx_entropy', False))
    def exp_approx_entropy_byapprox(self, ij):
        return self._exp_with_rij(ij, type(self)._approx_entropy, use_map=False)

    @_criterion("E[Pred Total Variance] (MAP)", True, True, chooser=minimize, lookahead=('_total_variance', True))
    def exp_total_variance(self, ij):
        return self._exp_with_rij(ij, type(self)._total_variance, use_map=True)

    @_criterion("E[Pred Total Variance] (Approx)", True, True, chooser=minimize, lookahead=('_total_variance', False))
    def exp_total_variance_byapprox(self, ij):
        return self._exp_with_rij(ij, type(self)._total_variance, use_map=False)

    # -- one-step lookahead (mn_active_pmf.py:627-696) -------------------------------------------
    quadrature_limit = 50             # subdivisions of the continuous lookahead's quadrature (scipy's default)
    _approx_init_uses_rng = False     # initialize_approx() draws from np.random (ActivePMF)

    def _lookahead_batch_ok(self):
        """Whether this model's hypotheses can go through the class' device batch (_lookahead_batch)."""
        return True

    def _hypothesis_model(self, i, j, v, keep_rng=False):
        """deepcopy(self) with R_ij = v added: the model a hypothesis is evaluated on.  The copy's constructor
        draws n d + m d uniforms from np.random (pmf_cy.pyx:74-75, overwritten by the copied state); keep_rng
        puts the generator back where it was - the batched paths account for those draws in ONE place
        (_hypothesis_values), whatever they build afterwards."""
        state = np.random.get_state() if keep_rng else None
        apmf = deepcopy(self)
        if state is not None:
            np.random.set_state(state)
        apmf.add_rating(i, j, v)
        return apmf

    def _hypothesis_values(self, hyps, fn, pass_v, procs=None):
        """fn(model refitted with R_ij = v) for every hypothesis (i, j, v) of `hyps`, in order
        (the reference's calculate_fn, mn_active_pmf.py:660-672).  The matrix-normal class fits
        all of them in one batch on the device (_lookahead_batch); otherwise every hypothesis is a
        deep copy re-fitted on its own, a few at a time on worker threads."""
        batch = getattr(self, '_lookahead_batch', None)
        if batch is not None and len(hyps) > 0 and lookahead_batched() and self._lookahead_batch_ok():
            # what calculate_fn's deepcopy + add_rating would have done besides the arithmetic: refuse a
            # hypothesis about an entry that is already rated (pmf_cy.pyx:143-150), and - when the reference
            # evaluates in-process (procs == 1; its default hands the pool to forked workers, whose draws the
            # parent never sees) - consume the n d + m d uniform draws of every copy's constructor
            # (pmf_cy.pyx:74-75), so that a seeded run continues with the same random stream.  This is the ONLY
            # place the batched path touches the generator: model objects it builds later copy with keep_rng
            for i, j, _ in hyps:
                if self._is_rated(i, j):
                    raise ValueError("can't rate already rated items")
            if procs == 1:
                np.random.random(len(hyps) * (self.num_users + self.num_items) * self.latent_d)
            return batch(hyps, fn, pass_v)
        # the copies are made here, in order: their constructors draw from the global RNG
        models = [self._hypothesis_model(i, j, v) for i, j, v in hyps]

        def evaluate(args):
            apmf, (i, j, v) = args
            if apmf.refit_lookahead:
                apmf.do_fit()
                apmf.initialize_approx()
            apmf.fit_normal()
            return fn(apmf, v=v) if pass_v else fn(apmf)
        if self.refit_lookahead and self._approx_init_uses_rng:
            procs = 1        # initialize_approx() draws from np.random: keep the reference's order
        return _parallel_map(evaluate, list(zip(models, hyps)), procs)

    def _exp_with_rij_pool(self, pool, fn, use_map=True, discretize=None, pass_v=False, procs=None):
        '''[E[fn(model with R_ij known)] for ij in pool] over the predictive distribution of R_ij
        (mn_active_pmf.py:627-696): MAP mean with variance sigma_sq if use_map, else the
        approximation's mean/variance; summed over rating_values when discretising, else
        integrated to +-2 sigma with scipy.  All hypotheses of the pool are evaluated together.'''
        pool = [(int(i), int(j)) for i, j in pool]
        if discretize is None:
            discretize = self.discrete_expectations
        if use_map:
            means = [float(np.dot(self.users[i, :], self.items[j, :])) for i, j in pool]
            variances = [self.sigma_sq] * len(pool)
        else:
            _, pi, pj = _pool_arrays(pool)
            pm, pv = self._score_pool(pi, pj)
            means, variances = [float(x) for x in pm], [float(x) for x in pv]
        name = getattr(fn, '__name__', '')
        points = self.rating_values
        out = []
        if discretize and points:
            hyps = [(i, j, v) for i, j in pool for v in points]
            flat = np.asarray(self._hypothesis_values(hyps, fn, pass_v, procs), dtype=float)
            evals_all = flat.reshape(len(pool), len(points))
            for (i, j), mean, var, evals in zip(pool, means, variances, evals_all):
                std = math.sqrt(var)
                if discretize == 'simps':
                    pdfs = stats.norm.pdf(points, loc=mean, scale=std)
                    import scipy.integrate
                    simpson = getattr(scipy.integrate, 'simpson', None) or scipy.integrate.simps
                    est = simpson(evals * pdfs, x=points)
                    s = "simps'ed"
                else:
                    cdfs = stats.norm.cdf(self.rating_bounds, loc=mean, scale=std)
                    est = (evals * np.diff(cdfs)).sum()
                    s = "summed"
                print("\t{:>20}({},{}) {}: {: 10.2f}".format(name, i, j, s, est))
                out.append(est)
            return out
        if discretize and points is None:
            warnings.warn("{} has no rating_values; doing integral".format(type(self).__name__))
        # Continuous integration: scipy's adaptive quadrature (QUADPACK, 21-point Gauss-Kronrod per interval) asks
        # for one rating value at a time - but the points of its FIRST pass over every interval do not depend on the
        # integrand: a dry run with a constant integrand lists them, the hypotheses (candidate, point) of the whole
        # pool are then evaluated as ONE batch, and the real quadrature is served from that cache.  A point it asks
        # for beyond the first pass (an interval it decides to bisect) is evaluated on its own, as before.  The
        # hypotheses are independent, so every value - and the integral - is the one-at-a-time result bit for bit.
        spans = []
        for mean, var in zip(means, variances):
            std = math.sqrt(var)
            spans.append((mean, std, mean - 2 * std, mean + 2 * std))
        caches = [dict() for _ in pool]
        plans = [None] * len(pool)
        if lookahead_batched() and len(pool) > 0:
            hyps, owner = [], []
            for c, ((i, j), span) in enumerate(zip(pool, spans)):
                plans[c] = _Quadrature(*span)
                for v in plans[c].first_pass:
                    if v not in caches[c]:
                        caches[c][v] = None
                        hyps.append((i, j, v))
                        owner.append(c)
            if hyps:
                for (i, j, v), c, val in zip(hyps, owner, self._hypothesis_values(hyps, fn, pass_v, procs)):
                    caches[c][v] = val
        for (i, j), (mean, std, left, right), cache, plan in zip(pool, spans, caches, plans):
            def integrand(v, i=i, j=j, cache=cache, plan=plan):
                v = float(v)
                val = cache.get(v)
                if val is None:
                    more = plan.bisection_points(v) if plan is not None else None
                    if more and v in more:
                        # the interval being bisected: the 21 points of each half as one batch
                        new = [x for x in dict.fromkeys(more) if cache.get(x) is None]
                        for x, y in zip(new, self._hypothesis_values([(i, j, x) for x in new], fn, pass_v, procs)):
                            cache[x] = y
                        val = cache[v]
                    else:
                        val = cache[v] = self._hypothesis_values([(i, j, v)], fn, pass_v, procs)[0]
                return val
            est = stats.norm.expect(integrand, loc=mean, scale=std, lb=left, ub=right, epsrel=.02,
                                    limit=self.quadrature_limit)
            s = "from {:5.1f} to {:5.1f}".format(left, right)
            print("\t{:>20}({},{}) {}: {: 10.2f}".format(name, i, j, s, est))
            out.append(est)
        return out

    def _exp_with_rij(self, ij, fn, use_map=True, discretize=None, pass_v=False):
        return self._exp_with_rij_pool([ij], fn, use_map, discretize, pass_v)[0]

    def _lookahead_spec(self, key):
        """(fn, use_map, discretize, pass_v) of a lookahead criterion, or None."""
        spec = getattr(key, 'lookahead', None)
        if spec is None:
            return None
        what, use_map = spec
        if isinstance(what, str):
            return getattr(type(self), what), use_map, None, False
        cutoff = what[1]

        def fn(apmf, v):
            return apmf._last_step_lookahead_helper(cutoff=cutoff, v=v)
        fn.__name__ = '_1step_{}'.format(cutoff)
        return fn, use_map, True, True

    # -- selection surface (mn_active_pmf.py:701-779) ----------------------------------------
    def _default_key(self):
        return type(self).pred_variance

    def pick_query_point(self, pool=None, key=None, procs=None, worker_pool=None):
        '''Next (i, j) to query: the first extremum of `key` over `pool` (default
        self.unrated) in the pool's iteration order, max or min per key.chooser.'''
        lazy = self.__dict__.get('_pairs_recipe') is not None     # sets not built yet (from_dense)
        full = pool is None or (not lazy and pool is self.unrated)
        if key is None:
            key = self._default_key()
        chooser = getattr(key, 'chooser', max)
        n_pool = self._n_unrated() if full else len(pool)
        if n_pool == 0:
            raise ValueError("can't pick a query point from an empty pool")
        elif n_pool == 1:
            return next(iter(self.unrated if full else pool))
        if full and chooser in (max, min):
            # whole unrated set: the device holds it as a mask; no list of 10^6 tuples is built
            fast = getattr(self, '_pick_unrated_device', None)
            got = fast(key, chooser) if fast is not None else None
            if got is not None:
                return got
            grid = getattr(key, 'grid', None)
            if grid is not None:
                return self._pick_from_grid(np.asarray(getattr(self, grid)(), dtype=np.float64), chooser)
        pool = list(self.unrated if pool is None else pool)
        impl = getattr(key, 'batched', None)
        if impl is not None and chooser in (max, min):
            # one launch scores the pool; the first extremum in pool order (what max() / min() over
            # zip(pool, vals) return) is np.argmax / np.argmin's first occurrence
            pool, pi, pj = _pool_arrays(pool)
            vals = np.asarray(getattr(self, impl)(pi, pj), dtype=np.float64)
            if not np.isnan(vals).any():
                return pool[int(np.argmax(vals) if chooser is max else np.argmin(vals))]
            return chooser(zip(pool, vals.tolist()), key=operator.itemgetter(1))[0]
        vals = self._get_key_vals(pool, key, procs, worker_pool)
        return chooser(zip(pool, vals), key=operator.itemgetter(1))[0]

    def _get_key_vals(self, pool, key, procs=None, worker_pool=None):
        '''Values of `key` for every (i, j) of `pool`, in pool order.  Batched criteria are
        one GPU launch; lookahead criteria evaluate all hypotheses of the pool together; the
        rest (random) is a plain map.'''
        impl = getattr(key, 'batched', None)
        if impl is not None:
            pool, pi, pj = _pool_arrays(pool)
            if not pool:
                return []
            return list(np.asarray(getattr(self, impl)(pi, pj), dtype=np.float64))
        pool = list(pool)
        spec = self._lookahead_spec(key)
        if spec is not None and pool:
            fn, use_map, discretize, pass_v = spec
            return self._exp_with_rij_pool(pool, fn, use_map, discretize, pass_v, procs)
        return [key(self, ij) for ij in pool]

    def get_key_evals(self, pool=None, key=None, procs=None, worker_pool=None):
        '''(n, m) matrix holding key's value at every element of pool, NaN elsewhere.'''
        lazy = self.__dict__.get('_pairs_recipe') is not None
        full = pool is None or (not lazy and pool is self.unrated)
        if key is None:
            key = self._default_key()
        grid = getattr(key, 'grid', None)
        if full and grid is not None and self._n_unrated() > 0:
            vals = np.asarray(getattr(self, grid)(), dtype=np.float64)
            return np.where(self._candidates().host, vals, np.nan)
        pool = list(self.unrated if pool is None else pool)
        evals = np.empty((self.num_users, self.num_items))
        evals.fill(np.nan)
        if pool:
            evals[tuple(zip(*pool))] = self._get_key_vals(pool, key, procs, worker_pool)
        return evals


# ---------------------------------------------------------------------------- experiment loop
def make_key_funcs(cls, with_entropy_bound=False):
    keys = {
        "random": cls.random_weighting,
        "pred-variance": cls.pred_variance,
        "total-variance": cls.exp_total_variance,
        "total-variance-approx": cls.exp_total_variance_byapprox,
        "uv-entropy": cls.exp_approx_entropy,
        "uv-entropy-approx": cls.exp_approx_entropy_byapprox,
        "pred": cls.pred,
        "prob-ge-3.5": cls.prob_ge_3_5,
        "prob-ge-.5": cls.prob_ge_half,
        "1step-ge-3.5": cls.onestep_ge_3_5,
        "1step-ge-3.5-approx": cls.onestep_ge_3_5_approx,
        "1step-ge-.5": cls.onestep_ge_half,
        "1step-ge-.5-approx": cls.onestep_ge_half_approx,
    }
    if with_entropy_bound:
        keys["pred-entropy-bound"] = cls.exp_pred_entropy_bound
        keys["pred-entropy-bound-approx"] = cls.exp_pred_entropy_bound_byapprox
    return keys


def query_loop(apmf, real, picker_key, fit_normal, fit_sigmas, test_on=None, with_pred=True,
               matrix_vals=False, initial_fit=False, refit_with_lls=False):
    """select -> query -> refit generator behind full_test / _full_test_threaded
    (mn_active_pmf.py:795-894): yields (n_rated, rmse, (i,j), vals[, predicted_matrix])."""
    name = picker_key.nice_name
    total = real.size

    def result(*parts):
        return parts + ((apmf.predicted_matrix(),) if with_pred and parts[2] is not None
                        else ((None,) if with_pred else ()))

    if initial_fit:
        if fit_sigmas:
            apmf.fit_with_sigmas()
        else:
            apmf.do_fit()
        apmf.initialize_approx()
        if fit_normal:
            apmf.fit_normal()
    rmse = apmf.rmse(real, test_on) if test_on is not None else apmf.rmse(real)
    print("{:<40} Initial RMSE: {:.5}".format(name, rmse))
    yield result(apmf._n_rated(), rmse, None, None)

    while apmf._n_unrated():
        n = apmf._n_rated() + 1
        print("{:<40} Picking query point {}...".format(name, n))
        if apmf._n_unrated() == 1:
            i, j = next(iter(apmf.unrated))
            if matrix_vals:
                vals = np.empty((apmf.num_users, apmf.num_items))
                vals.fill(np.nan)
            else:
                vals = None
        elif matrix_vals:
            vals = apmf.get_key_evals(key=picker_key)
            if picker_key.chooser in (max, min):
                i, j = apmf._pick_from_grid(vals, picker_key.chooser)
            else:
                i, j = picker_key.chooser(apmf.unrated, key=vals.__getitem__)
        else:
            pool = list(apmf.unrated)
            vals = apmf._get_key_vals(pool, picker_key, None, None)
            i, j = picker_key.chooser(zip(pool, vals), key=operator.itemgetter(1))[0]

        apmf.add_rating(i, j, real[i, j])
        print("{:<40} Queried ({}, {}); {}/{} known".format(name, i, j, apmf._n_rated(), total))
        if refit_with_lls:
            # full_test refits with plain fit_lls/fit_normal_kls, never re-initialising
            for ll in apmf.fit_lls():
                pass
            if fit_normal:
                for kl in apmf.fit_normal_kls():
                    assert kl > -1e5
        else:
            if fit_sigmas:
                apmf.fit_with_sigmas()
            else:
                apmf.do_fit()
            if fit_normal:
                if apmf.refit_lookahead:
                    apmf.initialize_approx()
                apmf.fit_normal()
        rmse = apmf.rmse(real, test_on) if test_on is not None else apmf.rmse(real)
        print("{:<40} RMSE {}: {:.5}".format(name, n, rmse))
        yield result(apmf._n_rated(), rmse, (i, j), vals)


def run_compare(cls, key_funcs, key_names, real, ratings, rating_vals, latent_d, knowable,
                test_on, steps, discrete_exp, refit_lookahead, fit_sigmas, apmf, sig_priors,
                fit_type, do_threading, with_pred):
    """Body of compare() (mn_active_pmf.py:922-1003 / active_pmf.py:1003-1092)."""
    if apmf:
        assert (apmf.num_users, apmf.num_items) == real.shape
        assert np.all(apmf.ratings == ratings)
        assert set(apmf.rating_values) == set(rating_vals)
        apmf.discrete_expectations = discrete_exp
    else:
        kw = dict(latent_d=latent_d, rating_values=rating_vals, discrete_expectations=discrete_exp,
                  refit_lookahead=refit_lookahead, fit_type=fit_type)
        apmf = None
        if isinstance(knowable, np.ndarray) and knowable.dtype == bool and knowable.shape == np.shape(real):
            # the CLI hands the knowable MASK over: when `ratings` is what make_ratings() writes for
            # this matrix (row-major order of the known mask, choose_training.py:13-17) the model is
            # built from the dense arrays - device container, no 10^6-tuple sets up front
            ratings = np.asarray(ratings, dtype=float)
            ii, jj = ratings[:, 0].astype(np.int64), ratings[:, 1].astype(np.int64)
            known = np.zeros(np.shape(real), bool)
            known[ii, jj] = True
            ki, kj = known.nonzero()
            if ki.size == ii.size and np.array_equal(ki, ii) and np.array_equal(kj, jj) and \
                    np.array_equal(np.asarray(real, float)[ii, jj], ratings[:, 2]):
                apmf = cls.from_dense(real, known, knowable=knowable, **kw)
            else:
                knowable = zip(*knowable.nonzero())
        if apmf is None:
            apmf = cls(ratings, knowable=knowable, **kw)
        apmf.sig_u_mean, apmf.sig_u_var, apmf.sig_v_mean, apmf.sig_v_var = sig_priors
        print("Doing initial fit")
        if fit_sigmas:
            apmf.fit_with_sigmas()
        else:
            apmf.do_fit()
        if any(key_funcs[name].do_normal_fit for name in key_names):
            apmf.initialize_approx()
            print("Initial approximation fit")
            apmf.fit_normal()

    results = {
        '_real': real,
        '_ratings': ratings,
        '_rating_vals': rating_vals,
        '_initial_apmf': deepcopy(apmf),
    }
    # The reference runs one thread per key (sharing a forked process pool for the lookahead
    # keys) when do_threading is set, and the keys one after the other otherwise
    # (mn_active_pmf.py:968-1001).  Same here: with do_threading each key's query loop runs on
    # its own thread = its own Context / stream, up to APMF_LOOKAHEAD_THREADS at a time, and the
    # lookahead map inside a key then stays sequential (the streams are already taken).  As in
    # the reference, keys that draw from the global RNGs ('random') are not reproducible run to
    # run in that mode.
    def run_key(key_name):
        key = key_funcs[key_name]
        if do_threading:
            res = query_loop(deepcopy(apmf), real, key, key.do_normal_fit, fit_sigmas,
                             test_on=test_on, with_pred=with_pred, matrix_vals=True)
        else:
            res = query_loop(deepcopy(apmf), real, key, key.do_normal_fit, fit_sigmas,
                             test_on=test_on, with_pred=with_pred, initial_fit=True,
                             refit_with_lls=True)
        return list(itertools.islice(res, steps))

    if do_threading and len(key_names) > 1 and lookahead_threads() > 1:
        for key_name, res in zip(key_names, _parallel_map(run_key, key_names)):
            results[key_name] = res
    else:
        for key_name in key_names:
            results[key_name] = run_key(key_name)
    return results
