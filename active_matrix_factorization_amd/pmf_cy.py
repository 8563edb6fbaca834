"""Drop-in for the reference's ``pmf_cy`` extension module (python-pmf/pmf_cy.pyx, .pxd).

Same class name, constructor, public attributes, method names and error behaviour as the
reference's Cython ``cdef class ProbabilisticMatrixFactorization`` - so pickles, subclasses
and the select -> query -> refit loops that import ``pmf_cy`` keep working - but every
per-rating loop (gradient, log_likelihood, update_sigma, fit_lls, fit_minibatches,
predicted_matrix) runs as HIP kernels on an MI355X through libapmf_hip.so.  State stays in
host NumPy float64 arrays exactly as in the reference; device buffers are a cache that is
never pickled or deep-copied.  There is no CPU fallback.
"""
import itertools
import random
import warnings
from copy import deepcopy

import numpy as np

try:
    from . import _lib
except ImportError:      # imported as a top-level module (package dir on sys.path)
    import _lib

DTYPE = float


def rmse(exp, obs):
    # `cpdef float rmse` in the reference returns a C float: the value is rounded to
    # single precision (pmf_cy.pyx:25-26)
    return float(np.float32(np.sqrt(np.mean((obs - exp) ** 2))))


def rmse_on(exp, obs, on):
    return float(np.float32(np.sqrt(np.mean((obs[on] - exp[on]) ** 2))))  # pmf_cy.pyx:28-29


class _PairsRecipe(object):
    """How to build the `rated` / `unrated` sets of a model that was created from dense arrays
    (from_dense): the construction the reference performs (pmf_cy.pyx:69-72) and every
    add_ratings() since, replayed on first access - the sets' iteration order, which breaks ties
    in the selection loop, comes out as if they had been built eagerly.  Until then the masks
    below answer everything the accelerated paths ask (counts, candidate mask, membership)."""
    __slots__ = ('pairs0', 'knowable', 'adds', 'rated_mask', 'cand_mask', 'n_rated', 'n_unrated')

    def __init__(self, pairs0, knowable, n, m):
        self.pairs0 = pairs0                  # (N0, 2) float64: ratings[:, :2] at construction
        self.knowable = knowable              # None (everything) or a bool (n, m) mask
        self.adds = []
        self.rated_mask = np.zeros((n, m), bool)
        self.rated_mask[pairs0[:, 0].astype(np.int64), pairs0[:, 1].astype(np.int64)] = True
        self.cand_mask = ~self.rated_mask if knowable is None else (knowable & ~self.rated_mask)
        self.n_rated = int(self.rated_mask.sum())
        self.n_unrated = int(self.cand_mask.sum())

    def clone(self):
        """Independent copy (the construction inputs are never modified and stay shared)."""
        c = _PairsRecipe.__new__(_PairsRecipe)
        c.pairs0, c.knowable = self.pairs0, self.knowable
        c.adds = [set(a) for a in self.adds]
        c.rated_mask, c.cand_mask = self.rated_mask.copy(), self.cand_mask.copy()
        c.n_rated, c.n_unrated = self.n_rated, self.n_unrated
        return c

    def build(self, n, m):
        rated = set(map(tuple, self.pairs0))
        if self.knowable is None:
            knowable = itertools.product(range(n), range(m))
        else:
            knowable = zip(*self.knowable.nonzero())       # what the reference's CLIs pass
        unrated = set(knowable).difference(rated)
        for new_items in self.adds:
            rated.update(new_items)
            unrated.difference_update(new_items)
        return rated, unrated


class _DeviceCache(object):
    """Per-object device state: never part of __getstate__, rebuilt lazily after copy/unpickle."""
    __slots__ = ('ctx', 'ratings', 'ratings_key', 'resident', 'cand')

    def __init__(self):
        self.ctx = None
        self.ratings = None
        self.ratings_key = None
        self.resident = None     # device-resident posterior approximation (MNActivePMF)
        self.cand = None         # candidate ("unrated") mask, host + device (_active_base._Candidates)


_PUBLIC_STATE = (
    'latent_d', 'num_users', 'num_items',
    'learning_rate', 'min_learning_rate', 'stop_thresh',
    'sigma_sq', 'sigma_u_sq', 'sigma_v_sq', 'fit_type',
    'sig_u_mean', 'sig_u_var', 'sig_v_mean', 'sig_v_var',
    'ratings', 'users', 'items',
    'subtract_mean', 'mean_rating', 'rated', 'unrated',
)


class ProbabilisticMatrixFactorization(object):
    """PMF MAP model (Salakhutdinov & Mnih 2007); API of pmf_cy.pyx:34-441."""

    # the reference is a cdef class: instances of the base class have no __dict__, Python
    # subclasses do.  __getstate__ below reproduces that split.
    _base_attrs = frozenset(_PUBLIC_STATE)

    def __init__(self, rating_tuples, latent_d=1, subtract_mean=False, knowable=None,
                 fit_type=('batch',)):
        if rating_tuples is None:
            raise TypeError("Argument 'rating_tuples' must not be None")
        # defaults of __cinit__ (pmf_cy.pyx:35-47)
        self.learning_rate = 1e-4
        self.min_learning_rate = 1e-10
        self.stop_thresh = 1e-2
        self.sigma_sq = 1.
        self.sigma_u_sq = 10.
        self.sigma_v_sq = 10.
        self.sig_u_mean = self.sig_v_mean = 0.
        self.sig_u_var = self.sig_v_var = -1.

        self.latent_d = int(latent_d)
        self.subtract_mean = bool(subtract_mean)
        if fit_type is None:
            warnings.warn('passed None fit_type; using batch')
            fit_type = ('batch',)
        self.fit_type = tuple(fit_type)

        self.ratings = np.asarray(rating_tuples, dtype=float)
        if self.ratings.ndim != 2 or self.ratings.shape[1] != 3:
            raise TypeError("invalid rating tuple length")
        self.mean_rating = float(np.mean(self.ratings[:, 2]))

        self.num_users = n = int(np.max(self.ratings[:, 0]) + 1)
        self.num_items = m = int(np.max(self.ratings[:, 1]) + 1)

        # built exactly as the reference builds them (pmf_cy.pyx:69-72): set iteration order
        # decides ties in the selection loop, so the construction sequence is part of the API
        # (from_dense() defers this very construction until somebody looks at the sets)
        if isinstance(knowable, _PairsRecipe):
            self.__dict__['_pairs_recipe'] = knowable
        else:
            self.rated = set(map(tuple, self.ratings[:, :2]))
            if knowable is None:
                knowable = itertools.product(range(n), range(m))
            self.unrated = set(knowable).difference(self.rated)

        self.users = np.random.random((n, self.latent_d))
        self.items = np.random.random((m, self.latent_d))

    # ------------------------------------------------------------------ the rated / unrated sets
    def _materialize_pairs(self):
        rec = self.__dict__.get('_pairs_recipe')
        if rec is not None:
            self.__dict__['_pairs_recipe'] = None
            self.__dict__['_rated_set'], self.__dict__['_unrated_set'] = rec.build(self.num_users, self.num_items)

    def _pairs_get(name):
        def get(self):
            self._materialize_pairs()
            return self.__dict__.get(name)

        def set_(self, value):
            self._materialize_pairs()
            self.__dict__[name] = value
        return property(get, set_)

    rated = _pairs_get('_rated_set')
    unrated = _pairs_get('_unrated_set')
    del _pairs_get

    def _n_rated(self):
        rec = self.__dict__.get('_pairs_recipe')
        return rec.n_rated if rec is not None else len(self.rated)

    def _n_unrated(self):
        rec = self.__dict__.get('_pairs_recipe')
        return rec.n_unrated if rec is not None else len(self.unrated)

    @classmethod
    def from_dense(cls, real, known, knowable=None, **kwargs):
        """Model from the dense inputs of the reference's drivers: `real` (n, m) and the boolean
        mask `known` of the initially observed entries; `knowable` an optional (n, m) mask of the
        entries that may be queried (default: all).  `ratings` is the list make_ratings() writes
        (choose_training.py:13-17: row-major order of the mask), the device container is built from
        the dense arrays by device kernels (apmf_ratings_create_dense) and the `rated` / `unrated`
        sets - 10^6 tuples at 943 x 1682 - are only built if something asks for them."""
        real = np.asarray(real)
        known = np.asarray(known, dtype=bool)
        if real.ndim != 2 or known.shape != real.shape:
            raise TypeError("real and known must be (n, m) arrays of the same shape")
        ii, jj = known.nonzero()
        ratings = np.column_stack([ii, jj, real[ii, jj]]).astype(float)
        n, m = int(ii.max()) + 1, int(jj.max()) + 1          # the reference's rule (pmf_cy.pyx:66-67)
        kn = None if knowable is None else np.asarray(knowable, dtype=bool)[:n, :m]
        recipe = _PairsRecipe(ratings[:, :2].copy(), kn, n, m)
        self = cls(ratings, knowable=recipe, **kwargs)
        dev = self._device()
        dev.ratings = _lib.Ratings.from_dense(dev.ctx, np.asarray(real, float)[:n, :m], known[:n, :m])
        dev.ratings_key = self._ratings_key()
        return self

    # ------------------------------------------------------------------ copying / pickling
    def __copy__(self):
        res = type(self)(self.ratings, self.latent_d)
        res.__setstate__(res.__getstate__())   # sic (pmf_cy.pyx:78-82)
        return res

    def __deepcopy__(self, memodict):
        rec = self._recipe_clone()
        res = type(self)(self.ratings.copy(), **({} if rec is None else {'knowable': rec}))
        res.__setstate__(deepcopy(self.__getstate__(_pairs=rec is None), memodict))
        return res

    def _recipe_clone(self):
        """A model built by from_dense() whose rated / unrated sets nobody has asked for yet is copied
        WITHOUT building them (10^6 tuples at 943 x 1682, per lookahead hypothesis): the copy gets its
        own recipe and masks."""
        rec = self.__dict__.get('_pairs_recipe')
        return rec.clone() if rec is not None else None

    def __setstate__(self, state):
        if state is None:
            raise TypeError("Argument 'state' must not be None")
        for k, v in state.items():
            if k == '__dict__':
                for real_k, real_v in v.items():
                    if real_k != '_dev':
                        setattr(self, real_k, real_v)
            else:
                setattr(self, k, v)

    def __getstate__(self, _pairs=True):
        # _pairs=False (deep copies of a model whose sets are still a recipe): everything but the sets
        return {k: getattr(self, k) for k in _PUBLIC_STATE if _pairs or k not in ('rated', 'unrated')}

    # the storage behind the rated / unrated properties: state carries only 'rated' / 'unrated', like
    # the reference's (pmf_cy.pyx:90-126)
    _private_attrs = frozenset(('_dev', '_rated_set', '_unrated_set', '_pairs_recipe'))

    def _extra_dict(self):
        """What a Python subclass of the reference's cdef class keeps in its __dict__."""
        return {k: v for k, v in self.__dict__.items()
                if k not in self._base_attrs and k not in self._private_attrs}

    # ------------------------------------------------------------------ device plumbing
    def _device(self):
        dev = self.__dict__.get('_dev')
        if dev is None:
            dev = _DeviceCache()
            self.__dict__['_dev'] = dev
        if dev.ctx is None:
            dev.ctx = _lib.default_context()
        return dev

    def _device_ratings(self):
        """CSR-by-user/by-item of self.ratings on the device; rebuilt when the array changes
        (add_ratings rebinds self.ratings, fit_minibatches shuffles it in place)."""
        dev = self._device()
        r = self.ratings
        key = self._ratings_key()
        if dev.ratings is None or dev.ratings_key != key:
            if dev.ratings is not None:
                dev.ratings.free()
            dev.ratings = dev.ctx.ratings(r, self.num_users, self.num_items)
            dev.ratings_key = key
        return dev.ctx, dev.ratings

    def _ratings_key(self):
        r = self.ratings
        return (id(r), r.shape[0], float(r[:, 2].sum()), float(r[-1, 0]), float(r[-1, 1]),
                self.num_users, self.num_items)

    def _hyper(self):
        return (self.latent_d, float(self.sigma_sq), float(self.sigma_u_sq), float(self.sigma_v_sq),
                bool(self.subtract_mean), float(self.mean_rating))

    def _check_factors(self, users, items):
        users = np.ascontiguousarray(self.users if users is None else users, dtype=np.float64)
        items = np.ascontiguousarray(self.items if items is None else items, dtype=np.float64)
        if users.shape != (self.num_users, self.latent_d) or \
                items.shape != (self.num_items, self.latent_d):
            raise ValueError("users/items have the wrong shape")
        return users, items

    # ------------------------------------------------------------------ data
    def add_rating(self, i, j, rating):
        self.add_ratings([int(i), int(j), float(rating)])

    def add_ratings(self, extra):
        cols = self.ratings.shape[1]
        extra = np.array(extra, ndmin=2)
        if len(extra.shape) != 2 or extra.shape[1] != cols:
            raise TypeError("bad shape for extra")
        assert np.max(extra[:, 0] + 1) <= self.num_users
        assert np.max(extra[:, 1] + 1) <= self.num_items

        rating_vals = getattr(self, 'rating_values', None)
        if rating_vals is not None:
            if not set(rating_vals).issuperset(extra[:, 2]):
                raise ValueError("got ratings with bad values")

        new_items = set((int(i), int(j)) for i, j in extra[:, :2])
        rec = self.__dict__.get('_pairs_recipe')
        if rec is not None:
            # the sets have not been asked for yet: keep the masks and the replay list up to date
            ni = np.array(sorted(new_items), dtype=np.int64).reshape(-1, 2)
            if rec.rated_mask[ni[:, 0], ni[:, 1]].any():
                raise ValueError("can't rate already rated items")
            rec.adds.append(new_items)
            rec.n_unrated -= int(rec.cand_mask[ni[:, 0], ni[:, 1]].sum())
            rec.rated_mask[ni[:, 0], ni[:, 1]] = True
            rec.cand_mask[ni[:, 0], ni[:, 1]] = False
            rec.n_rated += len(new_items)
        else:
            if not new_items.isdisjoint(self.rated):
                raise ValueError("can't rate already rated items")
            self.rated.update(new_items)
            self.unrated.difference_update(new_items)

        self.ratings = np.append(self.ratings, extra, 0)
        self.mean_rating = float(np.mean(self.ratings[:, 2]))

    # ------------------------------------------------------------------ model evaluation
    def prediction_for(self, i, j, users=None, items=None):
        users = self.users if users is None else users
        items = self.items if items is None else items
        p = float(np.dot(users[int(i)], items[int(j)]))
        return p + self.mean_rating if self.subtract_mean else p

    def log_likelihood(self, users=None, items=None):
        """pmf_cy.pyx:170-193, on the device (K1: SDDMM over the CSR + norms)."""
        return self._ll_and_sqerr(users, items)[0]

    def _ll_and_sqerr(self, users=None, items=None):
        users, items = self._check_factors(users, items)
        ctx, rt = self._device_ratings()
        d, sig, sig_u, sig_v, sub, mr = self._hyper()
        return ctx.pmf_log_likelihood(rt, ctx.array(users), ctx.array(items), d, sig, sig_u, sig_v,
                                      sub, mr)

    def ll_prior_adjustment(self):
        return float(-.5 * (np.log(self.sigma_sq) * self.ratings.shape[0]
                            + self.num_users * self.latent_d * np.log(self.sigma_u_sq)
                            + self.num_items * self.latent_d * np.log(self.sigma_v_sq)))

    def full_ll(self, users=None, items=None):
        return self.log_likelihood(users, items) + self.ll_prior_adjustment()

    def gradient(self, ratings=None):
        """pmf_cy.pyx:204-223, on the device (K1: gather over CSR-by-user and CSR-by-item)."""
        users, items = self._check_factors(None, None)
        d, sig, sig_u, sig_v, sub, mr = self._hyper()
        if ratings is None:
            ctx, rt = self._device_ratings()
            tmp = None
        else:   # a minibatch slice: its own short-lived CSR
            ctx = self._device().ctx
            rt = tmp = ctx.ratings(ratings, self.num_users, self.num_items)
        gU, gV = ctx.empty(users.shape), ctx.empty(items.shape)
        ctx.pmf_gradient(rt, ctx.array(users), ctx.array(items), d, sig, sig_u, sig_v, sub, mr, gU, gV)
        res = gU.to_host(), gV.to_host()
        if tmp is not None:
            tmp.free()
        return res

    def update_sigma(self):
        sq_error = self._ll_and_sqerr()[1]
        self.sigma_sq = sq_error / self.ratings.shape[0]

    def update_sigma_uv(self):
        d, n, m = self.latent_d, self.num_users, self.num_items
        user_norm2 = np.sum(self.users * self.users)
        item_norm2 = np.sum(self.items * self.items)
        if self.sig_u_var > 0:
            self.sigma_u_sq = user_norm2 / (
                n * d + 2 + 2 * (np.log(self.sigma_u_sq) - self.sig_u_mean) / self.sig_u_var)
        else:
            self.sigma_u_sq = user_norm2 / n / d
        if self.sig_v_var > 0:
            self.sigma_v_sq = item_norm2 / (
                m * d + 2 + 2 * (np.log(self.sigma_v_sq) - self.sig_v_mean) / self.sig_v_var)
        else:
            self.sigma_v_sq = item_norm2 / m / d

    # ------------------------------------------------------------------ batch fit
    def _fit_on_device(self, max_steps, lr, old_ll):
        """Run up to max_steps accepted iterations of the learning-rate search
        (pmf_cy.pyx:265-291) on the device, starting from self.users/items."""
        ctx, rt = self._device_ratings()
        users, items = self._check_factors(None, None)
        d, sig, sig_u, sig_v, sub, mr = self._hyper()
        dU, dV = ctx.array(users), ctx.array(items)
        trace, steps, lr, conv, ll = ctx.pmf_fit_lls(
            rt, dU, dV, d, sig, sig_u, sig_v, sub, mr, lr, float(self.min_learning_rate),
            float(self.stop_thresh), max_steps=max_steps, old_ll=old_ll, return_ll=True)
        if steps:
            # the reference rebinds to fresh arrays on every accepted step
            self.users, self.items = dU.to_host(), dV.to_host()
        return trace, lr, conv, ll

    def fit_lls(self):
        """Generator of the log-likelihood after each accepted step, like the reference's:
        state is visible (and sigmas may be changed) between yields."""
        lr, old_ll, conv = float(self.learning_rate), None, False
        while not conv:
            trace, lr, conv, old_ll = self._fit_on_device(1, lr, old_ll)
            for ll in trace:
                yield float(ll)

    def fit(self):
        # same result as exhausting fit_lls(), without the per-step host round trip
        self._fit_on_device(-1, float(self.learning_rate), None)

    def do_fit(self):
        kind, *args = self.fit_type
        if kind == 'batch':
            self.fit(*args)
        elif kind == 'mini-valid':
            self.fit_minibatches_until_validation(*args)
        else:
            raise ValueError("unknown fit type '{}'".format(kind))

    # ------------------------------------------------------------------ minibatch fit
    def fit_minibatches(self, batch_size, lr=1, momentum=.8, ratings=None):
        """pmf_cy.pyx:308-351.  NOTE: like the reference this shuffles `ratings` (default
        self.ratings) in place with np.random.shuffle - the RNG stays on the host so the
        stream matches - and yields the training RMSE after every epoch."""
        batch_size = int(batch_size)
        if ratings is None:
            ratings = self.ratings
        ctx = self._device().ctx
        d, sig, sig_u, sig_v, sub, mr = self._hyper()
        n, m = self.num_users, self.num_items
        users, items = self._check_factors(None, None)
        dU, dV = ctx.array(users), ctx.array(items)
        u_inc, v_inc = ctx.zeros(users.shape), ctx.zeros(items.shape)
        rows = self.ratings[:, :2].astype(int)
        while True:
            np.random.shuffle(ratings)
            ctx.pmf_minibatch_epoch(ratings, n, m, batch_size, float(lr), float(momentum), dU, dV,
                                    u_inc, v_inc, d, sig, sig_u, sig_v, sub, mr)
            self.users, self.items = dU.to_host(), dV.to_host()
            pred = self.predicted_matrix()
            rows = self.ratings[:, :2].astype(int)
            train_pred = pred[rows[:, 0], rows[:, 1]]
            # `cdef float err` (pmf_cy.pyx:321,349)
            yield float(np.float32(np.sqrt(np.mean((train_pred - self.ratings[:, 2]) ** 2))))

    def fit_minibatches_validation(self, batch_size, valid_size, **kwargs):
        total = self.ratings.shape[0]
        valid_idx_set = set(random.sample(range(total), int(valid_size)))
        train_idx = tuple(i for i in range(total) if i not in valid_idx_set)
        train = self.ratings[train_idx, :]
        valid_idx = list(valid_idx_set)
        valid_ijs = tuple(self.ratings[valid_idx, :2].T.astype(int))
        valid_real = self.ratings[valid_idx, 2]
        for train_err in self.fit_minibatches(batch_size, ratings=train, **kwargs):
            valid_pred = self.predicted_matrix()[valid_ijs]
            valid_err = float(np.float32(np.sqrt(np.mean((valid_pred - valid_real) ** 2))))
            yield train_err, valid_err

    def fit_minibatches_until_validation(self, *args, stop_thresh=1e-3, **kw):
        last_valid = np.float32(np.inf)
        for train, valid in self.fit_minibatches_validation(*args, **kw):
            if valid > last_valid - stop_thresh:
                break
            last_valid = valid

    # ------------------------------------------------------------------ hyper-parameter fits
    def fit_with_sigmas_lls(self, noise_every=5, users_every=2):
        cont = True
        while cont:
            cont = False
            for i, ll in enumerate(self.fit_lls()):
                if i % noise_every == 0:
                    self.update_sigma()
                if i % users_every == 0:
                    self.update_sigma_uv()
                yield ll
                cont = True
            self.update_sigma()
            self.update_sigma_uv()

    def fit_with_sigmas(self, noise_every=10, users_every=5):
        for ll in self.fit_with_sigmas_lls(noise_every, users_every):
            pass

    # ------------------------------------------------------------------ predictions
    def predicted_matrix(self, u=None, v=None):
        """U V' (+ mean_rating), pmf_cy.pyx:410-420, as a dense (n, m) float64 array."""
        u = np.ascontiguousarray(self.users if u is None else u, dtype=np.float64)
        v = np.ascontiguousarray(self.items if v is None else v, dtype=np.float64)
        ctx = self._device().ctx
        n, m, d = u.shape[0], v.shape[0], u.shape[1]
        ld = _lib.grid_ld(m)
        out = ctx.empty((n, ld))
        ctx.pmf_predicted_matrix(ctx.array(u), ctx.array(v), n, m, d,
                                 self.mean_rating if self.subtract_mean else 0., out, ld)
        return ctx.download_grid(out, m)

    def rmse(self, real, on=None):
        if on is None:
            return rmse(self.predicted_matrix(), real)
        return rmse_on(self.predicted_matrix(), real, on)

    def print_latent_vectors(self):
        print("Users:")
        for i in range(self.num_users):
            print("%d: %s" % (i, self.users[i, :]))
        print("\nItems:")
        for j in range(self.num_items):
            print("%d: %s" % (j, self.items[j, :]))

    def save_latent_vectors(self, prefix):
        self.users.dump(prefix + "%sd_users.pickle" % self.latent_d)
        self.items.dump(prefix + "%sd_items.pickle" % self.latent_d)


def parse_fit_type(string):
    """'mini-valid,100,50' -> ('mini-valid', 100, 50)  (pmf_cy.pyx:444-456)"""
    def conv(x):
        for fn in (int, float):
            try:
                return fn(x)
            except ValueError:
                pass
        return x
    return tuple(conv(x) for x in string.split(','))


def fake_ratings(noise=.25, num_users=100, num_items=100, num_ratings=30, latent_dimension=10):
    """Synthetic data of the reference's demo (pmf_cy.pyx:461-474): same RNG call sequence."""
    u = np.random.normal(0, 2, (num_users, latent_dimension))
    v = np.random.normal(0, 2, (num_items, latent_dimension))
    ratings = []
    for i in range(num_users):
        for j in random.sample(range(num_items), num_ratings):
            ratings.append((i, j, np.dot(u[i], v[j]) + np.random.normal(scale=noise)))
    return (np.array(ratings), u, v)
