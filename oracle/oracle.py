"""ctypes loader for the CPU oracle (oracle/apmf_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py - never by active_matrix_factorization_amd.
Each wrapper takes/returns float64 NumPy arrays in the reference's layouts.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libapmf_oracle.so')

_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_long)


def build(force=False):
    src = os.path.join(_HERE, 'apmf_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE])
    return _SO


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_pmf_log_likelihood.restype = C.c_double
        L.orc_pmf_log_likelihood.argtypes = [_dp, C.c_long, _dp, _dp, C.c_int, C.c_int, C.c_int,
                                             C.c_double, C.c_double, C.c_double, C.c_int, C.c_double]
        L.orc_pmf_sq_error.restype = C.c_double
        L.orc_pmf_sq_error.argtypes = [_dp, C.c_long, _dp, _dp, C.c_int, C.c_int, C.c_double]
        L.orc_pmf_gradient.restype = None
        L.orc_pmf_gradient.argtypes = [_dp, C.c_long, _dp, _dp, C.c_int, C.c_int, C.c_int,
                                       C.c_double, C.c_double, C.c_double, C.c_int, C.c_double,
                                       _dp, _dp]
        L.orc_pmf_fit_lls.restype = C.c_long
        L.orc_pmf_fit_lls.argtypes = [_dp, C.c_long, _dp, _dp, C.c_int, C.c_int, C.c_int,
                                      C.c_double, C.c_double, C.c_double, C.c_int, C.c_double,
                                      C.c_double, C.c_double, C.c_double, _dp, C.c_long, C.c_long]
        for name, nint in (('orc_mn_quadexpect', 10), ('orc_mn_exp_squared', 6),
                           ('orc_mn_exp_a2bc', 8), ('orc_mn_tripexpect', 8)):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [_dp, _dp, _dp] + [C.c_int] * nint
        L.orc_mn_exp_dotprod_sq.restype = C.c_double
        L.orc_mn_exp_dotprod_sq.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_mn_pred_means_vars.restype = None
        L.orc_mn_pred_means_vars.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_mn_pred_means_vars_compact.restype = None
        L.orc_mn_pred_means_vars_compact.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_mn_score_rows.restype = C.c_long
        L.orc_mn_score_rows.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, _dp]
        L.orc_mn_score_rows_omp.restype = C.c_long
        L.orc_mn_score_rows_omp.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, _dp,
                                            C.c_int]
        for name, nl in (('orc_full_tripexpect', 3), ('orc_full_quadexpect', 4),
                         ('orc_full_exp_squared', 2), ('orc_full_exp_a2bc', 3)):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [_dp, _dp, C.c_long] + [C.c_long] * nl
        L.orc_full_exp_dotprod_sq.restype = C.c_double
        L.orc_full_exp_dotprod_sq.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_long, C.c_int, C.c_int]
        L.orc_full_pred_means_vars.restype = None
        L.orc_full_pred_means_vars.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.orc_sample_feature.restype = C.c_int
        L.orc_sample_feature.argtypes = [_dp, C.c_int, _lp, _dp, C.c_long, _dp, _dp, C.c_double,
                                         C.c_double, _dp, _dp]
        L.orc_gibbs_half_sweep.restype = C.c_int
        L.orc_gibbs_half_sweep.argtypes = [C.c_long, C.c_int, _lp, _lp, _dp, _dp, _dp, _dp,
                                           C.c_double, C.c_double, _dp, _dp]
        L.orc_sample_stats.restype = None
        L.orc_sample_stats.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.c_double, _dp, _dp, _dp]
        _lib = L
    return _lib


# ------------------------------------------------------------------ PMF
def pmf_log_likelihood(ratings, U, V, sig=1., sig_u=10., sig_v=10., subtract_mean=False,
                       mean_rating=0.):
    ratings, U, V = _f64(ratings), _f64(U), _f64(V)
    return lib().orc_pmf_log_likelihood(_p(ratings), ratings.shape[0], _p(U), _p(V), U.shape[0],
                                        V.shape[0], U.shape[1], sig, sig_u, sig_v,
                                        int(subtract_mean), mean_rating)


def pmf_sq_error(ratings, U, V, subtract_mean=False, mean_rating=0.):
    ratings, U, V = _f64(ratings), _f64(U), _f64(V)
    return lib().orc_pmf_sq_error(_p(ratings), ratings.shape[0], _p(U), _p(V), U.shape[1],
                                  int(subtract_mean), mean_rating)


def pmf_gradient(ratings, U, V, sig=1., sig_u=10., sig_v=10., subtract_mean=False,
                 mean_rating=0.):
    ratings, U, V = _f64(ratings), _f64(U), _f64(V)
    gU, gV = np.empty_like(U), np.empty_like(V)
    lib().orc_pmf_gradient(_p(ratings), ratings.shape[0], _p(U), _p(V), U.shape[0], V.shape[0],
                           U.shape[1], sig, sig_u, sig_v, int(subtract_mean), mean_rating,
                           _p(gU), _p(gV))
    return gU, gV


def pmf_fit_lls(ratings, U, V, sig=1., sig_u=10., sig_v=10., subtract_mean=False,
                mean_rating=0., lr=1e-4, min_lr=1e-10, stop_thresh=1e-2, max_steps=-1,
                max_trace=100000):
    ratings = _f64(ratings)
    U, V = _f64(U).copy(), _f64(V).copy()
    trace = np.empty(max_trace)
    k = lib().orc_pmf_fit_lls(_p(ratings), ratings.shape[0], _p(U), _p(V), U.shape[0],
                              V.shape[0], U.shape[1], sig, sig_u, sig_v, int(subtract_mean),
                              mean_rating, lr, min_lr, stop_thresh, _p(trace), max_trace,
                              max_steps)
    return U, V, trace[:min(k, max_trace)].copy()


# ------------------------------------------------------------------ matrix normal
def mn_exp_dotprod_sq(num_users, mean, R, Cl, i, j):
    mean, R, Cl = _f64(mean), _f64(R), _f64(Cl)
    return lib().orc_mn_exp_dotprod_sq(num_users, _p(mean), _p(R), _p(Cl), R.shape[0],
                                       mean.shape[1], i, j)


def mn_moment(name, mean, R, Cl, *idx):
    mean, R, Cl = _f64(mean), _f64(R), _f64(Cl)
    return getattr(lib(), 'orc_mn_' + name)(_p(mean), _p(R), _p(Cl), R.shape[0], mean.shape[1], *idx)


def mn_pred_means_vars(n, m, mean, R, Cl, want_e2=False):
    mean, R, Cl = _f64(mean), _f64(R), _f64(Cl)
    d = mean.shape[1]
    pm, pv, e2 = np.empty((n, m)), np.empty((n, m)), np.empty((n, m))
    lib().orc_mn_pred_means_vars(n, m, d, _p(mean), _p(R), _p(Cl), _p(pm), _p(pv), _p(e2))
    return (pm, pv, e2) if want_e2 else (pm, pv)


def mn_pred_means_vars_compact(n, m, mean, cross, diag, Cl):
    mean, cross, diag, Cl = _f64(mean), _f64(cross), _f64(diag), _f64(Cl)
    pm, pv = np.empty((n, m)), np.empty((n, m))
    lib().orc_mn_pred_means_vars_compact(n, m, mean.shape[1], _p(mean), _p(cross), _p(diag),
                                         _p(Cl), _p(pm), _p(pv))
    return pm, pv


def mn_score_rows(n, m, mean, R, Cl, row_lo, row_hi):
    mean, R, Cl = _f64(mean), _f64(R), _f64(Cl)
    out = np.empty((row_hi - row_lo, m))
    lib().orc_mn_score_rows(n, m, mean.shape[1], _p(mean), _p(R), _p(Cl), row_lo, row_hi, _p(out))
    return out


def mn_score_rows_threads(n, m, mean, R, Cl, row_lo, row_hi, threads):
    """mn_score_rows with the rows dealt to `threads` OpenMP threads (bench.py's all-cores leg)."""
    mean, R, Cl = _f64(mean), _f64(R), _f64(Cl)
    out = np.empty((row_hi - row_lo, m))
    lib().orc_mn_score_rows_omp(n, m, mean.shape[1], _p(mean), _p(R), _p(Cl), row_lo, row_hi, _p(out),
                                int(threads))
    return out


# ------------------------------------------------------------------ full covariance
def full_moment(name, mean, cov, *idx):
    mean, cov = _f64(mean), _f64(cov)
    return getattr(lib(), 'orc_full_' + name)(_p(mean), _p(cov), cov.shape[0], *idx)


def full_exp_dotprod_sq(n, d, mean, cov, i, j):
    mean, cov = _f64(mean), _f64(cov)
    return lib().orc_full_exp_dotprod_sq(n, d, _p(mean), _p(cov), cov.shape[0], i, j)


def full_pred_means_vars(n, m, d, mean, cov, want_e2=False):
    mean, cov = _f64(mean), _f64(cov)
    pm, pv, e2 = np.empty((n, m)), np.empty((n, m)), np.empty((n, m))
    lib().orc_full_pred_means_vars(n, m, d, _p(mean), _p(cov), _p(pm), _p(pv), _p(e2))
    return (pm, pv, e2) if want_e2 else (pm, pv)


# ------------------------------------------------------------------ Gibbs
def sample_feature(oth, idx, rat, alpha, mu, beta, mean_sub, z):
    oth, rat, alpha, mu, z = _f64(oth), _f64(rat), _f64(alpha), _f64(mu), _f64(z)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = np.empty(oth.shape[1])
    rc = lib().orc_sample_feature(_p(oth), oth.shape[1], idx.ctypes.data_as(_lp), _p(rat),
                                  idx.size, _p(alpha), _p(mu), beta, mean_sub, _p(z), _p(out))
    if rc:
        raise np.linalg.LinAlgError('oracle sample_feature rc=%d' % rc)
    return out


def gibbs_half_sweep(ptr, idx, rat, oth, alpha, mu, beta, mean_sub, z):
    oth, rat, alpha, mu, z = _f64(oth), _f64(rat), _f64(alpha), _f64(mu), _f64(z)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    rows = ptr.size - 1
    out = np.empty((rows, oth.shape[1]))
    rc = lib().orc_gibbs_half_sweep(rows, oth.shape[1], ptr.ctypes.data_as(_lp),
                                    idx.ctypes.data_as(_lp), _p(rat), _p(oth), _p(alpha), _p(mu),
                                    beta, mean_sub, _p(z), _p(out))
    if rc:
        raise np.linalg.LinAlgError('oracle gibbs_half_sweep rc=%d' % rc)
    return out


def sample_stats(Us, Vs, add=0., cutoff=0.):
    Us, Vs = _f64(Us), _f64(Vs)
    S, n, d = Us.shape
    m = Vs.shape[1]
    mean, var, prob = np.empty((n, m)), np.empty((n, m)), np.empty((n, m))
    lib().orc_sample_stats(_p(Us), _p(Vs), S, n, m, d, add, cutoff, _p(mean), _p(var), _p(prob))
    return mean, var, prob


def csr_by_row(ratings, side, rows):
    """Stable CSR of the (N,3) ratings by user (side=0) or item (side=1): the
    per-row order is the storage order, as samples() builds it (bayes_pmf.py:241-255)."""
    ratings = _f64(ratings)
    key = ratings[:, side].astype(np.int64)
    order = np.argsort(key, kind='stable')
    ptr = np.zeros(rows + 1, np.int64)
    np.add.at(ptr, key + 1, 1)
    ptr = np.cumsum(ptr)
    return ptr, ratings[order, 1 - side].astype(np.int64), ratings[order, 2].copy()
