#!/usr/bin/env python3
"""Variational fit (fit_normal_kls) at the MovieLens shape: time per
accepted step and where it goes.  Dev tool; numbers go to BASELINE.md."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from active_matrix_factorization_amd import mn_active_pmf, matrix_normal_exps_cy as mnx
from active_matrix_factorization_amd._active_base import project_psd

# SHAPE="n,m,d,N" (default: the MovieLens shape of bench.py); e.g. SHAPE=94,425,5,500 = configs[3]
shape = os.environ.get('SHAPE')
np.random.seed(1)
if shape:
    n_, m_, d_, N_ = (int(x) for x in shape.split(','))
    flat = np.random.choice(n_ * m_, N_, replace=False)
    ratings = np.c_[flat // m_, flat % m_, np.random.randint(1, 6, N_)].astype(float)
else:
    d_ = 10
    ratings, _ = bench.synth_ratings(0)
a = mn_active_pmf.MNActivePMF(ratings, latent_d=d_)
a.do_fit()
a.initialize_approx()
out = {}
mnx.matrixnormal_gradient(a); a.kl_divergence(); project_psd(a.cov_useritems.copy(), 1e-5)   # warm-up (first use pages rocBLAS in)
t0 = time.perf_counter(); g = mnx.matrixnormal_gradient(a); out['gradient_s'] = time.perf_counter() - t0
t0 = time.perf_counter(); kl = a.kl_divergence(); out['kl_s'] = time.perf_counter() - t0
t0 = time.perf_counter(); p = project_psd(a.cov_useritems - 1e-4 * g[1], 1e-5); out['project_psd_s'] = time.perf_counter() - t0
trials = [0]
_orig = a.kl_divergence
def _counting(*args, **kw):
    trials[0] += 1
    return _orig(*args, **kw)
a.kl_divergence = _counting
it = a.fit_normal_kls()
nsteps = int(os.environ.get('STEPS', 3))
t0 = time.perf_counter(); kls = [next(it) for _ in range(nsteps)]; out['accepted_steps'] = nsteps; out['accepted_steps_s'] = time.perf_counter() - t0
out['kls'] = kls[:3]; out['kl_evaluations_in_those_steps'] = trials[0]
out['s_per_trial'] = out['accepted_steps_s'] / max(trials[0], 1)
print(json.dumps(out, indent=1))

if os.environ.get('FULL'):
    # the whole fit: how often the spectrum really had to be clipped, time per phase
    from active_matrix_factorization_amd import _lib as _l, _active_base as _ab
    ctx = _l.default_context()
    stats = {'psd_calls': 0, 'psd_clipped': 0, 'psd_s': 0., 'grad_s': 0., 'kl_s': 0., 'grad_calls': 0}
    _pp = ctx.dense_project_psd
    def _psd(*args, **kw):
        t = time.perf_counter(); r = _pp(*args, **kw); stats['psd_s'] += time.perf_counter() - t
        stats['psd_calls'] += 1; stats['psd_clipped'] += bool(r); return r
    ctx.dense_project_psd = _psd
    _g = mnx.matrixnormal_gradient
    def _grad(x):
        t = time.perf_counter(); r = _g(x); stats['grad_s'] += time.perf_counter() - t; stats['grad_calls'] += 1; return r
    mn_active_pmf.matrixnormal_gradient = _grad
    _k = _orig
    def _kl(*args, **kw):
        t = time.perf_counter(); r = _k(*args, **kw); stats['kl_s'] += time.perf_counter() - t; trials[0] += 1; return r
    a.kl_divergence = _kl
    trials[0] = 0
    t0 = time.perf_counter(); n_acc = sum(1 for _ in a.fit_normal_kls()); stats['total_s'] = time.perf_counter() - t0
    stats['accepted_after_warmup'] = n_acc; stats['trials'] = trials[0]
    print(json.dumps(stats, indent=1))
