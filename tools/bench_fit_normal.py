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
