#!/usr/bin/env python3
"""Secondary timings at the MovieLens-100k shape (943 x 1682): Gibbs sweeps (K3), sample
statistics (K4), class-level selection and fit.  Not the headline bench; numbers go to BASELINE.md."""
import os, sys, time, json
from itertools import islice
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from active_matrix_factorization_amd import _lib
from active_matrix_factorization_amd import pmf_cy, mn_active_pmf, bayes_pmf

out = {}
ratings, rated = bench.synth_ratings(0)
n, m = bench.N_USERS, bench.N_ITEMS
ctx = _lib.default_context()

# ---- K3 kernel: one half-sweep pair at d = 15
d = 15
rs = np.random.RandomState(0)
rt = ctx.ratings(ratings, n, m)
U, V = ctx.array(rs.normal(size=(n, d))), ctx.array(rs.normal(size=(m, d)))
al, mu = ctx.array(np.eye(d) * 2.), ctx.array(np.zeros(d))
zu, zv = ctx.array(rs.normal(size=(n, d))), ctx.array(rs.normal(size=(m, d)))
ou, ov = ctx.empty((n, d)), ctx.empty((m, d))
for _ in range(3):
    ctx.gibbs_half_sweep(rt, 0, V, d, al, mu, 2., 0., zu, ou)
    ctx.gibbs_half_sweep(rt, 1, ou, d, al, mu, 2., 0., zv, ov)
ctx.sync(); t0 = time.perf_counter(); R = 50
for _ in range(R):
    ctx.gibbs_half_sweep(rt, 0, V, d, al, mu, 2., 0., zu, ou)
    ctx.gibbs_half_sweep(rt, 1, ou, d, al, mu, 2., 0., zv, ov)
ctx.sync(); dt = (time.perf_counter() - t0) / R
out['gibbs_sweep_kernel_ms_d15'] = dt * 1e3
out['gibbs_row_updates_per_s_kernel'] = (n + m) / dt

# ---- class level: BayesianPMF.samples() yields (2 sweeps + hyper-parameters on host)
np.random.seed(0)
b = bayes_pmf.BayesianPMF(ratings, latent_d=15)
it = b.samples()
next(it); t0 = time.perf_counter(); S = 20
samps = list(islice(it, S))
dt = (time.perf_counter() - t0) / S
out['bayes_samples_yield_ms_d15'] = dt * 1e3
out['gibbs_row_updates_per_s_class'] = 2 * (n + m) / dt

# ---- K4: pred_variance over S = 128 samples
samps128 = (samps * 7)[:128]
b.pred_variance(samps128)
t0 = time.perf_counter()
for _ in range(3):
    v = b.pred_variance(samps128)
out['bayes_pred_variance_S128_ms_class'] = (time.perf_counter() - t0) / 3 * 1e3
Us, Vs = ctx.array(np.array([u for u, _ in samps128])), ctx.array(np.array([w for _, w in samps128]))
ld = _lib.grid_ld(m); gv = ctx.empty((n, ld))
ctx.sample_stats(128, n, m, 15, ld, Us, Vs, 0., 0., var=gv); ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
    ctx.sample_stats(128, n, m, 15, ld, Us, Vs, 0., 0., var=gv)
ctx.sync(); dt = (time.perf_counter() - t0) / 5
out['k4_S128_d15_ms'] = dt * 1e3
out['k4_GFLOPs_f64'] = 128 * n * m * (2 * 15 + 4) / dt / 1e9

# ---- MNActivePMF: f64 selection over the whole unrated pool (class level, host arrays in)
np.random.seed(1)
a = mn_active_pmf.MNActivePMF(ratings, latent_d=10)
t0 = time.perf_counter(); a.do_fit(); out['pmf_do_fit_s_class'] = time.perf_counter() - t0
a.initialize_approx(random_cov=True)
a.cov_useritems = a.cov_useritems / (n + m)
pool = list(a.unrated)
t0 = time.perf_counter()
vals = a._get_key_vals(pool, mn_active_pmf.MNActivePMF.pred_variance, None, None)
out['mn_pred_variance_full_pool_s_class'] = time.perf_counter() - t0
out['mn_pool_size'] = len(pool)
t0 = time.perf_counter(); pick = a.pick_query_point(pool=pool); out['mn_pick_query_point_full_pool_s_class'] = time.perf_counter() - t0
t0 = time.perf_counter(); tv = a._total_variance(); out['mn_total_variance_s_class'] = time.perf_counter() - t0
print(json.dumps(out, indent=1))
