#!/usr/bin/env python3
"""K1 probe (dev tool): grad-steps/s of apmf_pmf_fit_lls at the bench shape."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from active_matrix_factorization_amd import _lib
ctx = _lib.Context(0)
ratings, _ = bench.synth_ratings(0)
n, m, d = bench.N_USERS, bench.N_ITEMS, bench.LATENT_D
rt = ctx.ratings(ratings, n, m)
rs = np.random.RandomState(7)
dU, dV = ctx.array(rs.random_sample((n, d))), ctx.array(rs.random_sample((m, d)))
lr = 1e-4
for G in (25, 200):
    _, _, lr, _ = ctx.pmf_fit_lls(rt, dU, dV, d, 1., 10., 10., False, 0., lr, 1e-10, 0., max_steps=G, max_trace=0)
    t0 = time.perf_counter(); tot = 0
    for _ in range(10):
        _, s, lr, _ = ctx.pmf_fit_lls(rt, dU, dV, d, 1., 10., 10., False, 0., lr, 1e-10, 0., max_steps=G, max_trace=0)
        tot += s
    dt = time.perf_counter() - t0
    print('G=%d: %.0f grad-steps/s (%.1f us/step)' % (G, tot / dt, dt / tot * 1e6))
gU, gV = ctx.empty((n, d)), ctx.empty((m, d))
ctx.profile_begin(50)
for _ in range(50):
    ctx.pmf_gradient(rt, dU, dV, d, 1., 10., 10., False, 0., gU, gV)
ms = ctx.profile_end()
print('k_pmf_grad: min %.1f us  med %.1f us' % (ms.min() * 1e3, np.median(ms) * 1e3))
