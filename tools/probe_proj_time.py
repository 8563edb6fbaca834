#!/usr/bin/env python3
"""Dev probe: time of a batched fit by projection route for a few sizes (same search started from covariances
with a few eigenvalues near the clipping threshold)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from active_matrix_factorization_amd import _lib
from test_gpu_batch import _problem, _unrated

ctx = _lib.Context(0)
for (n, m, d) in ((10, 10, 2), (30, 40, 3), (50, 78, 3), (70, 122, 3), (100, 156, 4), (150, 234, 5)):
    rs, ratings, mean, S, O = _problem(n, m, d, 11)
    N = n + m
    H = 8
    w, v = np.linalg.eigh(S)
    S1 = (v * np.concatenate([np.full(2, 3e-5), w[2:]])) @ v.T
    S1 = (S1 + S1.T) / 2
    rt = ctx.ratings(ratings, n, m)
    for rep in range(2):
        for mode in (None, 'full'):
            if mode:
                os.environ['APMF_BATCH_EIG'] = mode
            else:
                os.environ.pop('APMF_BATCH_EIG', None)
            b = ctx.mn_batch(rt, d, H)
            b.set_state(ctx.array(mean), ctx.array(S1), ctx.array(O))
            ex = _unrated(np.random.RandomState(3), ratings, n, m, H)
            b.set_extra([e[0] for e in ex], [e[1] for e in ex], [float(1 + k % 5) for k in range(H)])
            ctx.sync()
            t = time.perf_counter()
            lr, kl, nsteps, conv, _ = b.fit(1., 10., 10., 1e-5, 1e-5, max_steps=40)
            dt = time.perf_counter() - t
            if rep:
                print(N, mode or 'eigenpairs', '%.3f s' % dt, b.fit_stats(), flush=True)
            b.free()
