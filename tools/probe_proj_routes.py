#!/usr/bin/env python3
"""Dev probe: the two projection routes of a batched fit (lowest eigenpairs / full eigendecomposition) against
each other along a search started from the identity covariances: state differences after a growing number of
accepted steps, and how many projections each route did."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from active_matrix_factorization_amd import _lib
from test_gpu_batch import _problem, _unrated

ctx = _lib.Context(0)
n, m, d, seed = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (70, 90, 3, 5)))
rs, ratings, mean, S, O = _problem(n, m, d, seed)
N = n + m
H = 4
extra = _unrated(rs, ratings, n, m, H)
rt = ctx.ratings(ratings, n, m)
for steps in (10, 20, 40, 80, 160, 320, 640):
    res = {}
    for mode in (None, 'full'):
        if mode:
            os.environ['APMF_BATCH_EIG'] = mode
        else:
            os.environ.pop('APMF_BATCH_EIG', None)
        b = ctx.mn_batch(rt, d, H)
        b.set_state(ctx.array(.1 * mean), ctx.array(np.eye(N)), ctx.array(np.eye(d)))
        b.set_extra([e[0] for e in extra], [e[1] for e in extra], [float(1 + k % 5) for k in range(H)])
        lr, kl, nsteps, conv, _ = b.fit(1., 10., 10., 1e-5, 1e-4, max_steps=steps)
        st = b.fit_stats()
        ds = ctx.empty((N, N))
        Ss = []
        for q in range(H):
            b.get_state(q, None, ds, None)
            Ss.append(ds.to_host().copy())
        res[mode] = (st, kl.copy(), Ss, lr.copy())
        b.free()
    a, f = res[None], res['full']
    print(steps, 'low', a[0]['proj_low'], a[0]['proj_full'], 'full', f[0]['proj_full'], 'trials', a[0]['trials'], f[0]['trials'],
          'dS', ['%.1e' % np.abs(x - y).max() for x, y in zip(a[2], f[2])],
          'dKL', ['%.1e' % abs(x - y) for x, y in zip(a[1], f[1])], 'lr equal', (a[3] == f[3]).tolist(), flush=True)
