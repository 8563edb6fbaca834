#!/usr/bin/env python3
"""Dev probe: a batched lookahead (10 hypotheses, slow lanes and worker threads in play) gives the same
values when repeated in one process and across lane counts."""
import os, sys, random
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from active_matrix_factorization_amd import mn_active_pmf as mn

np.random.seed(0); random.seed(0)
n, m, d, known = 94, 425, 5, 500
u, v = np.random.normal(0, 1, (n, d)), np.random.normal(0, 1, (m, d))
x = u @ v.T
real = np.clip(np.round(3 + x / x.std() * 1.2), 1, 5)
flat = np.random.choice(n * m, known, replace=False)
ratings = np.c_[flat // m, flat % m, real[flat // m, flat % m]].astype(float)
a = mn.MNActivePMF(ratings, latent_d=d, rating_values=range(1, 6), discrete_expectations=True)
a.do_fit(); a.initialize_approx(); a.fit_normal()
pool = sorted(a.unrated)[:2]
out = []
import time
for lanes in (sys.argv[1].split(',') if len(sys.argv) > 1 else ('3', '1', '3')):
    os.environ['APMF_BATCH_LANES'] = lanes
    t = time.perf_counter()
    vals = a._get_key_vals(pool, mn.KEY_FUNCS['total-variance'])
    print('lanes %s: %.3f s' % (lanes, time.perf_counter() - t), flush=True)
    out.append(vals)
    print(lanes, [repr(float(x)) for x in vals], flush=True)
assert all(o == out[0] for o in out), 'batched lookahead values differ between runs / lane counts'
print('identical')
