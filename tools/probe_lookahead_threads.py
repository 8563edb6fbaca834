#!/usr/bin/env python3
"""Dev probe: where does a threaded lookahead spend its time?  (fit C-call intervals per thread)"""
import io, os, sys, random, threading, time
from contextlib import redirect_stdout
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from active_matrix_factorization_amd import mn_active_pmf as mn, _lib

np.random.seed(0); random.seed(0)
n, m, d, known = 94, 425, 5, 500
u, v = np.random.normal(0, 1, (n, d)), np.random.normal(0, 1, (m, d))
x = u @ v.T
real = np.clip(np.round(3 + x / x.std() * 1.2), 1, 5)
flat = np.random.choice(n * m, known, replace=False)
ratings = np.c_[flat // m, flat % m, real[flat // m, flat % m]].astype(float)
a = mn.MNActivePMF(ratings, latent_d=d, rating_values=range(1, 6), discrete_expectations=True)
a.do_fit(); a.initialize_approx()
for _ in zip(range(150), a.fit_normal_kls()):
    pass

events = []
orig = _lib.Context.mn_fit_normal
def timed(self, *args, **kw):
    t0 = time.perf_counter()
    r = orig(self, *args, **kw)
    events.append((threading.current_thread().name, t0, time.perf_counter(), r[1]))
    return r
_lib.Context.mn_fit_normal = timed

pool = sorted(a.unrated)[:4]
key = mn.KEY_FUNCS['total-variance']
allvals = {}
for procs in (1, 4):
    events.clear()
    t = time.perf_counter()
    with redirect_stdout(io.StringIO()):
        vals = a._get_key_vals(pool, key, procs=procs)
    wall = time.perf_counter() - t
    allvals[procs] = list(vals)
    infit = sum(e[2] - e[1] for e in events)
    steps = sum(e[3] for e in events)
    print('procs=%d wall %.2f s; %d fit calls, %d accepted steps, summed time inside the C call %.2f s; threads %s'
          % (procs, wall, len(events), steps, infit, sorted(set(e[0] for e in events))), flush=True)
print('values identical across widths:', allvals[1] == allvals[4], allvals[1][:2])
