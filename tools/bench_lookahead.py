#!/usr/bin/env python3
"""Batched lookahead at BASELINE.json configs[3]'s shape (94 x 425 rank 5, 500 known ratings): the
'total-variance' criterion over the first K unrated entries (5 K hypotheses fitted to convergence in one device
batch).  Prints JSON; run under rocprofv3 --kernel-trace --stats for the per-kernel breakdown."""
import json
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('APMF_SEGV_BT'):       # diagnosis runs (tools/repro_trace_segv.sh): native + Python stacks on a fault
    import ctypes
    import faulthandler
    faulthandler.enable(all_threads=True)
    ctypes.CDLL(os.environ['APMF_SEGV_BT'])
from active_matrix_factorization_amd import mn_active_pmf as mn  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
np.random.seed(0)
random.seed(0)
n, m, d, known = 94, 425, 5, 500
u, v = np.random.normal(0, 1, (n, d)), np.random.normal(0, 1, (m, d))
x = u @ v.T
real = np.clip(np.round(3 + x / x.std() * 1.2), 1, 5)
flat = np.random.choice(n * m, known, replace=False)
ratings = np.c_[flat // m, flat % m, real[flat // m, flat % m]].astype(float)
a = mn.MNActivePMF(ratings, latent_d=d, rating_values=range(1, 6), discrete_expectations=True)
a.do_fit()
a.initialize_approx()
a.fit_normal()
pool = sorted(a.unrated)[:K]
a._get_key_vals(pool[:1], mn.KEY_FUNCS['total-variance'])          # warm: library code objects
t = time.perf_counter()
vals = a._get_key_vals(pool, mn.KEY_FUNCS['total-variance'])
dt = time.perf_counter() - t
print(json.dumps({'candidates': K, 'hypotheses': 5 * K, 'seconds': dt, 'hypotheses_per_s': 5 * K / dt,
                  'argmin': int(np.argmin(vals)), 'min': float(np.min(vals))}))
