#!/usr/bin/env python3
"""configs[3] as the reference's Makefile runs it (results/drugbank-94x425/Makefile:62-73: rank 20, keys random / pred-variance /
pred / prob-ge-3.5), two query steps per key through mn_active_pmf.compare() under the seed of fixture section AF: wall time
and the picks (the reference needs 19 minutes for the same run).  python tools/time_c4_makefile.py"""
import contextlib
import io
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from active_matrix_factorization_amd import mn_active_pmf as mn
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'mn_c4_94x425_d5.npz'))
real, ratings = g['real'].astype(float), g['ratings']
np.random.seed(95); random.seed(95)
keys = ['random', 'pred-variance', 'pred', 'prob-ge-3.5']
t = time.perf_counter()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    res = mn.compare(keys, real, ratings.copy(), rating_vals=(0, 1), latent_d=20, do_threading=False, steps=3, discrete_exp=True, refit_lookahead=True)
print('compare() rank 20, 4 keys, 2 query steps each: %.2f s' % (time.perf_counter() - t))
for k in keys:
    print(k, [s[2] for s in res[k][1:]], [round(float(s[1]), 5) for s in res[k]])
