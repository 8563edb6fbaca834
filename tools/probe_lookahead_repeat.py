#!/usr/bin/env python3
"""Dev probe: is one lookahead hypothesis (deep copy + add_rating + fit_normal + total variance)
reproducible when repeated in one process?"""
import io, os, sys, random
from contextlib import redirect_stdout
from copy import deepcopy
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
a.do_fit(); a.initialize_approx()
for _ in zip(range(100), a.fit_normal_kls()):
    pass
base = (a.mean.copy(), a.cov_useritems.copy(), a.cov_latents.copy())
res = []
for rep in range(3):
    b = deepcopy(a)
    b.add_rating(0, 0, 3)
    kls = list(b.fit_normal_kls())
    tv = b._total_variance()
    res.append((len(kls), kls[0], kls[-1], tv, b.cov_useritems.copy()))
    print(rep, len(kls), repr(kls[0]), repr(kls[-1]), repr(tv), flush=True)
print('base unchanged', all(np.array_equal(p, q) for p, q in zip(base, (a.mean, a.cov_useritems, a.cov_latents))))
print('identical', all(r[0] == res[0][0] and r[3] == res[0][3] and np.array_equal(r[4], res[0][4]) for r in res[1:]))
