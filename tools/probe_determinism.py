#!/usr/bin/env python3
"""Dev probe: are the building blocks of the variational fit bit-reproducible call to call?"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from active_matrix_factorization_amd import _lib

ctx = _lib.default_context()
rs = np.random.RandomState(0)
n, m, d, N = 94, 425, 5, 500
flat = rs.choice(n * m, N, replace=False)
ratings = np.c_[flat // m, flat % m, rs.randint(1, 6, N)].astype(float)
rt = ctx.ratings(ratings, n, m)
T = n + m
mean = rs.normal(size=(T, d))
a = rs.normal(size=(T, T)) / np.sqrt(T)
S = a @ a.T + .5 * np.eye(T)
b = rs.normal(size=(d, d))
O = b @ b.T / d + .5 * np.eye(d)


def same(x, y):
    return bool(np.array_equal(np.asarray(x), np.asarray(y)))


out = {}
# inverse / logdet
res = []
for _ in range(3):
    inv = ctx.empty((T, T))
    ld = ctx.dense_inv_logdet(ctx.array(S), T, inv, want_logdet=True)
    res.append((inv.to_host(), ld))
out['inv_logdet'] = all(same(res[0][0], r[0]) and res[0][1] == r[1] for r in res[1:])
# project_psd, clipped path
res = []
for _ in range(3):
    x = ctx.array(S - 0.6 * np.eye(T))
    ctx.dense_project_psd(x, T, 1e-5)
    res.append(x.to_host())
out['project_psd_clipped'] = all(same(res[0], r) for r in res[1:])
# gradient / value
res = []
for _ in range(3):
    g = ctx.empty((T, d)), ctx.empty((T, T)), ctx.empty((d, d))
    ctx.mn_kl_gradient(rt, d, ctx.array(mean), ctx.array(S), ctx.array(O), 1., 10., 10., *g)
    kl = ctx.mn_kl_value(rt, d, ctx.array(mean), ctx.array(S), ctx.array(O), 1., 10., 10.)
    res.append([x.to_host() for x in g] + [kl])
out['kl_gradient'] = all(all(same(a_, b_) for a_, b_ in zip(res[0][:3], r[:3])) for r in res[1:])
out['kl_value'] = all(res[0][3] == r[3] for r in res[1:])
# the fit loop
res = []
for _ in range(3):
    dm, dS, dO = ctx.array(mean), ctx.array(np.eye(T)), ctx.array(np.eye(d))
    tr = ctx.mn_fit_normal(rt, d, dm, dS, dO, 1., 10., 10., 1e-5, 1e-4, None, max_steps=150)
    res.append((tr[0], dS.to_host()))
out['fit_150_steps_trace'] = all(same(res[0][0], r[0]) for r in res[1:])
out['fit_150_steps_cov'] = all(same(res[0][1], r[1]) for r in res[1:])
if not out['fit_150_steps_trace']:
    k = int(np.argmax(res[0][0][:min(len(res[0][0]), len(res[1][0]))] != res[1][0][:min(len(res[0][0]), len(res[1][0]))]))
    out['first_divergent_step'] = k
print(out)
