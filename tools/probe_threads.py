#!/usr/bin/env python3
"""Dev probe: the dense building blocks and the device fit loop, driven from several threads at
once (one Context each) against the single-thread results."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from active_matrix_factorization_amd import _lib

rs = np.random.RandomState(0)
n, m, d, N = 94, 425, 5, 500
flat = rs.choice(n * m, N, replace=False)
ratings = np.c_[flat // m, flat % m, rs.randint(1, 6, N)].astype(float)
T = n + m
mean = rs.normal(size=(T, d))
a = rs.normal(size=(T, T)) / np.sqrt(T)
S = a @ a.T + .5 * np.eye(T)


def work(which):
    ctx = _lib.default_context()
    if which == 'inv':
        inv = ctx.empty((T, T))
        ld = ctx.dense_inv_logdet(ctx.array(S), T, inv, want_logdet=True)
        return inv.to_host(), ld
    if which == 'psd':
        x = ctx.array(S - 0.6 * np.eye(T))
        ctx.dense_project_psd(x, T, 1e-5)
        return x.to_host(), 0.
    if which == 'klv':
        rt = ctx.ratings(ratings, n, m)
        kl = ctx.mn_kl_value(rt, d, ctx.array(mean), ctx.array(S), ctx.array(np.eye(d) * 1.3), 1., 10., 10.)
        return np.zeros(1), kl
    if which == 'psdfast':
        x = ctx.array(S + 0.25 * (a + a.T) * 1e-3)
        flag = ctx.dense_project_psd(x, T, 1e-5)
        return x.to_host(), float(flag)
    if which == 'small':
        O = np.eye(d) * 1.3 + 0.01 * (a[:d, :d] + a[:d, :d].T)
        inv = ctx.empty((d, d))
        ld = ctx.dense_inv_logdet(ctx.array(O), d, inv, want_logdet=True)
        x = ctx.array(O)
        ctx.dense_project_psd(x, d, 1e-5)
        return np.r_[inv.to_host().ravel(), x.to_host().ravel()], ld
    if which.startswith('fit') and which != 'fit':
        rt = ctx.ratings(ratings, n, m)
        dm, dS, dO = ctx.array(mean), ctx.array(np.eye(T)), ctx.array(np.eye(d))
        tr = ctx.mn_fit_normal(rt, d, dm, dS, dO, 1., 10., 10., 1e-5, 1e-4, None, max_steps=int(which[3:]))
        return np.r_[tr[0], dS.to_host().ravel()], float(tr[0][-1])
    if which == 'grad':
        rt = ctx.ratings(ratings, n, m)
        g = ctx.empty((T, d)), ctx.empty((T, T)), ctx.empty((d, d))
        ctx.mn_kl_gradient(rt, d, ctx.array(mean), ctx.array(S), ctx.array(np.eye(d)), 1., 10., 10., *g)
        return g[1].to_host(), float(g[0].to_host().sum())
    rt = ctx.ratings(ratings, n, m)
    dm, dS, dO = ctx.array(mean), ctx.array(np.eye(T)), ctx.array(np.eye(d))
    tr = ctx.mn_fit_normal(rt, d, dm, dS, dO, 1., 10., 10., 1e-5, 1e-4, None, max_steps=120)
    return dS.to_host(), float(tr[0][-1])


def in_one_thread(which):
    box = []
    t = threading.Thread(target=lambda: box.append(work(which)))
    t.start(); t.join()
    return box[0]


for which in ('fit20', 'fit120'):
    ref = work(which)
    again = work(which)
    fresh = in_one_thread(which)       # a new Context, no concurrency
    print(which, 'main again == main:', bool(np.array_equal(again[0], ref[0]) and again[1] == ref[1]),
          '| fresh context, alone == main:', bool(np.array_equal(fresh[0], ref[0]) and fresh[1] == ref[1]), flush=True)
    results = [None] * 6

    def run(k):
        for _ in range(3):
            results[k] = work(which)
    ts = [threading.Thread(target=run, args=(k,)) for k in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    ok = all(np.array_equal(r[0], ref[0]) and r[1] == ref[1] for r in results)
    worst = max(float(np.abs(r[0] - ref[0]).max()) for r in results)
    print(which, 'threads == single:', ok, 'max abs diff', worst, flush=True)
    if not ok and which.startswith('fit'):
        k = int(which[3:])
        for r in results:
            dif = np.nonzero(r[0][:k] != ref[0][:k])[0]
            if dif.size:
                j = int(dif[0])
                print('   first divergent accepted step', j, repr(ref[0][j]), repr(r[0][j]), 'prev equal:', j == 0 or ref[0][j-1] == r[0][j-1])
            else:
                print('   trace identical; only the covariance differs')
