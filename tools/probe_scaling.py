#!/usr/bin/env python3
"""Dev probe: do concurrent variational fits overlap on one GPU?  Wall time of K fits in one
process vs the same K fits in each of P concurrent processes / threads."""
import os, subprocess, sys, threading, time
import numpy as np


def setup():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from active_matrix_factorization_amd import _lib
    rs = np.random.RandomState(0)
    n, m, d, N = 94, 425, 5, 500
    flat = rs.choice(n * m, N, replace=False)
    ratings = np.c_[flat // m, flat % m, rs.randint(1, 6, N)].astype(float)
    return _lib, ratings, rs.normal(size=(n + m, d)), n, m, d


def fits(_lib, ratings, mean, n, m, d, K):
    ctx = _lib.default_context()
    rt = ctx.ratings(ratings, n, m)
    T = n + m
    for _ in range(K):
        dm, dS, dO = ctx.array(mean), ctx.array(np.eye(T)), ctx.array(np.eye(d))
        ctx.mn_fit_normal(rt, d, dm, dS, dO, 1., 10., 10., 1e-5, 1e-4, None, max_steps=100)


if len(sys.argv) > 1 and sys.argv[1] == 'child':
    args = setup()
    fits(*args, 1)                       # warm-up
    t = time.perf_counter()
    fits(*args, int(sys.argv[2]))
    print(time.perf_counter() - t)
    sys.exit(0)

me = os.path.abspath(__file__)
K = 6
for P in (() if os.environ.get('THREADS_ONLY') else (1, 2, 4)):
    t = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, me, 'child', str(K)], stdout=subprocess.PIPE, text=True) for _ in range(P)]
    inner = [float(p.communicate()[0].strip()) for p in procs]
    print('processes %d: %d fits each, timed part max %.2f s (wall incl. start-up %.1f s) -> %.1f fits/s'
          % (P, K, max(inner), time.perf_counter() - t, P * K / max(inner)), flush=True)
args = setup()
fits(*args, 1)
for P in (1, 4, 6, 8):
    ths = [threading.Thread(target=fits, args=args + (K,)) for _ in range(P)]
    t = time.perf_counter()
    [x.start() for x in ths]; [x.join() for x in ths]
    dt = time.perf_counter() - t
    print('threads   %d: %d fits each, %.2f s -> %.1f fits/s' % (P, K, dt, P * K / dt), flush=True)
