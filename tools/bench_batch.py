#!/usr/bin/env python3
"""Throughput of the batched variational-fit engine (csrc/k6_batch.hip) at BASELINE.json
configs[3]'s shape (94 x 425 rank 5, 500 known ratings): H hypotheses, each the base model plus one
rating, advanced `steps` accepted steps.  Prints JSON; run under rocprofv3 --kernel-trace --stats
for the per-kernel breakdown."""
import json
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from active_matrix_factorization_amd import _lib  # noqa: E402


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    n, m, d, known = (int(x) for x in os.environ.get('APMF_SHAPE', '94,425,5,500').split(','))
    np.random.seed(0)
    random.seed(0)
    u, v = np.random.normal(0, 1, (n, d)), np.random.normal(0, 1, (m, d))
    x = u @ v.T
    real = np.clip(np.round(3 + x / x.std() * 1.2), 1, 5)
    flat = np.random.choice(n * m, known, replace=False)
    ii, jj = flat // m, flat % m
    ratings = np.c_[ii, jj, real[ii, jj]].astype(float)
    rated = set(zip(ii.tolist(), jj.tolist()))
    cands = [(i, j) for i in range(n) for j in range(m) if (i, j) not in rated][:H]
    N = n + m
    ctx = _lib.Context(0)
    rt = ctx.ratings(ratings, n, m)
    mean = np.vstack([u, v]) + .1 * np.random.normal(size=(N, d))
    out = {'H': H, 'steps': steps, 'n': n, 'm': m, 'd': d}
    for label, hh in (('warm', min(H, 4)), ('run', H)):
        b = ctx.mn_batch(rt, d, hh)
        b.set_state(ctx.array(mean), ctx.array(np.eye(N)), ctx.array(np.eye(d)))
        b.set_extra([c[0] for c in cands[:hh]], [c[1] for c in cands[:hh]],
                    [float(1 + k % 5) for k in range(hh)])
        ctx.sync()
        t = time.perf_counter()
        lr, kl, nsteps, conv, _ = b.fit(1., 10., 10., 1e-5, 1e-4, max_steps=steps)
        dt = time.perf_counter() - t
        if label == 'run':
            out['fit_s'] = dt
            out['accepted_steps_total'] = int(nsteps.sum())
            out['us_per_hypothesis_step'] = dt / max(int(nsteps.sum()), 1) * 1e6
            t = time.perf_counter()
            tv = b.total_variance()
            out['total_variance_sweep_s'] = time.perf_counter() - t
            out['kl_mean'] = float(kl.mean())
            out['tv_mean'] = float(tv.mean())
        b.free()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
