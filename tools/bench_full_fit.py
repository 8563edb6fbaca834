#!/usr/bin/env python3
"""The full-covariance batch engine (csrc/k8_fullbatch.hip) at BASELINE.json configs[3]'s shape: 94 x 425 rank 5,
K = (n + m) d = 2595 (a 54 MB covariance per state): H hypotheses advanced a few accepted steps; the KL of the
starting point against the single-matrix routine.  Prints JSON."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from active_matrix_factorization_amd import _lib  # noqa: E402


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n, m, d, known = 94, 425, 5, 500
    rs = np.random.RandomState(0)
    flat = rs.choice(n * m, known, replace=False)
    ratings = np.c_[flat // m, flat % m, rs.randint(1, 6, known)].astype(float)
    rated = set(zip((flat // m).tolist(), (flat % m).tolist()))
    free = [(i, j) for i in range(n) for j in range(m) if (i, j) not in rated][:H]
    K = (n + m) * d
    ctx = _lib.Context(0)
    rt = ctx.ratings(ratings, n, m)
    mean = rs.normal(size=K)
    dm = ctx.array(mean)
    dc = ctx.array(np.eye(K))
    out = {'H': H, 'steps': steps, 'K': K}
    t = time.perf_counter()
    ref_kl = ctx.full_kl_value(rt, d, dm, dc, 1., 10., 10.)
    out['single_kl_value_s'] = time.perf_counter() - t
    b = ctx.full_batch(rt, d, H)
    b.set_state(dm, dc)
    b.set_extra([-1] + [f[0] for f in free[1:]], [-1] + [f[1] for f in free[1:]], [0.] + [3.] * (H - 1))
    kl0 = b.kl(1., 10., 10.)
    out['kl0_rel_err_vs_single'] = abs(kl0[0] - ref_kl) / abs(ref_kl)
    ctx.sync()
    t = time.perf_counter()
    lr, kl, st, conv, _ = b.fit(1., 10., 10., 1e-5, 1e-4, None, max_steps=steps)
    dt = time.perf_counter() - t
    out.update(fit_s=dt, accepted=int(st.sum()), ms_per_hypothesis_step=1e3 * dt / max(int(st.sum()), 1),
               kl_first=float(kl0[0]), kl_last=float(kl[0]), stats=b.fit_stats())
    t = time.perf_counter()
    tv = b.total_variance()
    out['total_variance_s'] = time.perf_counter() - t
    out['tv0'] = float(tv[0])
    print(json.dumps(out))


if __name__ == '__main__':
    main()
