"""K4 (sample statistics) alone at BASELINE.json configs[2]'s shape: 943 x 1682, d = 15, S = 128.
`python tools/bench_k4.py [S] [d]`; run under rocprofv3 for the kernel time / counters."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from active_matrix_factorization_amd import _lib  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = int(sys.argv[2]) if len(sys.argv) > 2 else 15
n, m = 943, 1682
rs = np.random.RandomState(0)
ctx = _lib.Context(0)
Us, Vs = ctx.array(rs.normal(size=(S, n, d))), ctx.array(rs.normal(size=(S, m, d)))
ld = _lib.grid_ld(m)
gv, gm, gp = ctx.empty((n, ld)), ctx.empty((n, ld)), ctx.empty((n, ld))
ctx.sample_stats(S, n, m, d, ld, Us, Vs, 0., 3., mean=gm, var=gv, prob=gp)
ctx.sync()
t0 = time.perf_counter()
for _ in range(10):
    ctx.sample_stats(S, n, m, d, ld, Us, Vs, 0., 3., mean=gm, var=gv, prob=gp)
ctx.sync()
dt = (time.perf_counter() - t0) / 10
dp = (d + 3) // 4 * 4
print(json.dumps({'S': S, 'd': d, 'ms': dt * 1e3, 'TFLOPs_mfma_padded': 2 * dp * S * n * m / dt / 1e12,
                  'TFLOPs_useful': (2 * d + 5) * S * n * m / dt / 1e12}))
