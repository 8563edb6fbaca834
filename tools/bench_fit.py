"""Time fit_lls at the BASELINE configs[1] shape (943x1682, d=10, N=100000): accepted steps/s and
trials/s.  `python tools/bench_fit.py [steps]`"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from active_matrix_factorization_amd import pmf_cy  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.RandomState(0)
n, m, d, N = 943, 1682, 10, 100000
flat = rng.choice(n * m, N, replace=False)
rat = np.column_stack([flat // m, flat % m, rng.randint(1, 6, N)]).astype(float)
p = pmf_cy.ProbabilisticMatrixFactorization(rat, latent_d=d, subtract_mean=True)
p.stop_thresh = 0.                # never converge by threshold: time a fixed number of steps
lr = float(p.learning_rate)
tr, lr, conv, old = p._fit_on_device(50, lr, None)          # warm
t = time.perf_counter()
tr, lr, conv, old = p._fit_on_device(steps, lr, old)
dt = time.perf_counter() - t
p.learning_rate = lr
print('accepted %d steps in %.3f s: %.1f us/step, lr=%.3g' % (len(tr), dt, dt / max(len(tr), 1) * 1e6, p.learning_rate))
