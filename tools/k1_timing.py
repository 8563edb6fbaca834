"""Per-wave phase stamps of k_fit_iter (needs a build with -DAPMF_K1_TIMING: `make K1_TIMING=1`)."""
import ctypes
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from active_matrix_factorization_amd import _lib, pmf_cy  # noqa: E402

rng = np.random.RandomState(0)
n, m, d, N = 943, 1682, 10, 100000
flat = rng.choice(n * m, N, replace=False)
rat = np.column_stack([flat // m, flat % m, rng.randint(1, 6, N)]).astype(float)
np.random.seed(1)
p = pmf_cy.ProbabilisticMatrixFactorization(rat, latent_d=d, subtract_mean=True)
p.stop_thresh = 0.
lr = float(p.learning_rate)
tr, lr, conv, old = p._fit_on_device(200, lr, None)
lib = _lib.load()
buf = np.zeros(8 * 16384, dtype=np.uint64)
lib.apmf_k1_stamps.argtypes = [ctypes.c_void_p]
assert lib.apmf_k1_stamps(buf.ctypes.data) == 0
st = buf.reshape(-1, 8)[:, :5].astype(np.int64)
st = st[st[:, 0] > 0]
# the last launches are no-ops after max_steps was reached only when the state is done; here
# the final launch evaluated a trial: keep waves with all stamps from the same launch
good = st[(st[:, 4] >= st[:, 3]) & (st[:, 3] >= st[:, 2]) & (st[:, 2] >= st[:, 1]) & (st[:, 1] >= st[:, 0])]
t0 = good[:, 0].min()
rel = (good - t0) * 10.    # 100 MHz -> ns
print('waves', len(good), 'of', len(st))
for i, name in enumerate(['start', 'decided', 'row ready', 'row pass done', 'end']):
    print('%-14s min %7.0f  mean %7.0f  p90 %7.0f  max %7.0f ns' % (name, rel[:, i].min(), rel[:, i].mean(), np.percentile(rel[:, i], 90), rel[:, i].max()))
for i, name in enumerate(['prologue', 'row setup', 'row pass', 'epilogue']):
    dd = rel[:, i + 1] - rel[:, i]
    print('%-14s mean %7.0f  p90 %7.0f  max %7.0f ns' % (name, dd.mean(), np.percentile(dd, 90), dd.max()))
