import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from active_matrix_factorization_amd import _lib
ctx = _lib.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2625
rs = np.random.RandomState(0)
a = rs.normal(size=(N, N)) / np.sqrt(N)
spd = a @ a.T + .5 * np.eye(N)
d = ctx.array(spd); inv = ctx.empty((N, N))
def T(name, f):
    t0 = time.perf_counter(); r = f(); ctx.sync(); print('%-28s %.3f s' % (name, time.perf_counter() - t0), flush=True); return r
if len(sys.argv) > 2:   # launch some of our own kernels first
    x = ctx.array(rs.normal(size=(1000, 10))); y = ctx.empty((1000, 64))
    ctx.pmf_predicted_matrix(x, x, 1000, 1000, 10, 0., ctx.empty((1000, 1024)), 1024); ctx.sync()
    print('own kernel launched first')
T('logdet only (getrf) #1', lambda: ctx.dense_inv_logdet(d, N, None))
T('logdet only (getrf) #2', lambda: ctx.dense_inv_logdet(d, N, None))
T('inverse (getrf+getri) #1', lambda: ctx.dense_inv_logdet(d, N, inv, want_logdet=False))
T('inverse (getrf+getri) #2', lambda: ctx.dense_inv_logdet(d, N, inv, want_logdet=False))
d2 = ctx.array(spd - 0.6 * np.eye(N))
T('project_psd #1', lambda: ctx.dense_project_psd(d2, N, 1e-5))
d2 = ctx.array(spd - 0.6 * np.eye(N))
T('project_psd #2', lambda: ctx.dense_project_psd(d2, N, 1e-5))
