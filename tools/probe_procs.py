#!/usr/bin/env python3
"""Dev probe: which building block of the variational fit changes its result when other
processes keep the GPU busy?  (Distinct result hashes per operation, alone vs concurrent.)"""
import hashlib, os, subprocess, sys
import numpy as np

OPS = ('potrf_inv', 'lu_inv', 'psd_fast', 'psd_clip', 'small', 'grad_mean', 'grad_cov', 'grad_lat', 'klv', 'trial')

if len(sys.argv) > 1 and sys.argv[1] == 'child':
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
    Sns = S + 1e-3 * a                     # not symmetric: LU path
    O = np.eye(d) * 1.3 + 0.01 * (a[:d, :d] + a[:d, :d].T)
    ctx = _lib.default_context()
    rt = ctx.ratings(ratings, n, m)
    h = lambda x: hashlib.md5(np.ascontiguousarray(x).tobytes()).hexdigest()[:8]
    seen = {k: set() for k in OPS}
    for rep in range(int(sys.argv[2])):
        inv = ctx.empty((T, T)); ld = ctx.dense_inv_logdet(ctx.array(S), T, inv, want_logdet=True)
        seen['potrf_inv'].add(h(np.r_[inv.to_host().ravel(), ld]))
        inv = ctx.empty((T, T)); ld = ctx.dense_inv_logdet(ctx.array(Sns), T, inv, want_logdet=True)
        seen['lu_inv'].add(h(np.r_[inv.to_host().ravel(), ld]))
        x = ctx.array(S); ctx.dense_project_psd(x, T, 1e-5); seen['psd_fast'].add(h(x.to_host()))
        x = ctx.array(S - .6 * np.eye(T)); ctx.dense_project_psd(x, T, 1e-5); seen['psd_clip'].add(h(x.to_host()))
        inv = ctx.empty((d, d)); ld = ctx.dense_inv_logdet(ctx.array(O), d, inv, want_logdet=True)
        seen['small'].add(h(np.r_[inv.to_host().ravel(), ld]))
        g = ctx.empty((T, d)), ctx.empty((T, T)), ctx.empty((d, d))
        ctx.mn_kl_gradient(rt, d, ctx.array(mean), ctx.array(S), ctx.array(O), 1., 10., 10., *g)
        seen['grad_mean'].add(h(g[0].to_host())); seen['grad_cov'].add(h(g[1].to_host())); seen['grad_lat'].add(h(g[2].to_host()))
        seen['klv'].add(h(np.array([ctx.mn_kl_value(rt, d, ctx.array(mean), ctx.array(S), ctx.array(O), 1., 10., 10.)])))
        dm, dS, dO = ctx.array(mean), ctx.array(np.eye(T)), ctx.array(np.eye(d))
        tr = ctx.mn_fit_normal(rt, d, dm, dS, dO, 1., 10., 10., 1e-5, 1e-4, None, max_steps=2)
        seen['trial'].add(h(np.r_[tr[0], dS.to_host().ravel(), dm.to_host().ravel(), dO.to_host().ravel()]))
    print(' '.join('%s:%s' % (k, ','.join(sorted(seen[k]))) for k in OPS))
    sys.exit(0)

me = os.path.abspath(__file__)
solo = subprocess.run([sys.executable, me, 'child', '3'], capture_output=True, text=True).stdout.strip()
ref = dict(kv.split(':') for kv in solo.split())
print('solo:', solo)
procs = [subprocess.Popen([sys.executable, me, 'child', '25'], stdout=subprocess.PIPE, text=True) for _ in range(5)]
bad = {k: 0 for k in OPS}
for p in procs:
    got = dict(kv.split(':') for kv in p.communicate()[0].strip().split())
    for k in OPS:
        if got[k] != ref[k]:
            bad[k] += 1
print('processes (of 5) with a result different from solo, per operation:', bad)
