#!/usr/bin/env python3
"""The eigenvalue bounds that settle a trial's certificate without its factorisation are rigorous: a lookahead pool fitted
to convergence gives the SAME bits with them and with APMF_BATCH_CERT=factor (every certificate by its factorisation).
Runs both in fresh processes on pools of K candidates x 5 rating values at configs[3]'s shape for several seeds and
compares every value.  python tools/check_bound_rigor.py [K] [seeds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, random, io, contextlib
import numpy as np
sys.path.insert(0, %r)
from active_matrix_factorization_amd import mn_active_pmf as mn
K, seed = int(sys.argv[1]), int(sys.argv[2])
np.random.seed(seed); random.seed(seed)
n, m, d, known = 94, 425, 5, 500
u, v = np.random.normal(0, 1, (n, d)), np.random.normal(0, 1, (m, d))
x = u @ v.T
real = np.clip(np.round(3 + x / x.std() * 1.2), 1, 5)
flat = np.random.choice(n * m, known, replace=False)
ratings = np.c_[flat // m, flat %% m, real[flat // m, flat %% m]].astype(float)
with contextlib.redirect_stdout(io.StringIO()):
    a = mn.MNActivePMF(ratings, latent_d=d, rating_values=range(1, 6), discrete_expectations=True)
    a.do_fit(); a.initialize_approx(); a.fit_normal()
    pool = sorted(a.unrated)[seed * 7:seed * 7 + K]
    vals = a._get_key_vals(pool, mn.KEY_FUNCS['total-variance'])
print(json.dumps({'vals': [float(x).hex() for x in vals], 'stats': {k: int(v) for k, v in mn.LAST_BATCH_STATS.items() if k != 'refit_s'}}))
''' % ROOT


def run(K, seed, mode):
    env = dict(os.environ)
    if mode:
        env['APMF_BATCH_CERT'] = mode
    else:
        env.pop('APMF_BATCH_CERT', None)
    out = subprocess.run([sys.executable, '-c', CHILD, str(K), str(seed)], env=env, capture_output=True, text=True, check=True)
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    for seed in range(1, seeds + 1):
        b, f = run(K, seed, None), run(K, seed, 'factor')
        same = b['vals'] == f['vals']
        st = b['stats']
        print('seed %d: %d hypotheses, %d trials, %d certified by a bound (%.0f %%), %d projections - %s'
              % (seed, 5 * K, st['trials'], st['cert_by_bound'], 100. * st['cert_by_bound'] / st['trials'], st['proj_low'] + st['proj_full'],
                 'same bits as with every certificate factored' if same else 'VALUES DIFFER'))
        if not same or f['stats']['trials'] != st['trials']:
            sys.exit(1)


if __name__ == '__main__':
    main()
