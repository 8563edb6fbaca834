#!/usr/bin/env python3
"""configs[3] pool of fixture W fitted TO CONVERGENCE on the device against the reference's converged values
(tests/golden/lookahead_c4_pool_converged.npz, make_golden.py section W2): per-entry relative differences,
rank agreement.  Writes gpurun_out/c4_converged.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from active_matrix_factorization_amd import mn_active_pmf as mn  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'mn_c4_94x425_d5.npz'))
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'lookahead_c4_pool_converged.npz'))
a = mn.MNActivePMF(g['ratings'].copy(), latent_d=5, rating_values=(0., 1.), discrete_expectations=True)
a.users, a.items = g['users'].copy(), g['items'].copy()
a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq = (float(x) for x in g['sigma'])
a.mean, a.cov_useritems, a.cov_latents = g['mean_fit'], g['cov_useritems_fit'], g['cov_latents_fit']
pool = [tuple(int(x) for x in ij) for ij in z['pool']]
vals = np.array(a._get_key_vals(pool, mn.KEY_FUNCS['total-variance']))
rel = vals / z['vals'] - 1
order = np.argsort(vals, kind='stable')
from scipy import stats
out = dict(vals=vals.tolist(), ref=z['vals'].tolist(), rel=rel.tolist(), worst_rel=float(np.abs(rel).max()),
           median_rel=float(np.median(np.abs(rel))), order=order.tolist(), ref_order=z['order'].tolist(),
           same_argmin=bool(order[0] == z['order'][0]), same_top3=bool(set(order[:3]) == set(z['order'][:3].tolist())),
           kendall_tau=float(stats.kendalltau(vals, z['vals'])[0]), spearman=float(stats.spearmanr(vals, z['vals'])[0]),
           spread_of_pool=float(z['vals'].max() / z['vals'].min() - 1))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'c4_converged.json'), 'w'), indent=1)
print({k: v for k, v in out.items() if k not in ('vals', 'ref', 'rel')})
print(np.round(rel, 6).tolist())
