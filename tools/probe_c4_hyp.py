#!/usr/bin/env python3
"""Dev probe: the two hypothesis re-fits of configs[3] entry (47, 307) step by step (KL after every accepted step,
total variance at the end), for comparison with tests/golden/lookahead_c4_hyp_traces.npz."""
import os, sys
from copy import deepcopy
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from active_matrix_factorization_amd import mn_active_pmf as mn

g = np.load(os.path.join(root, 'tests', 'golden', 'mn_c4_94x425_d5.npz'))
a = mn.MNActivePMF(g['ratings'].copy(), latent_d=5, rating_values=(0., 1.), discrete_expectations=True)
a.users, a.items = g['users'].copy(), g['items'].copy()
a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq = (float(x) for x in g['sigma'])
a.mean, a.cov_useritems, a.cov_latents = g['mean_fit'], g['cov_useritems_fit'], g['cov_latents_fit']
i, j = 47, 307
out = {}
for v in (0., 1.):
    b = deepcopy(a)
    b.add_rating(i, j, v)
    kls = list(b.fit_normal_kls())
    out['kl_trace_%d' % int(v)] = np.array(kls)
    out['total_variance_%d' % int(v)] = b._total_variance()
    print(v, len(kls), repr(kls[-1]), repr(out['total_variance_%d' % int(v)]), flush=True)
os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(root, 'gpurun_out', 'c4_hyp_ours.npz'), **out)
