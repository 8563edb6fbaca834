#!/usr/bin/env python3
"""Where does the full-covariance lookahead part from the reference?  Walks fixture
tests/golden/trace_c1_full_lookahead_steps.npz (make_golden.py section V: the reference's own
intermediate quantities of the selection step of section U) and prints, stage by stage, how far
the device path is from it: projection of the initial draw, KL and gradient there, the base
fit_normal trace, and every hypothesis re-fit started from the REFERENCE's base state.
Writes gpurun_out/full_trace.json."""
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from active_matrix_factorization_amd import active_pmf as full  # noqa: E402


def first_part(a, b, tol):
    n = min(len(a), len(b))
    rel = np.abs(np.asarray(a[:n]) - np.asarray(b[:n])) / np.maximum(np.abs(b[:n]), 1e-300)
    bad = np.nonzero(rel > tol)[0]
    return int(bad[0]) if bad.size else None, (float(rel.max()) if n else 0.)


def main():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'trace_c1_full_lookahead_steps.npz'))
    out = {}
    a = full.ActivePMF(g['ratings'].copy(), latent_d=2, rating_values=(1, 2, 3, 4, 5), discrete_expectations=True)
    a.users, a.items = g['users'].copy(), g['items'].copy()
    # 1. the projection of the initial draw
    cov0 = full.project_psd(g['init_draw'], min_eig=a.min_eig)
    out['init_cov_maxabs'] = float(np.abs(cov0 - g['init_cov']).max())
    out['init_cov_scale'] = float(np.abs(g['init_cov']).max())
    # 2. KL and gradient at the reference's initial point
    a.mean, a.cov = g['init_mean'].copy(), g['init_cov'].copy()
    out['init_kl'] = [float(a.kl_divergence()), float(g['init_kl'])]
    gm, gc = full.normal_gradient(a)
    out['init_grad_mean_maxabs'] = float(np.abs(gm - g['init_grad_mean']).max())
    out['init_grad_cov_maxrel'] = float(np.abs(gc - g['init_grad_cov']).max() / np.abs(g['init_grad_cov']).max())
    # 3. base fit from the reference's initial point, and from our own projection
    for name, c0 in (('ref_init', g['init_cov']), ('own_init', cov0)):
        a.mean, a.cov = g['init_mean'].copy(), np.array(c0)
        kls = list(a.fit_normal_kls())
        step, worst = first_part(kls, g['base_kls'], 1e-9)
        out['base_' + name] = dict(steps=len(kls), ref_steps=int(len(g['base_kls'])), first_part_1e9=step,
                                   max_rel=worst, final_kl=kls[-1], ref_final_kl=float(g['base_kls'][-1]),
                                   cov_maxabs=float(np.abs(a.cov - g['base_cov']).max()))
        print(name, out['base_' + name], flush=True)
    # 4. every hypothesis from the REFERENCE's base state
    a.mean, a.cov = g['base_mean'].copy(), g['base_cov'].copy()
    ijv = g['hyp_ijv']
    H = len(ijv)
    steps, tv, kl0, part = np.zeros(H, int), np.zeros(H), np.zeros(H), []
    traced = {tuple(int(x) for x in t) for t in g['traced']}
    for k, (i, j, v) in enumerate(ijv):
        i, j = int(i), int(j)
        b = copy.deepcopy(a)
        b.add_rating(i, j, float(v))
        kl0[k] = b.kl_divergence()
        kls = list(b.fit_normal_kls())
        steps[k], tv[k] = len(kls), b._total_variance()
        if (i, j) in traced:
            ref = g['trace_%d_%d_%d' % (i, j, int(v))]
            s, worst = first_part(kls, ref, 1e-9)
            part.append(dict(ijv=[i, j, float(v)], steps=len(kls), ref_steps=len(ref), first_part_1e9=s, max_rel=worst,
                             tv=tv[k], ref_tv=float(g['hyp_tv'][k])))
            print(part[-1], flush=True)
    out['traced'] = part
    out['hyp_kl0_maxrel'] = float(np.abs(kl0 / g['hyp_kl0'] - 1).max())
    out['hyp_same_steps'] = int((steps == g['hyp_steps']).sum())
    out['hyp_total'] = H
    rel = tv / g['hyp_tv'] - 1
    out['hyp_tv_rel'] = dict(median=float(np.median(np.abs(rel))), max=float(np.abs(rel).max()),
                             below_1e6=int((np.abs(rel) < 1e-6).sum()))
    out['hyp_steps_ours'] = steps.tolist()
    out['hyp_steps_ref'] = g['hyp_steps'].tolist()
    print({k: v for k, v in out.items() if not k.startswith('hyp_steps_')})
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'full_trace.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
