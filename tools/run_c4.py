#!/usr/bin/env python3
"""BASELINE.json configs[3] end to end on one MI355X: a 94 x 425 rank-5 matrix with 500 known
ratings (the shape of results/drugbank-94x425), MNActivePMF, the matrix-normal variance criterion
over the full unobserved grid plus one lookahead criterion on a small pool, through the same
compare() loop the reference's Makefile targets drive.  Synthetic integer ratings (the DrugBank
matrix itself does not travel).  Prints timings as JSON."""
import io
import json
import os
import random
import sys
import time
from contextlib import redirect_stdout

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from active_matrix_factorization_amd import mn_active_pmf as mn  # noqa: E402


def main():
    np.random.seed(0)
    random.seed(0)
    n, m, d, known, steps = 94, 425, 5, 500, 2
    u, v = np.random.normal(0, 1, (n, d)), np.random.normal(0, 1, (m, d))
    x = u @ v.T
    real = np.clip(np.round(3 + x / x.std() * 1.2), 1, 5)
    flat = np.random.choice(n * m, known, replace=False)
    ii, jj = flat // m, flat % m
    ratings = np.c_[ii, jj, real[ii, jj]].astype(float)
    out = {'n': n, 'm': m, 'd': d, 'known': known, 'steps': steps}

    def say(msg):
        print(msg, file=sys.stderr, flush=True)

    import threading
    t_start = time.perf_counter()
    stop = threading.Event()

    def heartbeat():            # a long silent phase would otherwise look hung to the job runner
        while not stop.wait(60):
            say('... %.0f s' % (time.perf_counter() - t_start))
    threading.Thread(target=heartbeat, daemon=True).start()

    log = io.StringIO()
    # warm-up on a toy problem: a fresh box pages the HIP / rocSOLVER / rocBLAS code objects in on
    # first use (seconds), which is not what is being measured
    rng_state = np.random.get_state()
    t = time.perf_counter()
    toy = mn.MNActivePMF(np.c_[np.arange(8), np.arange(8), np.arange(8) % 5 + 1.].astype(float), latent_d=2,
                         rating_values=range(1, 6))
    toy.do_fit()
    toy.initialize_approx()
    for _ in zip(range(5), toy.fit_normal_kls()):
        pass
    toy.pick_query_point()
    out['warmup_s'] = time.perf_counter() - t
    np.random.set_state(rng_state)      # the warm-up must not change the problem
    apmf = mn.MNActivePMF(ratings, latent_d=d, rating_values=range(1, 6), discrete_expectations=True)
    t = time.perf_counter()
    apmf.do_fit()
    out['initial_map_fit_s'] = time.perf_counter() - t
    apmf.initialize_approx()
    t = time.perf_counter()
    kls = []
    for kl in apmf.fit_normal_kls():            # the variational fit, step by step
        kls.append(kl)
        if len(kls) % 25 == 0:
            say('fit_normal: %d accepted steps, KL %.3f, %.1f s' % (len(kls), kl, time.perf_counter() - t))
    out['fit_normal_accepted_steps'] = len(kls)
    out['fit_normal_s'] = time.perf_counter() - t
    out['fit_normal_kl_first_last'] = [float(kls[0]), float(kls[-1])]

    # the variance criterion over the full unobserved grid: score + select, then the query loop
    t = time.perf_counter()
    pick = apmf.pick_query_point(key=mn.KEY_FUNCS['pred-variance'])
    out['pred_variance_full_grid_pick_s'] = time.perf_counter() - t
    out['pred_variance_pick'] = list(map(int, pick))
    t = time.perf_counter()
    with redirect_stdout(log):
        res = mn.compare(['pred-variance'], real, ratings, rating_vals=range(1, 6), latent_d=d,
                         steps=steps + 1, discrete_exp=True, do_threading=False, apmf=apmf)
    out['query_loop_%d_steps_s' % steps] = time.perf_counter() - t
    steps_res = res['pred-variance']
    out['rmse_first_last'] = [float(steps_res[0][1]), float(steps_res[-1][1])]
    out['picks'] = [list(map(int, s_[2])) for s_ in steps_res[1:]]
    say('query loop done')

    # one lookahead criterion over a small pool (each candidate: 5 hypothesised ratings, each with a
    # variational refit and a full-grid total-variance sweep): all hypotheses as ONE batch on the
    # device, against the per-hypothesis path of round 1 (deep copies on 4 worker threads)
    pool = sorted(apmf.unrated)[:2]
    key = mn.KEY_FUNCS['total-variance']
    for label, env, procs in (('batched', '1', None), ('threads4', '0', None), ('sequential', '0', 1)):
        if label == 'sequential' and os.environ.get('APMF_C4_SKIP_SEQ'):
            continue
        os.environ['APMF_LOOKAHEAD_BATCH'] = env
        t = time.perf_counter()
        with redirect_stdout(log):
            vals = apmf._get_key_vals(pool, key, procs=procs)
        out['lookahead_2cand_%s_s' % label] = time.perf_counter() - t
        out['lookahead_2cand_%s_vals' % label] = [float(x) for x in vals]
        say('lookahead %s done' % label)
    os.environ['APMF_LOOKAHEAD_BATCH'] = '1'
    # a bigger pool through the batch: hypotheses/s is what a full-grid lookahead costs
    big = int(os.environ.get('APMF_C4_BIG_POOL', '100'))
    pool_big = sorted(apmf.unrated)[:big]
    t = time.perf_counter()
    with redirect_stdout(log):
        vals_big = apmf._get_key_vals(pool_big, key)
    dt = time.perf_counter() - t
    out['lookahead_%dcand_batched_s' % big] = dt
    out['lookahead_hypotheses_per_s'] = 5 * big / dt
    out['lookahead_full_grid_estimate_s'] = len(apmf.unrated) * 5 / (5 * big / dt)
    out['lookahead_%dcand_argmin' % big] = list(map(int, pool_big[int(np.argmin(vals_big))]))
    say('lookahead big pool done')
    out['lookahead_pool'] = [list(map(int, ij)) for ij in pool]
    if '--refit-lookahead' in sys.argv:
        # the reference's own DrugBank setting (results/drugbank-94x425/Makefile:71): every hypothesis re-fits the
        # MAP estimate first (one batch of fit_lls searches) and restarts the approximation from it
        apmf.refit_lookahead = True
        pool_r = sorted(apmf.unrated)[:int(os.environ.get('APMF_C4_REFIT_POOL', '20'))]
        t = time.perf_counter()
        with redirect_stdout(log):
            vals_r = apmf._get_key_vals(pool_r, key)
        dt = time.perf_counter() - t
        out['refit_lookahead'] = {'candidates': len(pool_r), 'hypotheses': 5 * len(pool_r), 'seconds': dt,
                                  'hypotheses_per_s': 5 * len(pool_r) / dt,
                                  'map_refit_prologue_s': mn.LAST_BATCH_STATS.get('refit_s'),
                                  'fit_stats': {k: v for k, v in mn.LAST_BATCH_STATS.items() if k != 'refit_s'},
                                  'argmin': list(map(int, pool_r[int(np.argmin(vals_r))]))}
        apmf.refit_lookahead = False
        say('refit lookahead done')
    stop.set()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
