#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REAL reference.

Runs ONLY in the build container (needs /root/reference, Cython, gcc).  The
reference never travels: it is compiled into a scratch directory outside the
repository (default /tmp/apmf_ref_build, deleted afterwards unless --keep) and
only *data* - seeded inputs and the outputs the reference computed for them -
is written under tests/golden/.  No reference source, bytecode or .so is copied
into /root/repo.

Scratch-copy edits (ordinary NumPy-2 / SciPy era breakage):
  * numpy.float/int/complex aliases are restored before import;
  * normal_exps_cy.pyx: np.int_t -> np.intp_t (type no longer in the NumPy-2 pxd);
  * pmf_cy.pyx add_ratings: drop copy=False (NumPy-2 semantics);
  * active_pmf.py kl_divergence: int() cast of the float-valued (i, j) of a rating row;
  * bayes_pmf.py sample_wishart: integer `size`;
  * get_key_evals: index with a tuple, not a list, of index sequences (NumPy >= 1.23).
The arithmetic of every function on the hot path is untouched.

Usage:  python tests/golden/make_golden.py [--keep] [--scratch DIR]
"""
import argparse
import os
import random
import shutil
import subprocess
import sys
import time

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def build_reference(scratch):
    src = os.path.join(REF, 'python-pmf')
    if os.path.exists(scratch):
        shutil.rmtree(scratch)
    os.makedirs(scratch)
    for fn in os.listdir(src):
        if fn.endswith(('.py', '.pyx', '.pxd')):
            shutil.copy(os.path.join(src, fn), scratch)
    shutil.copy(os.path.join(REF, 'generate.py'), scratch)

    def edit(fn, old, new, count=None):
        p = os.path.join(scratch, fn)
        s = open(p).read()
        assert old in s, (fn, old)
        s = s.replace(old, new) if count is None else s.replace(old, new, count)
        open(p, 'w').write(s)

    edit('normal_exps_cy.pyx', 'np.int_t', 'np.intp_t')
    edit('pmf_cy.pyx', 'extra = np.array(extra, copy=False, ndmin=2)',
         'extra = np.array(extra, ndmin=2)')
    edit('active_pmf.py', 'for i, j, rating in self.ratings)',
         'for i, j, rating in ((int(a), int(b), c) for a, b, c in self.ratings))')
    edit('bayes_pmf.py', 'size=(n*(n-1)/2.)', 'size=(n*(n-1)//2)')
    for fn in ('mn_active_pmf.py', 'active_pmf.py'):
        edit(fn, 'evals[list(zip(*pool))]', 'evals[tuple(zip(*pool))]')
    edit('generate.py', 'otypes=[np.float]', 'otypes=[float]')
    edit('generate.py', 'np.array(min_fracs, copy=False)', 'np.asarray(min_fracs)')
    edit('generate.py', 'np.array(max_fracs, copy=False)', 'np.asarray(max_fracs)')

    open(os.path.join(scratch, 'setup_scratch.py'), 'w').write(
        "from setuptools import setup, Extension\n"
        "import numpy as np\n"
        "setup(ext_modules=[Extension(n, [n + '.c'], include_dirs=[np.get_include()],\n"
        "      extra_compile_args=['-O2', '-w'])\n"
        "      for n in ('pmf_cy', 'normal_exps_cy', 'matrix_normal_exps_cy')])\n")
    subprocess.check_call(['cython', '-3', 'pmf_cy.pyx', 'normal_exps_cy.pyx',
                           'matrix_normal_exps_cy.pyx'], cwd=scratch)
    subprocess.check_call([sys.executable, 'setup_scratch.py', '-q', 'build_ext',
                           '--inplace'], cwd=scratch,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scratch', default='/tmp/apmf_ref_build')
    ap.add_argument('--keep', action='store_true')
    ap.add_argument('--only', default='', help="'r2': only the fixtures added in round 2 (sections G-I); 'r2b' .. 'r2n': only section J .. U; 'r3a' .. 'r3i': sections V, W, X, Y, Z, AA, AB, AC, W2 (round 3); 'r4a' .. 'r4f': sections AD, AE, AF, AF2, AG, AH (round 4)")
    args = ap.parse_args()

    build_reference(args.scratch)
    sys.path.insert(0, args.scratch)

    import warnings
    warnings.simplefilter('ignore')
    import numpy as np
    np.float = float
    np.int = int
    np.complex = complex

    import pmf_cy
    import matrix_normal_exps_cy as mnx
    import normal_exps_cy as nx
    import mn_active_pmf
    import active_pmf
    import bayes_pmf
    import generate

    # keep the reference quiet
    import builtins
    _print = builtins.print
    for mod in (mn_active_pmf, active_pmf, bayes_pmf):
        mod.print = lambda *a, **k: None

    def save(name, **arrs):
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **arrs)
        _print('wrote %-28s %7.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))

    def seed(s):
        np.random.seed(s)
        random.seed(s)

    def one_thread(fn, *a):
        # round-3 sections: one BLAS thread in this process too (their worker processes limit themselves), so that
        # the recorded bits do not depend on OMP_NUM_THREADS / the core count of the machine that regenerates them
        from threadpoolctl import threadpool_limits
        with threadpool_limits(1):
            return fn(*a)

    def fake(n, m, d, per_user, noise=.25):
        # same recipe as pmf_cy.fake_ratings (pmf_cy.pyx:461-474)
        return pmf_cy.fake_ratings(noise=noise, num_users=n, num_items=m,
                                   num_ratings=per_user, latent_dimension=d)

    timings = {}
    if args.only == 'r2b':
        round2b(np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r2c':
        round2c(np, pmf_cy, save, seed, _print)
        return
    if args.only == 'r2d':
        round2d(np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r2e':
        round2e(np, bayes_pmf, save, seed, _print)
        return
    if args.only == 'r2f':
        round2f(np, active_pmf, save, seed, _print)
        return
    if args.only == 'r2g':
        round2g(np, bayes_pmf, save, seed, _print)
        return
    if args.only == 'r2h':
        round2h(np, active_pmf, save, seed, _print)
        return
    if args.only == 'r2i':
        round2i(np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r2j':
        round2j(np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r2k':
        round2k(np, bayes_pmf, save, seed, _print)
        return
    if args.only == 'r2l':
        round2l(np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r2m':
        round2m(np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r2n':
        round2n(np, active_pmf, save, seed, _print)
        return
    if args.only == 'r3a':
        round3a(np, active_pmf, save, seed, _print)
        return
    if args.only == 'r3c':
        one_thread(round3c, np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r3d':
        one_thread(round3d, np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r3e':
        one_thread(round3e, np, pmf_cy, save, seed, _print)
        return
    if args.only == 'r3f':
        one_thread(round3f, np, active_pmf, nx, save, seed, fake, _print)
        return
    if args.only == 'r3g':
        one_thread(round3g, np, bayes_pmf, save, seed, fake, _print)
        return
    if args.only == 'r3h':
        one_thread(round3h, np, pmf_cy, save, seed, _print)
        return
    if args.only == 'r3i':
        one_thread(round3i, np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r4a':
        one_thread(round4a, np, active_pmf, nx, save, seed, _print)
        return
    if args.only == 'r4b':
        one_thread(round4b, np, pmf_cy, bayes_pmf, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r4c':
        one_thread(round4c, np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r4d':
        one_thread(round4d, np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r4e':
        one_thread(round4e, np, bayes_pmf, save, seed, _print)
        return
    if args.only == 'r4f':
        one_thread(round4f, np, pmf_cy, save, seed, _print)
        return
    if args.only == 'r3b':
        one_thread(round3b, np, mn_active_pmf, save, seed, _print)
        return
    if args.only == 'r2':
        round2(np, pmf_cy, mn_active_pmf, bayes_pmf, save, seed, fake, _print)
        if not args.keep:
            shutil.rmtree(args.scratch)
        return

    # ------------------------------------------------------------------ A. PMF
    def pmf_case(name, ratings, d, subtract_mean, fit=True, sig=None):
        p = pmf_cy.ProbabilisticMatrixFactorization(
            ratings.copy(), latent_d=d, subtract_mean=subtract_mean)
        if sig is not None:
            p.sigma_sq, p.sigma_u_sq, p.sigma_v_sq = sig
        U0, V0 = p.users.copy(), p.items.copy()
        gU, gV = p.gradient()
        ll = p.log_likelihood()
        half = ratings.shape[0] // 3 + 1
        gUb, gVb = p.gradient(p.ratings[2:2 + half])
        out = dict(ratings=ratings, d=d, subtract_mean=int(subtract_mean),
                   sigma=np.array([p.sigma_sq, p.sigma_u_sq, p.sigma_v_sq]),
                   mean_rating=p.mean_rating,
                   U0=U0, V0=V0, gU=gU, gV=gV, ll=ll, full_ll=p.full_ll(),
                   batch_lo=2, batch_hi=2 + half, gU_batch=gUb, gV_batch=gVb,
                   pred0=p.predicted_matrix())
        if fit:
            trace = np.array(list(p.fit_lls()))
            out.update(ll_trace=trace, U_fit=p.users.copy(), V_fit=p.items.copy())
            real = np.dot(U0, V0.T)  # any dense target will do for rmse()
            out.update(rmse_target=real, rmse=np.float64(p.rmse(real)))
            on = np.zeros(real.shape, bool)
            on[::2, 1::3] = True
            out.update(rmse_on_mask=on, rmse_on=np.float64(p.rmse(real, on)))
            # closed-form variance updates (pmf_cy.pyx:225-255)
            p.update_sigma()
            p.update_sigma_uv()
            out.update(sigma_after=np.array([p.sigma_sq, p.sigma_u_sq, p.sigma_v_sq]))
            # fit_with_sigmas_lls (pmf_cy.pyx:384-403) from the SAME start: LL trace with the
            # variance updates interleaved, first 40 yields
            q = pmf_cy.ProbabilisticMatrixFactorization(ratings.copy(), latent_d=d,
                                                        subtract_mean=subtract_mean)
            q.users, q.items = U0.copy(), V0.copy()
            if sig is not None:
                q.sigma_sq, q.sigma_u_sq, q.sigma_v_sq = sig
            from itertools import islice as _isl
            sig_trace = np.array(list(_isl(q.fit_with_sigmas_lls(), 40)))
            out.update(sigmas_ll_trace=sig_trace, sigmas_U=q.users.copy(), sigmas_V=q.items.copy(),
                       sigmas_final=np.array([q.sigma_sq, q.sigma_u_sq, q.sigma_v_sq]))
        save(name, **out)

    # config 1: generate.py 10x10 rank-2 discrete, known diagonal
    seed(1)
    real_c1 = generate.gen_known_diag_counts(m=10, n=10, rank=2, known_pos=10,
                                             unknown_pos=90, cutoff=0)
    known = generate.known_diag(10, 10)
    ratings_c1 = np.zeros((known.sum(), 3))
    for idx, (i, j) in enumerate(np.transpose(known.nonzero())):
        ratings_c1[idx] = [i, j, real_c1[i, j]]
    seed(2)
    pmf_case('pmf_c1_10x10_d2', ratings_c1, 2, False)

    seed(3)
    r, _, _ = fake(50, 40, 3, 8)
    seed(4)
    pmf_case('pmf_50x40_d3', r, 3, False)
    seed(5)
    pmf_case('pmf_50x40_d3_submean', r, 3, True, sig=(0.7, 4.0, 6.0))

    seed(6)
    r, _, _ = fake(30, 70, 16, 12)
    seed(7)
    pmf_case('pmf_30x70_d16', r, 16, False)

    # headline shape, synthetic (inputs regenerated from the seed by the test:
    # only the outputs are stored).  943 x 1682, d = 10, N = 100 000.
    def c2_inputs():
        seed(0)
        n, m, d, N = 943, 1682, 10, 100000
        Ut = np.random.normal(0, 2, (n, d))
        Vt = np.random.normal(0, 2, (m, d))
        mask = np.zeros((n, m), bool)
        mask[np.arange(n), np.random.randint(0, m, n)] = True
        mask[np.random.randint(0, n, m), np.arange(m)] = True
        free = np.flatnonzero(~mask.ravel())
        extra = np.random.choice(free, N - int(mask.sum()), replace=False)
        mask.ravel()[extra] = True
        ii, jj = mask.nonzero()
        perm = np.random.permutation(ii.size)
        ii, jj = ii[perm], jj[perm]
        vals = (Ut[ii] * Vt[jj]).sum(1) + np.random.normal(0, .25, ii.size)
        ratings = np.column_stack([ii, jj, vals]).astype(float)
        U0 = np.random.random((n, d))
        V0 = np.random.random((m, d))
        return ratings, U0, V0
    ratings, U0, V0 = c2_inputs()
    p = pmf_cy.ProbabilisticMatrixFactorization(ratings.copy(), latent_d=10)
    p.users, p.items = U0.copy(), V0.copy()
    t0 = time.perf_counter(); gU, gV = p.gradient(); t1 = time.perf_counter()
    ll = p.log_likelihood(); t2 = time.perf_counter()
    timings['c2_gradient_s'] = t1 - t0
    timings['c2_loglik_s'] = t2 - t1
    save('pmf_c2_943x1682_d10', ratings_head=ratings[:64], N=ratings.shape[0],
         ratings_sum=ratings.sum(0), U0_sum=U0.sum(), V0_sum=V0.sum(),
         gU=gU, gV=gV, ll=ll)

    # --------------------------------------------------- B. matrix-normal ActivePMF
    def mn_case(name, ratings, d, random_cov, rating_values=None, grad=True):
        a = mn_active_pmf.MNActivePMF(ratings.copy(), latent_d=d,
                                      rating_values=rating_values)
        a.do_fit()
        a.initialize_approx(random_cov=random_cov)
        if random_cov:
            # keep the random covariances well conditioned but non-trivial
            a.cov_useritems /= a.cov_useritems.shape[0]
            a.cov_latents /= d
        n, m = a.num_users, a.num_items
        p_mean, p_var = a.approx_pred_means_vars()
        e2 = np.array([[mnx.exp_dotprod_sq(n, a.mean, a.cov_useritems,
                                           a.cov_latents, i, j)
                        for j in range(m)] for i in range(n)])
        pool = list(a.unrated)
        vals = a._get_key_vals(pool, mn_active_pmf.MNActivePMF.pred_variance, 1, None)
        pick_max = a.pick_query_point(key=mn_active_pmf.MNActivePMF.pred_variance)
        evals = a.get_key_evals(key=mn_active_pmf.MNActivePMF.pred_variance)
        out = dict(ratings=ratings, d=d, users=a.users, items=a.items,
                   mean=a.mean, cov_useritems=a.cov_useritems,
                   cov_latents=a.cov_latents, p_mean=p_mean, p_var=p_var, e2=e2,
                   pool=np.array(pool), pool_vals=np.array(vals),
                   pick_max=np.array(pick_max), key_evals=evals,
                   kl=a.kl_divergence(),
                   total_variance=a._total_variance())
        if grad:
            g_mean, g_cu, g_cl = mnx.matrixnormal_gradient(a)
            out.update(g_mean=g_mean, g_cov_useritems=g_cu, g_cov_latents=g_cl,
                       sigma=np.array([a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq]))
            kls = []
            for k, kl in enumerate(a.fit_normal_kls()):
                kls.append(kl)
                if k >= 2:
                    break
            out.update(kl_trace=np.array(kls), mean_fit=a.mean,
                       cov_useritems_fit=a.cov_useritems,
                       cov_latents_fit=a.cov_latents)
            pm2, pv2 = a.approx_pred_means_vars()
            out.update(p_mean_fit=pm2, p_var_fit=pv2)
        save(name, **out)

    seed(10)
    mn_case('mn_c1_10x10_d2_eye', ratings_c1, 2, False, rating_values=(1, 2, 3, 4, 5))
    seed(11)
    mn_case('mn_c1_10x10_d2_rand', ratings_c1, 2, True, rating_values=(1, 2, 3, 4, 5))
    seed(12)
    r, _, _ = fake(24, 17, 5, 4)
    seed(13)
    mn_case('mn_24x17_d5_rand', r, 5, True)
    seed(14)
    r, _, _ = fake(12, 20, 10, 5)
    seed(15)
    mn_case('mn_12x20_d10_rand', r, 10, True, grad=False)

    # timing of the reference scoring loop (for BASELINE.md; not a fixture)
    seed(16)
    r, _, _ = fake(300, 200, 10, 20)
    a = mn_active_pmf.MNActivePMF(r, latent_d=10)
    a.initialize_approx(random_cov=True)
    pool = list(a.unrated)[:20000]
    t0 = time.perf_counter()
    a._get_key_vals(pool, mn_active_pmf.MNActivePMF.pred_variance, 1, None)
    timings['mn_pred_variance_per_cand_s'] = (time.perf_counter() - t0) / len(pool)

    # ------------------------------------------------- C. full-covariance ActivePMF
    def full_case(name, ratings, d):
        a = active_pmf.ActivePMF(ratings.copy(), latent_d=d)
        a.do_fit()
        a.initialize_approx()
        a.cov /= 8.0
        p_mean, p_var = a.approx_pred_means_vars()
        pool = list(a.unrated)
        vals = a._get_key_vals(pool, active_pmf.ActivePMF.pred_variance, 1, None)
        pick = a.pick_query_point(key=active_pmf.ActivePMF.pred_variance)
        out = dict(ratings=ratings, d=d, users=a.users, items=a.items,
                   mean=a.mean, cov=a.cov, p_mean=p_mean, p_var=p_var,
                   pool=np.array(pool), pool_vals=np.array(vals), pick_max=np.array(pick),
                   kl=a.kl_divergence(), u_idx=a.u, v_idx=a.v,
                   sigma=np.array([a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq]))
        g_mean, g_cov = nx.normal_gradient(a)
        out.update(g_mean=g_mean, g_cov=g_cov)
        kls = []
        for k, kl in enumerate(a.fit_normal_kls()):
            kls.append(kl)
            if k >= 2:
                break
        out.update(kl_trace=np.array(kls), mean_fit=a.mean, cov_fit=a.cov)
        save(name, **out)

    seed(20)
    r, _, _ = fake(6, 5, 3, 2)
    seed(21)
    full_case('full_6x5_d3', r, 3)
    seed(22)
    r, _, _ = fake(12, 9, 5, 3)
    seed(23)
    full_case('full_12x9_d5', r, 5)

    # ------------------------------------------------------------- D. BayesianPMF
    def bayes_case(name, ratings, d, subtract_mean, S, rating_values=None):
        b = bayes_pmf.BayesianPMF(ratings.copy(), latent_d=d,
                                  subtract_mean=subtract_mean,
                                  rating_values=rating_values)
        b.fit()
        users, items = b.users.copy(), b.items.copy()
        out = dict(ratings=ratings, d=d, subtract_mean=int(subtract_mean),
                   users=users, items=items, mean_rating=b.mean_rating)

        # one sample_feature call with every operand recorded
        seed(100)
        mu = np.random.normal(size=d)
        A = np.random.normal(size=(d, d))
        alpha = A.dot(A.T) + d * np.eye(d)
        idx = np.flatnonzero(ratings[:, 0] == 0)
        rated = ratings[idx, 1].astype(int)
        rvals = ratings[idx, 2]
        seed(101)
        z = np.random.normal(0, 1, d)
        seed(101)
        feat = b.sample_feature(0, True, mu, alpha, items, rated, rvals)
        out.update(sf_mu=mu, sf_alpha=alpha, sf_rated=rated, sf_ratings=rvals,
                   sf_z=z, sf_out=feat)

        # hyper-parameter draw
        seed(102)
        hmu, halpha = b.sample_hyperparam(users, True)
        out.update(hp_seed=102, hp_mu=hmu, hp_alpha=halpha)

        # chain prefix: S yields from seed 103
        seed(103)
        from itertools import islice
        samps = list(islice(b.samples(), S))
        out.update(chain_seed=103,
                   chain_U=np.array([u for u, v in samps]),
                   chain_V=np.array([v for u, v in samps]))
        unr = np.array(list(b.unrated)).T
        which = tuple(unr)
        out.update(which=unr,
                   pred_variance=b.pred_variance(samps, which=which),
                   pred_variance_full=b.pred_variance(samps),
                   predict_full=b.predict(samps),
                   prob_ge=b.prob_ge_cutoff(samps, 3.5, which=which),
                   prob_ge_full=b.prob_ge_cutoff(samps, 0.5),
                   total_variance=b.total_variance(samps, which=which),
                   pick_argmax=unr[:, np.argmax(b.pred_variance(samps, which=which))])
        save(name, **out)

    seed(30)
    r, _, _ = fake(20, 15, 4, 5)
    seed(31)
    bayes_case('bayes_20x15_d4', r, 4, True, 8)
    seed(32)
    bayes_case('bayes_c1_10x10_d2', ratings_c1, 2, False, 6,
               rating_values=(1, 2, 3, 4, 5))
    # dof = d + N_rows > 81 + d  -> the chi-square/Bartlett branch of sample_wishart
    seed(33)
    r, _, _ = fake(100, 90, 15, 20)
    seed(34)
    bayes_case('bayes_100x90_d15', r, 15, True, 3)

    # timing: one samples() yield at the headline Gibbs shape (d = 15)
    ratings, _, _ = c2_inputs()
    b = bayes_pmf.BayesianPMF(ratings, latent_d=15)
    it = b.samples()
    seed(35)
    next(it)
    t0 = time.perf_counter(); next(it)
    timings['gibbs_yield_943x1682_d15_s'] = time.perf_counter() - t0

    # --------------------------------------- E. config-1 selection trace (plumbing)
    seed(40)
    res = mn_active_pmf.compare(['random', 'pred-variance'], real_c1, ratings_c1,
                                rating_vals=(1, 2, 3, 4, 5), latent_d=2,
                                do_threading=False, steps=5)
    out = dict(real=real_c1, ratings=ratings_c1, seed=40)
    for key in ('random', 'pred-variance'):
        steps = res[key]
        out[key + '_n_rated'] = np.array([s[0] for s in steps])
        out[key + '_rmse'] = np.array([s[1] for s in steps], dtype=np.float64)
        out[key + '_picks'] = np.array([s[2] for s in steps[1:]])
        for k, s in enumerate(steps[1:]):
            out['%s_vals_%d' % (key, k)] = np.array(s[3], dtype=np.float64)
    init = res['_initial_apmf']
    out.update(init_users=init.users, init_items=init.items, init_mean=init.mean,
               init_cov_useritems=init.cov_useritems,
               init_cov_latents=init.cov_latents)
    save('trace_c1_mn', **out)

    # ------------------------- E2. one-step lookahead criteria on config 1 (mn_active_pmf.py:600-696)
    seed(41)
    a = mn_active_pmf.MNActivePMF(ratings_c1.copy(), latent_d=2, rating_values=(1, 2, 3, 4, 5),
                                  discrete_expectations=True)
    a.do_fit()
    a.initialize_approx()
    a.fit_normal()
    cand = sorted(a.unrated)[7]
    save('lookahead_c1_mn', ratings=ratings_c1, users=a.users, items=a.items, mean=a.mean,
         cov_useritems=a.cov_useritems, cov_latents=a.cov_latents, cand=np.array(cand),
         exp_total_variance=a.exp_total_variance(cand),
         exp_total_variance_byapprox=a.exp_total_variance_byapprox(cand),
         exp_approx_entropy=a.exp_approx_entropy(cand),
         onestep_ge_3_5=a.onestep_ge_3_5(cand),
         prob_ge_3_5=a.prob_ge_3_5(cand), pred=a.pred(cand))

    # ------------------------- F. moments (what test_normal_exps.py pins statistically)
    seed(50)
    dim = 6
    mn_ = np.random.normal(0, 3, dim)
    cov_ = active_pmf.project_psd(np.random.normal(0, 2, (dim, dim)), 1e-5)
    save('moments_full',
         mean=mn_, cov=cov_,
         tripexpect=nx.tripexpect(mn_, cov_, 0, 1, 2),
         quadexpect=nx.quadexpect(mn_, cov_, 0, 1, 2, 3),
         exp_squared=nx.exp_squared(mn_, cov_, 1, 4),
         exp_a2bc=nx.exp_a2bc(mn_, cov_, 2, 3, 5),
         exp_dotprod_sq=nx.exp_dotprod_sq(
             np.arange(3).reshape(1, 3).T, np.arange(3, 6).reshape(1, 3).T,
             mn_, cov_, 0, 0))
    M_ = np.random.normal(0, 2, (5, 3))
    R_ = active_pmf.project_psd(np.random.normal(0, 1, (5, 5)), 1e-5)
    C_ = active_pmf.project_psd(np.random.normal(0, 1, (3, 3)), 1e-5)
    save('moments_mn', mean=M_, cov_rows=R_, cov_cols=C_,
         tripexpect=mnx.tripexpect(M_, R_, C_, 0, 0, 1, 1, 2, 2),
         quadexpect=mnx.quadexpect(M_, R_, C_, 0, 0, 1, 1, 2, 2, 3, 0),
         exp_squared=mnx.exp_squared(M_, R_, C_, 0, 1, 4, 2),
         exp_a2bc=mnx.exp_a2bc(M_, R_, C_, 1, 1, 2, 0, 4, 2),
         exp_dotprod_sq=mnx.exp_dotprod_sq(2, M_, R_, C_, 1, 2))

    round2(np, pmf_cy, mn_active_pmf, bayes_pmf, save, seed, fake, _print)
    round2b(np, mn_active_pmf, save, seed, _print)
    round2c(np, pmf_cy, save, seed, _print)
    round2d(np, mn_active_pmf, save, seed, _print)
    round2e(np, bayes_pmf, save, seed, _print)
    round2f(np, active_pmf, save, seed, _print)
    round2g(np, bayes_pmf, save, seed, _print)
    round2h(np, active_pmf, save, seed, _print)
    round2i(np, mn_active_pmf, save, seed, _print)
    round2j(np, mn_active_pmf, save, seed, _print)
    round2k(np, bayes_pmf, save, seed, _print)
    round2l(np, mn_active_pmf, save, seed, _print)
    round2m(np, mn_active_pmf, save, seed, _print)
    round2n(np, active_pmf, save, seed, _print)
    round3a(np, active_pmf, save, seed, _print)
    one_thread(round3b, np, mn_active_pmf, save, seed, _print)
    one_thread(round3c, np, mn_active_pmf, save, seed, _print)
    one_thread(round3d, np, mn_active_pmf, save, seed, _print)
    one_thread(round3e, np, pmf_cy, save, seed, _print)
    one_thread(round3f, np, active_pmf, nx, save, seed, fake, _print)
    one_thread(round3g, np, bayes_pmf, save, seed, fake, _print)
    one_thread(round3h, np, pmf_cy, save, seed, _print)
    one_thread(round3i, np, mn_active_pmf, save, seed, _print)

    with open(os.path.join(HERE, 'reference_timings.json'), 'w') as f:
        import json
        json.dump(dict(timings, host='build container, 1 thread',
                       note='reference Cython timed by make_golden.py'), f, indent=1)
    _print(timings)

    if not args.keep:
        shutil.rmtree(args.scratch)


def headline_inputs(np, seed):
    """The 943 x 1682, N = 100 000 synthetic ratings of section A (tests/conftest.py:c2_inputs draws the same)."""
    seed(0)
    n, m, d, N = 943, 1682, 10, 100000
    Ut = np.random.normal(0, 2, (n, d))
    Vt = np.random.normal(0, 2, (m, d))
    mask = np.zeros((n, m), bool)
    mask[np.arange(n), np.random.randint(0, m, n)] = True
    mask[np.random.randint(0, n, m), np.arange(m)] = True
    free = np.flatnonzero(~mask.ravel())
    extra = np.random.choice(free, N - int(mask.sum()), replace=False)
    mask.ravel()[extra] = True
    ii, jj = mask.nonzero()
    perm = np.random.permutation(ii.size)
    ii, jj = ii[perm], jj[perm]
    vals = (Ut[ii] * Vt[jj]).sum(1) + np.random.normal(0, .25, ii.size)
    return np.column_stack([ii, jj, vals]).astype(float)


def round2c(np, pmf_cy, save, seed, _print):
    """K. fit_lls at the headline shape (BASELINE.json configs[1]: 943 x 1682, rank 10, N = 100 000), from the
    real reference: the log-likelihood after each of the first 20 accepted steps of the learning-rate search
    from section A's starting point, and the factors there (sampled rows + sums)."""
    from itertools import islice
    t0 = time.perf_counter()
    ratings = headline_inputs(np, seed)
    U0 = np.random.random((943, 10))        # the draws that follow the ratings in section A / conftest.c2_inputs
    V0 = np.random.random((1682, 10))
    p = pmf_cy.ProbabilisticMatrixFactorization(ratings.copy(), latent_d=10)
    p.users, p.items = U0.copy(), V0.copy()
    lls = list(islice(p.fit_lls(), 20))
    _print('headline fit_lls, 20 accepted steps: %.0f s' % (time.perf_counter() - t0))
    rows_u, rows_v = np.array([0, 1, 500, 942]), np.array([0, 7, 1000, 1681])
    save('pmf_c2_fit_943x1682_d10', N=ratings.shape[0], U0_sum=U0.sum(), V0_sum=V0.sum(), ll_trace=np.array(lls),
         rows_u=rows_u, rows_v=rows_v, U_rows=p.users[rows_u], V_rows=p.items[rows_v], U_sum=p.users.sum(),
         V_sum=p.items.sum(), learning_rate=float(p.learning_rate) if hasattr(p, 'learning_rate') else np.nan)


def round2d(np, mn_active_pmf, save, seed, _print):
    """L. The variational fit at the headline shape (943 x 1682, rank 10, N = 100 000; (n+m) = 2625), from the
    real reference: KL at initialize_approx() (mean = the section-A starting factors, identity covariances)
    and after each of the first 3 fit_normal_kls steps, and the state there (mean, cov_latents, the diagonal
    and 5000 sampled entries of the 2625 x 2625 cov_useritems)."""
    from itertools import islice
    MN = mn_active_pmf.MNActivePMF
    t0 = time.perf_counter()
    ratings = headline_inputs(np, seed)
    U0 = np.random.random((943, 10))
    V0 = np.random.random((1682, 10))
    a = MN(ratings.copy(), latent_d=10)
    a.users, a.items = U0.copy(), V0.copy()
    a.initialize_approx()
    kl0 = a.kl_divergence()
    _print('headline MN: KL at start %.6g, %.0f s' % (kl0, time.perf_counter() - t0))
    kls = []
    for kl in islice(a.fit_normal_kls(), 3):
        kls.append(kl)
        _print('headline MN: step %d KL %.6g, %.0f s' % (len(kls), kl, time.perf_counter() - t0))
    rs = np.random.RandomState(82)
    si, sj = rs.randint(0, 2625, 5000), rs.randint(0, 2625, 5000)
    save('mn_c2_fit_943x1682_d10', N=ratings.shape[0], U0_sum=U0.sum(), V0_sum=V0.sum(), kl0=kl0,
         kl_trace3=np.array(kls), mean3=a.mean.copy(), cov_latents3=a.cov_latents.copy(),
         cov_diag3=np.diag(a.cov_useritems).copy(), si=si, sj=sj, cov_s3=a.cov_useritems[si, sj].copy(),
         sigma=np.array([a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq]))


def round2e(np, bayes_pmf, save, seed, _print):
    """M. BayesianPMF.exp_variance (bayes_pmf.py:457-525, 560-598) from the real reference, sequential
    (pool=None) under fixed seeds, on the 10 x 10 rank-2 discrete case of section D: five chain samples from
    seed 5, then the expected total variance of the first four unrated entries from seed 6, with and without
    fit_first, three lookahead samples per hypothesis."""
    from itertools import islice
    g = np.load(os.path.join(HERE, 'bayes_c1_10x10_d2.npz'))
    b = bayes_pmf.BayesianPMF(g['ratings'].copy(), latent_d=2, subtract_mean=False, rating_values=(1, 2, 3, 4, 5))
    b.users, b.items = g['users'].copy(), g['items'].copy()
    seed(5)
    samps = list(islice(b.samples(), 5))
    unr = np.array(sorted(b.unrated)[:4]).T
    which = tuple(unr)
    out = dict(which=unr, chain_U=np.array([u for u, v in samps]), chain_V=np.array([v for u, v in samps]))
    for fit_first in (True, False):
        seed(6)
        ev = b.exp_variance(samps, which=which, fit_first=fit_first, num_samps=3)
        out['exp_variance_fit%d' % int(fit_first)] = np.asarray(ev, float)
        out['rng_after_fit%d' % int(fit_first)] = np.random.get_state()[1][:8].copy()
        _print('exp_variance fit_first=%s: %s' % (fit_first, ev))
    save('bayes_c1_exp_variance', **out)


def round2f(np, active_pmf, save, seed, _print):
    """N. One-step lookahead criteria of the FULL-covariance class (active_pmf.py:477-700) on config 1's
    10 x 10 rank-2 ratings: do_fit, initialize_approx (seeded: it draws a random covariance), fit_normal,
    then the criteria for one candidate - each a deepcopy + add_rating + fit_normal per rating value."""
    g = np.load(os.path.join(HERE, 'lookahead_c1_mn.npz'))
    t0 = time.perf_counter()
    seed(43)
    a = active_pmf.ActivePMF(g['ratings'].copy(), latent_d=2, rating_values=(1, 2, 3, 4, 5),
                             discrete_expectations=True)
    a.do_fit()
    a.initialize_approx()
    a.fit_normal()
    _print('full-cov config 1: fitted, %.0f s' % (time.perf_counter() - t0))
    cand = sorted(a.unrated)[7]
    out = dict(ratings=g['ratings'], users=a.users, items=a.items, mean=a.mean, cov=a.cov, cand=np.array(cand),
               pred=a.pred(cand), pred_variance=a.pred_variance(cand), total_variance=a._total_variance(),
               approx_entropy=a._approx_entropy())
    for key in ('exp_total_variance', 'exp_approx_entropy', 'onestep_ge_3_5'):
        out[key] = getattr(a, key)(cand)
        _print('full-cov config 1: %s = %r, %.0f s' % (key, out[key], time.perf_counter() - t0))
    save('lookahead_c1_full', **out)


def round2g(np, bayes_pmf, save, seed, _print):
    """O. The Bayesian selection loop end to end (bayes_pmf.py:682-826: compare_active -> full_test):
    config 1's 10 x 10 rank-2 discrete matrix, keys random / pred-variance / prob-ge-3.5, 8 chain samples,
    4 steps each, sequential (procs=0) under seed 44: picks, RMSEs and the evaluation grids."""
    g = np.load(os.path.join(HERE, 'trace_c1_mn.npz'))
    real, ratings = g['real'], g['ratings']
    keys = ['random', 'pred-variance', 'prob-ge-3.5']
    seed(44)
    res = bayes_pmf.compare_active(keys, 2, real, ratings.copy(), rating_vals=(1, 2, 3, 4, 5), num_steps=5,
                                   procs=0, num_samps=8, subtract_mean=False)
    out = dict(real=real, ratings=ratings, seed=44, keys=np.array(keys))
    for key in keys:
        steps = res[key]
        out[key + '_n_rated'] = np.array([s[0] for s in steps])
        out[key + '_rmse'] = np.array([s[1] for s in steps], dtype=np.float64)
        out[key + '_picks'] = np.array([s[2] for s in steps[1:]])
        for k, s in enumerate(steps[1:]):
            out['%s_vals_%d' % (key, k)] = np.array(s[3], dtype=np.float64)
        _print('bayes compare %s: picks %s' % (key, out[key + '_picks'].tolist()))
    save('trace_c1_bayes', **out)


def round2h(np, active_pmf, save, seed, _print):
    """P. The full-covariance selection loop end to end (active_pmf.py:933-1090: compare -> full_test) on
    config 1's 10 x 10 rank-2 discrete matrix: keys random / pred-variance / pred, 4 steps each, unthreaded,
    under seed 45 (initialize_approx draws its random covariance from it): picks, RMSEs, evaluation grids
    and the initial approximation."""
    g = np.load(os.path.join(HERE, 'trace_c1_mn.npz'))
    real, ratings = g['real'], g['ratings']
    keys = ['random', 'pred-variance', 'pred']
    seed(45)
    res = active_pmf.compare(keys, latent_d=2, do_threading=False, steps=5,
                             real_ratings_vals=(real, ratings.copy(), (1, 2, 3, 4, 5)))
    out = dict(real=real, ratings=ratings, seed=45, keys=np.array(keys))
    for key in keys:
        steps = res[key]
        out[key + '_n_rated'] = np.array([s[0] for s in steps])
        out[key + '_rmse'] = np.array([s[1] for s in steps], dtype=np.float64)
        out[key + '_picks'] = np.array([s[2] for s in steps[1:]])
        for k, s in enumerate(steps[1:]):
            out['%s_vals_%d' % (key, k)] = np.array(s[3], dtype=np.float64)
        _print('full-cov compare %s: picks %s' % (key, out[key + '_picks'].tolist()))
    init = res['_initial_apmf']
    out.update(init_users=init.users, init_items=init.items, init_mean=init.mean, init_cov=init.cov)
    save('trace_c1_full', **out)


def round2i(np, mn_active_pmf, save, seed, _print):
    """Q. The matrix-normal lookahead criteria with refit_lookahead (mn_active_pmf.py:660-672: every hypothesis
    re-fits the MAP estimate, restarts the approximation from it and fits that) on the state of section E2."""
    g = np.load(os.path.join(HERE, 'lookahead_c1_mn.npz'))
    a = mn_active_pmf.MNActivePMF(g['ratings'].copy(), latent_d=2, rating_values=(1, 2, 3, 4, 5),
                                  discrete_expectations=True, refit_lookahead=True)
    a.users, a.items = g['users'].copy(), g['items'].copy()
    a.mean, a.cov_useritems, a.cov_latents = g['mean'].copy(), g['cov_useritems'].copy(), g['cov_latents'].copy()
    cand = tuple(int(x) for x in g['cand'])
    out = dict(cand=np.array(cand))
    for key in ('exp_total_variance', 'exp_approx_entropy', 'onestep_ge_3_5'):
        seed(46)
        out[key] = getattr(a, key)(cand)
        _print('MN refit lookahead: %s = %r' % (key, out[key]))
    save('lookahead_c1_mn_refit', **out)


def round2j(np, mn_active_pmf, save, seed, _print):
    """R. One selection step of the matrix-normal loop under the LOOKAHEAD criteria (mn_active_pmf.py:660-755):
    compare([...], steps=2) on config 1 - every one of the 90 unrated entries x 5 rating values is a deepcopy +
    add_rating + fit_normal to convergence (+ a full-grid sum / an entropy / a count) in the reference - under
    seed 47: per criterion the pick, the evaluation grid and the RMSEs."""
    g = np.load(os.path.join(HERE, 'trace_c1_mn.npz'))
    real, ratings = g['real'], g['ratings']
    keys = ['total-variance', 'uv-entropy', '1step-ge-3.5']
    t0 = time.perf_counter()
    seed(47)
    res = mn_active_pmf.compare(keys, real, ratings.copy(), rating_vals=(1, 2, 3, 4, 5), latent_d=2,
                                do_threading=False, steps=2, discrete_exp=True)
    init = res['_initial_apmf']
    out = dict(real=real, ratings=ratings, seed=47, keys=np.array(keys), init_users=init.users, init_items=init.items,
               init_mean=init.mean, init_cov_useritems=init.cov_useritems, init_cov_latents=init.cov_latents)
    for key in keys:
        steps = res[key]
        _print('lookahead selection step (%s): pick %s' % (key, steps[1][2]))
        out[key + '_n_rated'] = np.array([s[0] for s in steps])
        out[key + '_rmse'] = np.array([s[1] for s in steps], dtype=np.float64)
        out[key + '_pick'] = np.array(steps[1][2])
        out[key + '_vals'] = np.array(steps[1][3], dtype=np.float64)
    _print('  %.0f s' % (time.perf_counter() - t0))
    save('trace_c1_mn_lookahead', **out)


def round2k(np, bayes_pmf, save, seed, _print):
    """S. One selection step of the Bayesian loop under its LOOKAHEAD criterion 'exp-variance'
    (bayes_pmf.py:457-468,560-605 under compare_active -> full_test): on config 1 every unrated entry x rating
    value is a copy + add_rating + MAP re-fit + a Gibbs chain of 30 samples in the reference, sequentially
    (procs=0) under seed 48: the pick (arg-min), the evaluation grid, the RMSEs."""
    g = np.load(os.path.join(HERE, 'trace_c1_mn.npz'))
    real, ratings = g['real'], g['ratings']
    t0 = time.perf_counter()
    seed(48)
    res = bayes_pmf.compare_active(['exp-variance'], 2, real, ratings.copy(), rating_vals=(1, 2, 3, 4, 5), num_steps=2,
                                   procs=0, num_samps=8, subtract_mean=False)
    steps = res['exp-variance']
    _print('bayes lookahead selection step: pick %s, %.0f s' % (steps[1][2], time.perf_counter() - t0))
    save('trace_c1_bayes_lookahead', real=real, ratings=ratings, seed=48, n_rated=np.array([s[0] for s in steps]),
         rmse=np.array([s[1] for s in steps], dtype=np.float64), pick=np.array(steps[1][2]),
         vals=np.array(steps[1][3], dtype=np.float64))


def round2l(np, mn_active_pmf, save, seed, _print):
    """T. configs[3] (DrugBank 94 x 425 rank 5) under the LOOKAHEAD criterion: pick_query_point over a pool of
    three unrated entries by exp_total_variance (mn_active_pmf.py:660-696, 701-729) from the converged
    approximation of section G - six hypothesis re-fits to convergence, about 5 minutes each in the reference:
    the three criterion values and the pick (the arg-min)."""
    g = np.load(os.path.join(HERE, 'mn_c4_94x425_d5.npz'))
    MN = mn_active_pmf.MNActivePMF
    a = MN(g['ratings'].copy(), latent_d=5, rating_values=(0., 1.), discrete_expectations=True)
    a.users, a.items = g['users'].copy(), g['items'].copy()
    a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq = (float(x) for x in g['sigma'])
    a.mean, a.cov_useritems, a.cov_latents = g['mean_fit'].copy(), g['cov_useritems_fit'].copy(), g['cov_latents_fit'].copy()
    unrated = sorted(a.unrated)
    pool = [unrated[11], unrated[12], unrated[20011]]
    assert tuple(int(x) for x in g['cand']) == tuple(pool[0])
    t0 = time.perf_counter()
    vals = []
    for ij in pool:
        vals.append(a.exp_total_variance(ij))
        _print('c4 pool: %s -> %r, %.0f s' % (ij, vals[-1], time.perf_counter() - t0))
    pick = pool[int(np.argmin(vals))]          # what pick_query_point returns for a criterion whose chooser is min (:109, :729)
    save('lookahead_c4_pool', pool=np.array(pool), vals=np.array(vals, dtype=np.float64), pick=np.array(pick))


def round2m(np, mn_active_pmf, save, seed, _print):
    """T2. The two hypothesis re-fits behind the third value of section T, step by step: for entry (47, 307) of
    configs[3] and each rating value, deepcopy + add_rating + fit_normal_kls to convergence
    (mn_active_pmf.py:660-668): the KL after every accepted step and the total variance at the end."""
    from copy import deepcopy
    g = np.load(os.path.join(HERE, 'mn_c4_94x425_d5.npz'))
    t = np.load(os.path.join(HERE, 'lookahead_c4_pool.npz'))
    MN = mn_active_pmf.MNActivePMF
    a = MN(g['ratings'].copy(), latent_d=5, rating_values=(0., 1.), discrete_expectations=True)
    a.users, a.items = g['users'].copy(), g['items'].copy()
    a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq = (float(x) for x in g['sigma'])
    a.mean, a.cov_useritems, a.cov_latents = g['mean_fit'].copy(), g['cov_useritems_fit'].copy(), g['cov_latents_fit'].copy()
    i, j = (int(x) for x in t['pool'][2])
    out = dict(ij=np.array([i, j]))
    for v in (0., 1.):
        b = deepcopy(a)
        b.add_rating(i, j, v)
        kls = list(b.fit_normal_kls())
        out['kl_trace_%d' % int(v)] = np.array(kls)
        out['total_variance_%d' % int(v)] = b._total_variance()
        _print('c4 hypothesis (%d, %d) = %g: %d steps, KL %r, total variance %r' % (i, j, v, len(kls), kls[-1], out['total_variance_%d' % int(v)]))
    save('lookahead_c4_hyp_traces', **out)


def round2n(np, active_pmf, save, seed, _print):
    """U. One selection step of the full-covariance loop under the LOOKAHEAD criterion 'total-variance'
    (active_pmf.py:560-640 under compare()): config 1, every unrated entry x rating value a deepcopy + add_rating +
    fit_normal to convergence of the (n+m)d x (n+m)d approximation, under seed 49 (initialize_approx draws its
    random covariance from it): the pick, the evaluation grid, the RMSEs."""
    g = np.load(os.path.join(HERE, 'trace_c1_mn.npz'))
    real, ratings = g['real'], g['ratings']
    t0 = time.perf_counter()
    seed(49)
    res = active_pmf.compare(['total-variance'], latent_d=2, do_threading=False, steps=2, discrete_exp=True,
                             real_ratings_vals=(real, ratings.copy(), (1, 2, 3, 4, 5)))
    steps = res['total-variance']
    _print('full-covariance lookahead selection step: pick %s, %.0f s' % (steps[1][2], time.perf_counter() - t0))
    save('trace_c1_full_lookahead', real=real, ratings=ratings, seed=49, n_rated=np.array([s[0] for s in steps]),
         rmse=np.array([s[1] for s in steps], dtype=np.float64), pick=np.array(steps[1][2]),
         vals=np.array(steps[1][3], dtype=np.float64))


def _hyp_trace_full(args):
    """One hypothesis of the full-covariance lookahead, as calculate_fn runs it (active_pmf.py:660-668):
    deepcopy + add_rating + fit_normal_kls to convergence; the KL after every accepted step, the first accepted
    steps' states and the total variance at the end.  Module level so that a process pool can map it."""
    from copy import deepcopy
    import numpy as np
    a, i, j, v, keep, cut = args
    b = deepcopy(a)
    b.add_rating(i, j, v)
    kl0 = b.kl_divergence()
    kls, states, tv_cut = [], [], None
    for kl in b.fit_normal_kls():
        kls.append(kl)
        if len(kls) <= keep:
            states.append(np.concatenate([b.mean, b.cov.reshape(-1)]))
        if len(kls) == cut:
            tv_cut = b._total_variance()
    tv = b._total_variance()
    return kl0, np.array(kls), np.array(states), tv, tv if tv_cut is None else tv_cut


def round3a(np, active_pmf, save, seed, _print):
    """V. Section U step by step.  What compare(['total-variance'], do_threading=False) does under seed 49 up to
    its first pick (active_pmf.py:1012-1090, 795-832), with the intermediate quantities written down: the
    approximation full_test() starts from (its second initialize_approx), its KL, its first gradient, the KL after
    every accepted fit_normal_kls step and the state every 10 steps; then, for every one of the 450 hypotheses of
    the selection step, the number of accepted steps, the final KL and the total variance - and for fifteen of them
    (three candidates x five rating values) the whole KL trace and the first three accepted states."""
    import multiprocessing as mp
    from copy import deepcopy
    g = np.load(os.path.join(HERE, 'trace_c1_mn.npz'))
    u = np.load(os.path.join(HERE, 'trace_c1_full_lookahead.npz'))
    real, ratings = g['real'], g['ratings']
    t0 = time.perf_counter()
    seed(49)
    A = active_pmf.ActivePMF
    apmf = A(ratings.copy(), latent_d=2, rating_values=(1, 2, 3, 4, 5), discrete_expectations=True)
    apmf.sig_u_mean, apmf.sig_u_var, apmf.sig_v_mean, apmf.sig_v_var = 0, -1, 0, -1
    apmf.do_fit()
    apmf.initialize_approx()
    apmf.fit_normal()
    _init = deepcopy(apmf)                 # results['_initial_apmf']
    b = deepcopy(apmf)                     # the copy full_test() works on
    b.do_fit()
    out = dict(users=b.users.copy(), items=b.items.copy(), ratings=ratings)
    rng_state = np.random.get_state()
    b.initialize_approx()
    np.random.set_state(rng_state)
    out['init_draw'] = np.random.normal(0, 2, (b.approx_dim, b.approx_dim))      # what initialize_approx projected
    out['init_mean'], out['init_cov'] = b.mean.copy(), b.cov.copy()
    out['init_kl'] = b.kl_divergence()
    gm, gc = active_pmf.normal_gradient(b)
    out['init_grad_mean'], out['init_grad_cov'] = np.array(gm), np.array(gc)
    kls, snaps = [], []
    for kl in b.fit_normal_kls():
        kls.append(kl)
        if len(kls) <= 3 or len(kls) % 10 == 0:
            snaps.append(np.concatenate([[len(kls)], b.mean, b.cov.reshape(-1)]))
    out['base_kls'], out['base_snaps'] = np.array(kls), np.array(snaps)
    out['base_mean'], out['base_cov'] = b.mean.copy(), b.cov.copy()
    out['base_total_variance'] = b._total_variance()
    out['base_rmse'] = b.rmse(real)
    assert abs(out['base_rmse'] - u['rmse'][0]) < 1e-12, (out['base_rmse'], u['rmse'][0])
    _print('full-cov base fit: %d steps, KL %r, %.0f s' % (len(kls), kls[-1], time.perf_counter() - t0))
    pool = list(b.unrated)                 # the order full_test() evaluates in; section U's values are in it
    vals = b.rating_values
    ref_grid = np.asarray(u['vals'], float)
    pick = tuple(int(x) for x in u['pick'])
    assert ref_grid.shape == (len(pool),) and pool[int(np.argmin(ref_grid))] == pick
    order = np.argsort(ref_grid)
    traced = [pick, pool[int(order[3])], pool[int(order[-1])]]      # the reference's pick, its 4th best, its worst
    cut = 40          # accepted steps after which every hypothesis is ALSO evaluated (see below)
    jobs = [(b, i, j, float(v), 3 if (i, j) in traced else 0, cut) for i, j in pool for v in vals]
    with mp.Pool(7) as workers:
        got = workers.map(_hyp_trace_full, jobs, chunksize=1)
    H = len(jobs)
    out['hyp_ijv'] = np.array([(i, j, v) for _, i, j, v, _, _ in jobs])
    out['hyp_kl0'] = np.array([r[0] for r in got])
    out['hyp_steps'] = np.array([len(r[1]) for r in got])
    out['hyp_final_kl'] = np.array([r[1][-1] if len(r[1]) else np.nan for r in got])
    out['hyp_tv'] = np.array([r[3] for r in got])
    out['traced'] = np.array(traced)
    for (i, j) in traced:
        for v in vals:
            k = jobs.index((b, i, j, float(v), 3, cut))
            out['trace_%d_%d_%d' % (i, j, int(v))] = got[k][1]
            out['states_%d_%d_%d' % (i, j, int(v))] = got[k][2]
    # the criterion values recomputed from these hypotheses must be the grid section U recorded
    import math
    from scipy import stats
    est = np.full(len(pool), np.nan)
    tv = out['hyp_tv'].reshape(len(pool), len(vals))
    for k, (i, j) in enumerate(pool):
        mean = np.dot(b.users[i, :], b.items[j, :])
        cdfs = stats.norm.cdf(b.rating_bounds, loc=mean, scale=math.sqrt(b.sigma_sq))
        est[k] = (tv[k] * np.diff(cdfs)).sum()
    out['pool'], out['vals'] = np.array(pool), est
    # The same selection with every hypothesis stopped after `cut` accepted steps (or at its convergence, if
    # earlier): up to there the searches have not yet amplified rounding differences, so values AND pick are
    # determined - what an implementation with different summation orders can be held to exactly.
    head = np.full((H, 60), np.nan)
    for k, r in enumerate(got):
        head[k, :min(60, len(r[1]))] = r[1][:60]
    out['cut'], out['hyp_kl_head'], out['hyp_tv_cut'] = cut, head, np.array([r[4] for r in got])
    est_cut = np.full(len(pool), np.nan)
    tvc = out['hyp_tv_cut'].reshape(len(pool), len(vals))
    for k, (i, j) in enumerate(pool):
        mean = np.dot(b.users[i, :], b.items[j, :])
        cdfs = stats.norm.cdf(b.rating_bounds, loc=mean, scale=math.sqrt(b.sigma_sq))
        est_cut[k] = (tvc[k] * np.diff(cdfs)).sum()
    out['vals_cut'], out['pick_cut'] = est_cut, np.array(pool[int(np.argmin(est_cut))])
    srt = np.sort(est_cut)
    _print('cut at %d steps: pick %s, margin to the runner-up %.3g (relative %.3g)' % (
        cut, pool[int(np.argmin(est_cut))], srt[1] - srt[0], (srt[1] - srt[0]) / srt[0]))
    dev = np.abs(est - ref_grid).max()
    _print('full-cov hypotheses: %d, recomputed grid vs section U: max |diff| %.3g, %.0f s' % (H, dev, time.perf_counter() - t0))
    assert dev == 0, dev
    save('trace_c1_full_lookahead_steps', **out)


def _hyp_trunc_mn(args):
    """One matrix-normal hypothesis (mn_active_pmf.py:660-668) stopped after `steps` accepted fit_normal_kls
    steps: the KL trace and the total variance there."""
    import itertools
    from copy import deepcopy
    import numpy as np
    from threadpoolctl import threadpool_limits
    a, i, j, v, steps = args
    with threadpool_limits(1):      # 7 workers on 8 cores: one BLAS thread each
        b = deepcopy(a)
        b.add_rating(i, j, v)
        kls = np.array(list(itertools.islice(b.fit_normal_kls(), steps)))
        return kls, b._total_variance()


def round3b(np, mn_active_pmf, save, seed, _print):
    """W. configs[3] (DrugBank 94 x 425 rank 5): the RANKING of a realistic pool under the lookahead criterion.
    24 unrated entries x 2 rating values from the converged approximation of section G, every hypothesis
    (deepcopy + add_rating + fit_normal_kls, mn_active_pmf.py:660-668) advanced by at most 100 accepted steps -
    where the search has not yet amplified rounding differences; `lens` holds the count where it converged earlier - then _total_variance and the expectation over the
    rating values exactly as _exp_with_rij forms it (:676-681): 48 KL traces, 48 totals, 24 criterion values."""
    import math
    import multiprocessing as mp
    from scipy import stats
    g = np.load(os.path.join(HERE, 'mn_c4_94x425_d5.npz'))
    t = np.load(os.path.join(HERE, 'lookahead_c4_pool.npz'))
    MN = mn_active_pmf.MNActivePMF
    a = MN(g['ratings'].copy(), latent_d=5, rating_values=(0., 1.), discrete_expectations=True)
    a.users, a.items = g['users'].copy(), g['items'].copy()
    a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq = (float(x) for x in g['sigma'])
    a.mean, a.cov_useritems, a.cov_latents = g['mean_fit'].copy(), g['cov_useritems_fit'].copy(), g['cov_latents_fit'].copy()
    unrated = sorted(a.unrated)
    idx = sorted(set([11, 12, 20011] + [int(x) for x in np.linspace(100, len(unrated) - 100, 21)]))
    pool = [unrated[k] for k in idx]
    assert [tuple(int(x) for x in r) for r in t['pool']] == [unrated[11], unrated[12], unrated[20011]]
    steps = 100
    t0 = time.perf_counter()
    jobs = [(a, i, j, v, steps) for i, j in pool for v in a.rating_values]
    got = []
    with mp.Pool(7) as workers:
        for r in workers.imap(_hyp_trunc_mn, jobs, chunksize=1):
            got.append(r)
            _print('  hypothesis %d/%d, %.0f s' % (len(got), len(jobs), time.perf_counter() - t0))
    _print('c4 truncated pool: %d hypotheses x %d steps, %.0f s' % (len(jobs), steps, time.perf_counter() - t0))
    lens = np.array([len(r[0]) for r in got])      # a hypothesis may converge before the cut
    kls = np.full((len(got), steps), np.nan)
    for k, r in enumerate(got):
        kls[k, :lens[k]] = r[0]
    _print('   accepted steps: %s' % lens.tolist())
    tv = np.array([r[1] for r in got]).reshape(len(pool), 2)
    est = []
    for (i, j), evals in zip(pool, tv):
        mean = np.dot(a.users[i, :], a.items[j, :])
        cdfs = stats.norm.cdf(a.rating_bounds, loc=mean, scale=math.sqrt(a.sigma_sq))
        est.append((evals * np.diff(cdfs)).sum())
    save('lookahead_c4_pool_ranked', pool=np.array(pool), steps=steps, lens=lens, kls=kls,
         tv=tv, vals=np.array(est), order=np.argsort(est, kind='stable'))


def _hyp_refit_mn(args):
    """One matrix-normal hypothesis with refit_lookahead (mn_active_pmf.py:660-668): the MAP re-fit, the restarted
    approximation's KL, `steps` accepted fit_normal_kls steps and the total variance there."""
    import itertools
    from copy import deepcopy
    import numpy as np
    from threadpoolctl import threadpool_limits
    a, i, j, v, steps = args
    with threadpool_limits(1):
        b = deepcopy(a)
        b.add_rating(i, j, v)
        lls = [ll for ll in b.fit_lls()]
        b.initialize_approx()
        kl0 = b.kl_divergence()
        kls = np.array(list(itertools.islice(b.fit_normal_kls(), steps)))
        return b.users.copy(), b.items.copy(), np.array(lls), kl0, kls, b._total_variance()


def round3c(np, mn_active_pmf, save, seed, _print):
    """X. configs[3] as the reference's own Makefile runs it (results/drugbank-94x425/Makefile: LATENT_D = 20,
    --refit-lookahead): from the MAP estimate of a seeded rank-20 model on the DrugBank subset, every hypothesis of
    an 8-entry pool x 2 rating values re-fits the MAP estimate (fit_lls to convergence, pmf_cy.pyx:298-301),
    restarts the approximation from it (initialize_approx, :202-219) and advances it by 60 accepted fit_normal_kls
    steps; then _total_variance and the expectation over the rating values (:676-681).  Stored: the base MAP state,
    per hypothesis the log-likelihood trace of the re-fit, the KL of the restarted approximation (a function of the
    whole re-fitted MAP state), the KL trace, the total; the criterion values and their ranking; the re-fitted MAP
    state itself for the first two hypotheses."""
    import math
    import multiprocessing as mp
    from scipy import stats
    g = np.load(os.path.join(HERE, 'mn_c4_94x425_d5.npz'))
    seed(11)
    a = mn_active_pmf.MNActivePMF(g['ratings'].copy(), latent_d=20, rating_values=(0., 1.), discrete_expectations=True,
                                  refit_lookahead=True)
    a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq = (float(x) for x in g['sigma'])
    users0, items0 = a.users.copy(), a.items.copy()
    base_lls = np.array([ll for ll in a.fit_lls()])
    a.initialize_approx()
    _print('c4 d=20 base MAP: %d fit_lls values, ll %.6f, KL of the restarted approximation %.6f'
           % (len(base_lls), base_lls[-1], a.kl_divergence()))
    unrated = sorted(a.unrated)
    idx = [11, 12, 20011] + [int(x) for x in np.linspace(3000, len(unrated) - 3000, 5)]
    pool = [unrated[k] for k in idx]
    steps = 60
    t0 = time.perf_counter()
    jobs = [(a, i, j, v, steps) for i, j in pool for v in a.rating_values]
    got = []
    with mp.Pool(7) as workers:
        for r in workers.imap(_hyp_refit_mn, jobs, chunksize=1):
            got.append(r)
            _print('  hypothesis %d/%d: %d re-fit values, %d steps, %.0f s' % (len(got), len(jobs), len(r[2]), len(r[4]), time.perf_counter() - t0))
    lens = np.array([len(r[4]) for r in got])
    kls = np.full((len(got), steps), np.nan)
    for k, r in enumerate(got):
        kls[k, :lens[k]] = r[4]
    ll_lens = np.array([len(r[2]) for r in got])
    lls = np.full((len(got), ll_lens.max()), np.nan)
    for k, r in enumerate(got):
        lls[k, :ll_lens[k]] = r[2]
    tv = np.array([r[5] for r in got]).reshape(len(pool), 2)
    est = []
    for (i, j), evals in zip(pool, tv):
        mean = np.dot(a.users[i, :], a.items[j, :])
        cdfs = stats.norm.cdf(a.rating_bounds, loc=mean, scale=math.sqrt(a.sigma_sq))
        est.append((evals * np.diff(cdfs)).sum())
    save('lookahead_c4_d20_refit', users0=users0, items0=items0, base_lls=base_lls, users=a.users, items=a.items,
         pool=np.array(pool), steps=steps, lens=lens, kls=kls, ll_lens=ll_lens, lls=lls,
         kl0=np.array([r[3] for r in got]), tv=tv, vals=np.array(est), order=np.argsort(est, kind='stable'),
         map_users=np.array([got[0][0], got[1][0]]), map_items=np.array([got[0][1], got[1][1]]))


def _continuous_exp(args):
    """E[total variance | R_ij] by the reference's CONTINUOUS integration (mn_active_pmf.py:683-691:
    stats.norm.expect over mean +- 2 sd, epsrel .02), recording every integrand evaluation."""
    import warnings
    import numpy as np
    from threadpoolctl import threadpool_limits
    a, ij, use_map = args
    seen = []

    def fn(apmf, v):
        t = apmf._total_variance()
        seen.append((v, t))
        return t
    with threadpool_limits(1), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        est = a._exp_with_rij(ij, fn, use_map=use_map, pass_v=True)
    return est, np.array(seen)


def round3d(np, mn_active_pmf, save, seed, _print):
    """Y. The CONTINUOUS lookahead (rating_values=None - the class default): exp_total_variance and
    exp_total_variance_byapprox of two candidates on the state of section E2 (config 1), with every point the
    adaptive quadrature asked for and the integrand's value there (a fit_normal to convergence per point)."""
    import multiprocessing as mp
    g = np.load(os.path.join(HERE, 'lookahead_c1_mn.npz'))
    a = mn_active_pmf.MNActivePMF(g['ratings'].copy(), latent_d=2)
    a.users, a.items = g['users'].copy(), g['items'].copy()
    a.mean, a.cov_useritems, a.cov_latents = g['mean'].copy(), g['cov_useritems'].copy(), g['cov_latents'].copy()
    pool = sorted(a.unrated)[:2]
    jobs = [(a, ij, um) for um in (True, False) for ij in pool]
    t0 = time.perf_counter()
    with mp.Pool(4) as workers:
        got = workers.map(_continuous_exp, jobs, chunksize=1)
    _print('continuous lookahead: %d integrals, %s evaluations, %.0f s' % (len(got), [len(r[1]) for r in got], time.perf_counter() - t0))
    out = dict(pool=np.array(pool), est_map=np.array([got[0][0], got[1][0]]), est_approx=np.array([got[2][0], got[3][0]]))
    for k, name in enumerate(('map0', 'map1', 'approx0', 'approx1')):
        out['evals_' + name] = got[k][1]
    save('lookahead_c1_mn_continuous', **out)


def round3e(np, pmf_cy, save, seed, _print):
    """Z. The minibatch fit (pmf_cy.pyx:308-381) at the HEADLINE shape - 943 x 1682, rank 10, N = 100 000 (section
    H holds it at 50 x 40 only): two epochs of fit_minibatches(1000) - 100 minibatches each, momentum carried across -
    and three (train, validation) pairs of fit_minibatches_validation(1000, 5000) under fixed seeds.  Stored: the
    training errors, sampled rows and the sums of the fitted factors, the head of the shuffled ratings array."""
    from itertools import islice
    ratings = headline_inputs(np, seed)
    t0 = time.perf_counter()
    seed(90)
    p = pmf_cy.ProbabilisticMatrixFactorization(ratings.copy(), latent_d=10)
    U0, V0 = p.users.copy(), p.items.copy()
    seed(91)
    errs = list(islice(p.fit_minibatches(1000), 2))
    ru, rv = np.array([0, 1, 470, 942]), np.array([0, 1, 840, 1681])
    out = dict(seed_ctor=90, seed_fit=91, batch_size=1000, U0_sum=U0.sum(), V0_sum=V0.sum(),
               train_errs=np.array(errs, np.float64), rows_u=ru, rows_v=rv, U_rows=p.users[ru], V_rows=p.items[rv],
               U_sum=p.users.sum(), V_sum=p.items.sum(), U_abs_sum=np.abs(p.users).sum(), V_abs_sum=np.abs(p.items).sum(),
               ratings_after_head=p.ratings[:64].copy(), ratings_after_weighted=(p.ratings * np.arange(1, 100001)[:, None]).sum(0))
    seed(92)
    q = pmf_cy.ProbabilisticMatrixFactorization(ratings.copy(), latent_d=10)
    q.users, q.items = U0.copy(), V0.copy()
    seed(93)
    pairs = list(islice(q.fit_minibatches_validation(1000, 5000), 3))
    out.update(seed_pairs=93, valid_size=5000, valid_pairs=np.array(pairs, np.float64), Uv_rows=q.users[ru], Vv_rows=q.items[rv])
    _print('headline minibatch: train errs %r, pairs %r, %.0f s' % (errs, pairs, time.perf_counter() - t0))
    save('pmf_minibatch_c2_943x1682_d10', **out)


def round3f(np, active_pmf, nx, save, seed, fake, _print):
    """AA. The full-covariance class at ranks 10 and 16 (section C holds ranks 3 and 5): the scoring grids of
    approx_pred_means_vars (normal_exps_cy.exp_dotprod_sq, normal_exps_cy.pyx:111-135 - on the device the sweep for
    9 <= d <= 16, the headline rank among them), the pool values and the pick, the KL, its gradient and three accepted
    fit_normal_kls steps - exactly what section C records."""
    def full_case(name, ratings, d):
        a = active_pmf.ActivePMF(ratings.copy(), latent_d=d)
        a.do_fit()
        a.initialize_approx()
        a.cov /= 8.0
        p_mean, p_var = a.approx_pred_means_vars()
        pool = list(a.unrated)
        vals = a._get_key_vals(pool, active_pmf.ActivePMF.pred_variance, 1, None)
        pick = a.pick_query_point(key=active_pmf.ActivePMF.pred_variance)
        out = dict(ratings=ratings, d=d, users=a.users, items=a.items,
                   mean=a.mean, cov=a.cov, p_mean=p_mean, p_var=p_var,
                   pool=np.array(pool), pool_vals=np.array(vals), pick_max=np.array(pick),
                   kl=a.kl_divergence(), u_idx=a.u, v_idx=a.v,
                   sigma=np.array([a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq]))
        g_mean, g_cov = nx.normal_gradient(a)
        out.update(g_mean=g_mean, g_cov=g_cov)
        kls = []
        for k, kl in enumerate(a.fit_normal_kls()):
            kls.append(kl)
            if k >= 2:
                break
        out.update(kl_trace=np.array(kls), mean_fit=a.mean, cov_fit=a.cov)
        save(name, **out)

    t0 = time.perf_counter()
    seed(24)
    r, _, _ = fake(11, 14, 10, 4)
    seed(25)
    full_case('full_11x14_d10', r, 10)
    seed(26)
    r, _, _ = fake(7, 9, 16, 3)
    seed(27)
    full_case('full_7x9_d16', r, 16)
    _print('full-covariance ranks 10 and 16: %.0f s' % (time.perf_counter() - t0))


def round3g(np, bayes_pmf, save, seed, fake, _print):
    """AB. BayesianPMF.exp_variance (bayes_pmf.py:457-525, 560-598) at 100 x 90, rank 15, subtract_mean (section M
    holds the 10 x 10 rank-2 case): fake ratings (pmf_cy.fake_ratings, seed 9) rounded to 1 ... 5, four chain samples
    from seed 7, then the expected total variance of the first three unrated entries from seed 8, with and without
    fit_first, three lookahead samples per hypothesis; sequential (pool=None); expectation over the five rating
    values - the branch this SciPy still runs: the continuous one calls scipy.integrate.trapz (bayes_pmf.py:585),
    removed from SciPy since."""
    from itertools import islice
    t0 = time.perf_counter()
    seed(9)
    r, _, _ = fake(100, 90, 15, 20)
    r[:, 2] = np.clip(np.round(3 + r[:, 2] / r[:, 2].std() * 1.2), 1, 5)
    seed(10)
    b = bayes_pmf.BayesianPMF(r.copy(), latent_d=15, subtract_mean=True, rating_values=(1, 2, 3, 4, 5),
                              discrete_expectations=True)
    U0, V0 = b.users.copy(), b.items.copy()
    seed(7)
    samps = list(islice(b.samples(), 4))
    unr = np.array(sorted(b.unrated)[:3]).T
    which = tuple(unr)
    out = dict(ratings=r, U0=U0, V0=V0, which=unr, chain_U=np.array([u for u, v in samps]),
               chain_V=np.array([v for u, v in samps]))
    for fit_first in (True, False):
        seed(8)
        ev = b.exp_variance(samps, which=which, fit_first=fit_first, num_samps=3)
        out['exp_variance_fit%d' % int(fit_first)] = np.asarray(ev, float)
        out['rng_after_fit%d' % int(fit_first)] = np.random.get_state()[1][:8].copy()
        _print('exp_variance 100x90 fit_first=%s: %s, %.0f s' % (fit_first, ev, time.perf_counter() - t0))
    save('bayes_100x90_exp_variance', **out)


def round3h(np, pmf_cy, save, seed, _print):
    """AC. fit_with_sigmas_lls and the closed-form variance updates (pmf_cy.pyx:225-255, 384-403) at the HEADLINE
    shape - 943 x 1682, rank 10, N = 100 000 (section A holds them up to 50 x 40): update_sigma / update_sigma_uv at
    the seeded start, then the first 12 yields of fit_with_sigmas_lls - the variances re-estimated between steps -
    with sampled rows of the factors and the variances it ends with."""
    from itertools import islice
    ratings = headline_inputs(np, seed)
    t0 = time.perf_counter()
    seed(94)
    p = pmf_cy.ProbabilisticMatrixFactorization(ratings.copy(), latent_d=10)
    U0, V0 = p.users.copy(), p.items.copy()
    p.update_sigma()
    p.update_sigma_uv()
    out = dict(seed_ctor=94, U0_sum=U0.sum(), V0_sum=V0.sum(),
               sigma_after=np.array([p.sigma_sq, p.sigma_u_sq, p.sigma_v_sq]))
    q = pmf_cy.ProbabilisticMatrixFactorization(ratings.copy(), latent_d=10)
    q.users, q.items = U0.copy(), V0.copy()
    trace = np.array(list(islice(q.fit_with_sigmas_lls(), 12)))
    ru, rv = np.array([0, 1, 470, 942]), np.array([0, 1, 840, 1681])
    out.update(sigmas_ll_trace=trace, rows_u=ru, rows_v=rv, U_rows=q.users[ru], V_rows=q.items[rv],
               U_sum=q.users.sum(), V_sum=q.items.sum(),
               sigmas_final=np.array([q.sigma_sq, q.sigma_u_sq, q.sigma_v_sq]))
    _print('headline fit_with_sigmas_lls: %r ... %r, sigmas %r, %.0f s' % (trace[0], trace[-1], out['sigmas_final'], time.perf_counter() - t0))
    save('pmf_sigmas_c2_943x1682_d10', **out)


def round3i(np, mn_active_pmf, save, seed, _print):
    """W2. The pool of section W with every hypothesis fitted TO CONVERGENCE, as the reference's lookahead runs it
    (mn_active_pmf.py:660-681): 48 stopped searches of 250 ... 700 accepted steps each (an hour of single-threaded
    reference time over seven workers).  The searches are chaotic past ~100 steps (section T2), so a device run
    cannot reproduce these values to rounding; the fixture answers the question that leaves: how far do the
    CONVERGED criterion values and their ranking move."""
    import math
    import multiprocessing as mp
    from scipy import stats
    g = np.load(os.path.join(HERE, 'mn_c4_94x425_d5.npz'))
    w = np.load(os.path.join(HERE, 'lookahead_c4_pool_ranked.npz'))
    MN = mn_active_pmf.MNActivePMF
    a = MN(g['ratings'].copy(), latent_d=5, rating_values=(0., 1.), discrete_expectations=True)
    a.users, a.items = g['users'].copy(), g['items'].copy()
    a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq = (float(x) for x in g['sigma'])
    a.mean, a.cov_useritems, a.cov_latents = g['mean_fit'].copy(), g['cov_useritems_fit'].copy(), g['cov_latents_fit'].copy()
    pool = [tuple(int(x) for x in ij) for ij in w['pool']]
    t0 = time.perf_counter()
    jobs = [(a, i, j, v, None) for i, j in pool for v in a.rating_values]
    got = []
    with mp.Pool(7) as workers:
        for r in workers.imap(_hyp_trunc_mn, jobs, chunksize=1):
            got.append(r)
            _print('  hypothesis %d/%d: %d steps, %.0f s' % (len(got), len(jobs), len(r[0]), time.perf_counter() - t0))
    steps = np.array([len(r[0]) for r in got])
    tv = np.array([r[1] for r in got]).reshape(len(pool), 2)
    est = []
    for (i, j), evals in zip(pool, tv):
        mean = np.dot(a.users[i, :], a.items[j, :])
        cdfs = stats.norm.cdf(a.rating_bounds, loc=mean, scale=math.sqrt(a.sigma_sq))
        est.append((evals * np.diff(cdfs)).sum())
    save('lookahead_c4_pool_converged', pool=np.array(pool), steps=steps, final_kl=np.array([r[0][-1] for r in got]),
         tv=tv, vals=np.array(est), order=np.argsort(est, kind='stable'))


def round4a(np, active_pmf, nx, save, seed, _print):
    """AD. The FULL-covariance class at BASELINE.json configs[3]'s shape (DrugBank 94 x 425, rank 5: the approximation
    is N(mean, cov) over K = (94 + 425) 5 = 2595 numbers; sections C / AA hold it at K <= 256 only): the known entries of
    section G, do_fit, initialize_approx under a seed (project_psd of a K x K normal draw: 54 MB, so only its diagonal
    and 4000 sampled entries are kept - the test regenerates the draw from the seed), the KL there, the scoring grid of
    approx_pred_means_vars over all 39 950 entries and the pick over the unrated ones (normal_exps_cy.exp_dotprod_sq,
    normal_exps_cy.pyx:111-135), the KL gradient (normal_exps_cy.pyx:140-303: mean part whole, covariance part
    diagonal + samples) and two accepted fit_normal_kls steps (active_pmf.py:250-288)."""
    t0 = time.perf_counter()
    g = np.load(os.path.join(HERE, 'mn_c4_94x425_d5.npz'))
    ratings = g['ratings']
    A = active_pmf.ActivePMF
    seed(81)
    a = A(ratings.copy(), latent_d=5)
    a.do_fit()
    seed(82)
    a.initialize_approx()
    K = a.approx_dim
    rs = np.random.RandomState(83)
    si, sj = rs.randint(0, K, 4000), rs.randint(0, K, 4000)
    out = dict(ratings=ratings, d=5, users=a.users.copy(), items=a.items.copy(), seed_fit=81, seed_init=82, K=K,
               sigma=np.array([a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq]), samp_i=si, samp_j=sj,
               mean0=a.mean.copy(), cov_diag0=np.diag(a.cov).copy(), cov_samp0=a.cov[si, sj].copy(),
               cov_min_eig0=float(np.linalg.eigvalsh(a.cov).min()))
    _print('full c4: initialised, %.0f s' % (time.perf_counter() - t0))
    out['kl0'] = a.kl_divergence()
    pm, pv = a.approx_pred_means_vars()
    out.update(p_mean0=pm, p_var0=pv, pick0=np.array(a.pick_query_point(key=A.pred_variance)),
               total_variance0=a._total_variance())
    _print('full c4: scored, %.0f s' % (time.perf_counter() - t0))
    g_mean, g_cov = nx.normal_gradient(a)
    out.update(g_mean=g_mean, g_cov_diag=np.diag(g_cov).copy(), g_cov_samp=g_cov[si, sj].copy())
    _print('full c4: gradient, %.0f s' % (time.perf_counter() - t0))
    kls = []
    for k, kl in enumerate(a.fit_normal_kls()):
        kls.append(kl)
        _print('full c4: step %d kl %.6f, %.0f s' % (k, kl, time.perf_counter() - t0))
        if k >= 1:
            break
    pm, pv = a.approx_pred_means_vars()
    out.update(kl_trace=np.array(kls), mean_fit=a.mean.copy(), cov_diag_fit=np.diag(a.cov).copy(),
               cov_samp_fit=a.cov[si, sj].copy(), p_mean_fit=pm, p_var_fit=pv,
               pick_fit=np.array(a.pick_query_point(key=A.pred_variance)))
    save('full_c4_94x425_d5', **out)
    _print('full covariance at configs[3]: %.0f s' % (time.perf_counter() - t0))


def round4b(np, pmf_cy, bayes_pmf, mn_active_pmf, save, seed, _print):
    """AE. BASELINE.json configs[1] and configs[2] on the data set they NAME: the reference's own
    movielens-100k/ratings_matrix.npy.gz (943 x 1682 int8, 100 000 ratings; sections K, Z, AC and I use a synthetic matrix
    of that shape).  The matrix is kept as data (`ml100k_real`, 124 KB); ratings = its non-zeros in row-major order.
    PMF, rank 10, subtract_mean: the constructor's factors under seed 91, gradient and log-likelihood there, the first 20
    accepted fit_lls steps (pmf_cy.pyx:257-291), the closed-form variance updates.  BayesianPMF, rank 15, subtract_mean: two
    seeded samples() yields (bayes_pmf.py:227-302), pred_variance / predict over them and the arg-max pick.
    MNActivePMF, rank 10: the KL at initialize_approx() and the scoring of the full grid there + the pick."""
    import gzip
    from itertools import islice
    t0 = time.perf_counter()
    with gzip.open(os.path.join(REF, 'movielens-100k', 'ratings_matrix.npy.gz')) as f:
        real = np.load(f)
    n, m = real.shape
    ii, jj = real.nonzero()
    ratings = np.column_stack([ii, jj, real[ii, jj]]).astype(float)
    save('ml100k_real', real=real.astype(np.int8))
    # ---- PMF (configs[1])
    seed(91)
    p = pmf_cy.ProbabilisticMatrixFactorization(ratings.copy(), latent_d=10, subtract_mean=True)
    U0, V0 = p.users.copy(), p.items.copy()
    gU, gV = p.gradient()
    ll0 = p.log_likelihood()
    rows_u, rows_v = np.arange(0, n, 31), np.arange(0, m, 47)
    lls = list(islice(p.fit_lls(), 20))
    out = dict(seed_ctor=91, d=10, N=ratings.shape[0], mean_rating=p.mean_rating, U0_sum=U0.sum(), V0_sum=V0.sum(),
               gU=gU, gV=gV, ll0=ll0, ll_trace=np.array(lls), learning_rate=p.learning_rate,
               rows_u=rows_u, rows_v=rows_v, U_rows=p.users[rows_u].copy(), V_rows=p.items[rows_v].copy(),
               U_sum=p.users.sum(), V_sum=p.items.sum(), rmse=float(p.rmse(real.astype(float))))
    p.update_sigma()
    p.update_sigma_uv()
    out.update(sigma_after=np.array([p.sigma_sq, p.sigma_u_sq, p.sigma_v_sq]))
    save('ml100k_pmf_d10', **out)
    _print('ml100k PMF: %.0f s' % (time.perf_counter() - t0))
    # ---- BayesianPMF (configs[2])
    seed(92)
    b = bayes_pmf.BayesianPMF(ratings.copy(), latent_d=15, subtract_mean=True)
    seed(93)
    samps = list(islice(b.samples(), 2))
    pv, pr = b.pred_variance(samps), b.predict(samps)
    pi, pj = np.arange(0, n, 41), np.arange(0, m, 67)
    save('ml100k_bayes_d15', seed_ctor=92, seed_chain=93, d=15, U0_sum=b.users.sum(), V0_sum=b.items.sum(),
         rows_u=rows_u, rows_v=rows_v,
         chain_U_rows=np.array([u[rows_u] for u, v in samps]), chain_V_rows=np.array([v[rows_v] for u, v in samps]),
         chain_U_sum=np.array([u.sum() for u, v in samps]), chain_V_sum=np.array([v.sum() for u, v in samps]),
         pi=pi, pj=pj, pred_variance_sub=pv[np.ix_(pi, pj)], predict_sub=pr[np.ix_(pi, pj)],
         pred_variance_sum=pv.sum(), pick_argmax=np.array(np.unravel_index(np.argmax(pv), pv.shape)))
    _print('ml100k Bayes: %.0f s' % (time.perf_counter() - t0))
    # ---- MNActivePMF scoring on the real data: the metric's kernel fed by a model of the real ratings
    seed(94)
    a = mn_active_pmf.MNActivePMF(ratings.copy(), latent_d=10)
    for _ in islice(a.fit_lls(), 15):
        pass
    a.initialize_approx()
    pm, pv = a.approx_pred_means_vars()
    save('ml100k_mn_d10', seed_ctor=94, d=10, fit_steps=15, users_sum=a.users.sum(), items_sum=a.items.sum(),
         rows_u=rows_u, users_rows=a.users[rows_u].copy(), kl0=a.kl_divergence(),
         p_mean_rows=pm[rows_u].copy(), p_var_rows=pv[rows_u].copy(), p_var_sum=pv.sum(), p_mean_sum=pm.sum(),
         pick0=np.array(a.pick_query_point(key=mn_active_pmf.MNActivePMF.pred_variance)))
    _print('ml100k MN: %.0f s' % (time.perf_counter() - t0))


def round4c(np, mn_active_pmf, save, seed, _print):
    """AF. BASELINE.json configs[3] as the reference's Makefile RUNS it (results/drugbank-94x425/Makefile:62-73:
    mn_active_pmf.py --latent-d 20 --discrete-integration --refit-lookahead ... -- random pred-variance pred
    prob-ge-3.5): the selection loop compare() end to end on the DrugBank subset with the 518 known entries of
    section G, rank 20, under a seed - initial fit_lls + fit_normal to convergence, then per key and step: evaluate the
    criterion over the 39 432 unrated entries, pick, add_rating, re-fit, variational re-fit.  Two query steps per key.
    Recorded per key: the picks, the RMSEs, the number of known entries; for the first step the criterion values of
    sampled candidates; the initial model's factors and approximation (diagonal / samples of the 519 x 519 covariance)."""
    t0 = time.perf_counter()
    g = np.load(os.path.join(HERE, 'mn_c4_94x425_d5.npz'))
    real, ratings = g['real'].astype(float), g['ratings']
    keys = ['random', 'pred-variance', 'pred', 'prob-ge-3.5']
    seed(95)
    res = mn_active_pmf.compare(keys, real, ratings.copy(), rating_vals=(0, 1), latent_d=20, do_threading=False,
                                steps=3, discrete_exp=True, refit_lookahead=True)
    init = res['_initial_apmf']
    rs = np.random.RandomState(96)
    N = init.cov_useritems.shape[0]
    si, sj = rs.randint(0, N, 3000), rs.randint(0, N, 3000)
    out = dict(seed=95, keys=np.array(keys), latent_d=20, init_users=init.users, init_items=init.items,
               init_mean=init.mean, init_cov_latents=init.cov_latents, samp_i=si, samp_j=sj,
               init_cov_diag=np.diag(init.cov_useritems).copy(), init_cov_samp=init.cov_useritems[si, sj].copy(),
               init_kl=init.kl_divergence())
    ci, cj = rs.randint(0, real.shape[0], 2000), rs.randint(0, real.shape[1], 2000)
    out.update(cand_i=ci, cand_j=cj)
    # the whole initial approximation (2.1 MB): every key's FIRST pick is a function of it alone, so the test can
    # separate "the criterion and the pick from a given state" (exact) from "where the chaotic fit ended" (1e-3)
    out.update(init_cov_useritems=init.cov_useritems.copy())
    pool0 = list(init.unrated)
    out.update(pool0_head=np.array(pool0[:2000]))
    for key in keys:
        steps = res[key]
        out[key + '_n_rated'] = np.array([s[0] for s in steps])
        out[key + '_rmse'] = np.array([s[1] for s in steps], dtype=np.float64)
        out[key + '_picks'] = np.array([s[2] for s in steps[1:]])
        for k, s in enumerate(steps[1:]):
            vals = np.asarray(s[3], dtype=np.float64)
            # (an (n, m) grid with NaN at the rated entries - get_key_evals - for the criteria; sampled)
            out['%s_vals_%d' % (key, k)] = vals[ci, cj] if vals.ndim == 2 else vals.reshape(-1)[:2000]
            out['%s_vals_ndim_%d' % (key, k)] = vals.ndim
            if k == 0 and vals.ndim == 1 and vals.size == len(pool0):
                # the thirty best of the first step, in the criterion's order (max; all four keys maximise)
                top = np.argsort(-vals, kind='stable')[:30]
                out[key + '_top_vals'] = vals[top]
                out[key + '_top_ij'] = np.array([pool0[t] for t in top])
            out['%s_nanmax_%d' % (key, k)] = np.nanmax(vals)
    save('trace_c4_mn_d20', **out)
    _print('configs[3] compare() trace: %.0f s' % (time.perf_counter() - t0))


def round4d(np, mn_active_pmf, save, seed, _print):
    """AF2. The criteria of section AF evaluated by the reference ON A GIVEN STATE - the initial model of that run, which
    its fixture holds whole (rank 20, 519 x 519 covariance): pred-variance, pred and prob-ge-3.5 over all 39 432 unrated
    entries (mn_active_pmf.py:404-511 through get_key_evals, :764-779) and the pick of each (:701-729).  compare() itself
    re-fits every key's copy before its first pick, so the picks of section AF are NOT functions of this state; these are."""
    t0 = time.perf_counter()
    g = np.load(os.path.join(HERE, 'trace_c4_mn_d20.npz'))
    base = np.load(os.path.join(HERE, 'mn_c4_94x425_d5.npz'))
    MN = mn_active_pmf.MNActivePMF
    a = MN(base['ratings'].copy(), latent_d=int(g['latent_d']), rating_values=(0, 1), discrete_expectations=True,
           refit_lookahead=True)
    a.users, a.items = g['init_users'].copy(), g['init_items'].copy()
    a.mean, a.cov_useritems, a.cov_latents = g['init_mean'].copy(), g['init_cov_useritems'].copy(), g['init_cov_latents'].copy()
    out = dict(kl=a.kl_divergence())
    rs = np.random.RandomState(97)
    ci, cj = rs.randint(0, a.num_users, 3000), rs.randint(0, a.num_items, 3000)
    out.update(cand_i=ci, cand_j=cj)
    for key in ('pred-variance', 'pred', 'prob-ge-3.5'):
        kf = mn_active_pmf.KEY_FUNCS[key]
        ev = a.get_key_evals(key=kf, procs=1)
        flat = np.where(np.isnan(ev), -np.inf, ev).reshape(-1)
        top = np.argsort(-flat, kind='stable')[:50]
        out[key + '_top_ij'] = np.column_stack(np.unravel_index(top, ev.shape))
        out[key + '_top_vals'] = flat[top]
        out[key + '_samp'] = ev[ci, cj]
        out[key + '_nansum'] = np.nansum(ev)
        out[key + '_pick'] = np.array(a.pick_query_point(key=kf, procs=1))
        _print('  %s: pick %s, %.0f s' % (key, out[key + '_pick'], time.perf_counter() - t0))
    save('c4_d20_init_criteria', **out)


def round4e(np, bayes_pmf, save, seed, _print):
    """AG. The Bayesian selection loop end to end at BASELINE.json configs[2]'s size on the REAL MovieLens-100k matrix
    (bayes_pmf.py:682-826: compare_active -> full_test): rank 15, subtract_mean, a seeded half of the 100 000 ratings known,
    the other half the candidates (compare_active's `pickable`), keys random / pred-variance / prob-ge-3.5, 12 chain samples,
    two query steps each, sequential (procs=0) under seed 98.  Picks, RMSEs, sampled criterion values."""
    t0 = time.perf_counter()
    real = np.load(os.path.join(HERE, 'ml100k_real.npz'))['real'].astype(float)
    ii, jj = real.nonzero()
    # known entries: the first rating of every user and of every item (the reference's chain raises KeyError on a
    # row or column without one, bayes_pmf.py:287,296), then a seeded fill up to 50 000; the order of the list is the
    # row-major order of the matrix
    rs = np.random.RandomState(99)
    first_u = np.unique(ii, return_index=True)[1]
    order_j = np.argsort(jj, kind='stable')
    first_v = order_j[np.unique(jj[order_j], return_index=True)[1]]
    mask = np.zeros(ii.size, bool)
    mask[first_u] = True
    mask[first_v] = True
    rest = np.flatnonzero(~mask)
    mask[rs.permutation(rest)[:50000 - int(mask.sum())]] = True
    known = np.flatnonzero(mask)
    ratings = np.column_stack([ii[known], jj[known], real[ii[known], jj[known]]]).astype(float)
    keys = ['random', 'pred-variance', 'prob-ge-3.5']
    seed(98)
    res = bayes_pmf.compare_active(keys, 15, real, ratings.copy(), rating_vals=(1, 2, 3, 4, 5), num_steps=3,
                                   procs=0, num_samps=12, subtract_mean=True)
    out = dict(seed=98, seed_known=99, n_known=50000, keys=np.array(keys), latent_d=15, num_samps=12)
    for key in keys:
        steps = res[key]
        out[key + '_n_rated'] = np.array([s[0] for s in steps])
        out[key + '_rmse'] = np.array([s[1] for s in steps], dtype=np.float64)
        out[key + '_picks'] = np.array([s[2] for s in steps[1:]])
        for k, s in enumerate(steps[1:]):
            vals = np.asarray(s[3], dtype=np.float64)
            out['%s_vals_shape_%d' % (key, k)] = np.array(vals.shape)
            flat = np.where(np.isnan(vals), -np.inf, vals).reshape(-1)
            top = np.argsort(-flat, kind='stable')[:20]
            out['%s_top_idx_%d' % (key, k)] = top
            out['%s_top_vals_%d' % (key, k)] = flat[top]
            out['%s_head_%d' % (key, k)] = vals.reshape(-1)[:3000]
        _print('bayes ml100k %s: picks %s, %.0f s' % (key, out[key + '_picks'].tolist(), time.perf_counter() - t0))
    save('trace_ml100k_bayes_d15', **out)


def round4f(np, pmf_cy, save, seed, _print):
    """AH. BASELINE.json configs[1] - "PMF SGD fit" - on the REAL MovieLens-100k matrix: the minibatch momentum fit
    (pmf_cy.pyx:308-381), rank 10, subtract_mean, ratings = the matrix' non-zeros in row-major order: two epochs of
    fit_minibatches(1000) and three (train, validation) pairs of fit_minibatches_validation(1000, 5000) under fixed
    seeds - what section Z records on the synthetic matrix of that shape."""
    from itertools import islice
    t0 = time.perf_counter()
    real = np.load(os.path.join(HERE, 'ml100k_real.npz'))['real']
    ii, jj = real.nonzero()
    ratings = np.column_stack([ii, jj, real[ii, jj]]).astype(float)
    seed(100)
    p = pmf_cy.ProbabilisticMatrixFactorization(ratings.copy(), latent_d=10, subtract_mean=True)
    U0, V0 = p.users.copy(), p.items.copy()
    seed(101)
    errs = list(islice(p.fit_minibatches(1000), 2))
    ru, rv = np.array([0, 1, 470, 942]), np.array([0, 1, 840, 1681])
    out = dict(seed_ctor=100, seed_fit=101, batch_size=1000, U0_sum=U0.sum(), V0_sum=V0.sum(),
               train_errs=np.array(errs, np.float64), rows_u=ru, rows_v=rv, U_rows=p.users[ru], V_rows=p.items[rv],
               U_sum=p.users.sum(), V_sum=p.items.sum(), U_abs_sum=np.abs(p.users).sum(), V_abs_sum=np.abs(p.items).sum(),
               ratings_after_head=p.ratings[:64].copy(),
               ratings_after_weighted=(p.ratings * np.arange(1, ratings.shape[0] + 1)[:, None]).sum(0))
    seed(102)
    q = pmf_cy.ProbabilisticMatrixFactorization(ratings.copy(), latent_d=10, subtract_mean=True)
    q.users, q.items = U0.copy(), V0.copy()
    seed(103)
    pairs = list(islice(q.fit_minibatches_validation(1000, 5000), 3))
    out.update(seed_ctor2=102, seed_pairs=103, valid_size=5000, valid_pairs=np.array(pairs, np.float64),
               Uv_rows=q.users[ru], Vv_rows=q.items[rv])
    _print('ml100k minibatch: train errs %r, pairs %r, %.0f s' % (errs, pairs, time.perf_counter() - t0))
    save('ml100k_minibatch_d10', **out)


def round2b(np, mn_active_pmf, save, seed, _print):
    """J. The matrix-normal scoring sweep at the HEADLINE shape (BASELINE.json metric: 943 x 1682, rank 10),
    from the real reference: approx_pred_means_vars() over all 1 586 126 entries and pick_query_point over
    the 1 486 126 unrated ones, for a posterior state generated from a few small factors (the
    (n+m) x (n+m) covariance itself is 55 MB and is not stored):
        mean = Mf,   cov_useritems = A A' + 0.1 I  (A: (n+m) x 8),   cov_latents = B B' / d + 0.1 I."""
    MN = mn_active_pmf.MNActivePMF
    t0 = time.perf_counter()
    ratings = headline_inputs(np, seed)
    n, m, d = 943, 1682, 10
    seed(80)
    Mf = np.random.normal(0, 1, (n + m, d))
    A = np.random.normal(0, 1, (n + m, 8)) / 3.
    B = np.random.normal(0, 1, (d, d))
    a = MN(ratings.copy(), latent_d=d)
    a.mean = Mf.copy()
    a.cov_useritems = A @ A.T + .1 * np.eye(n + m)
    a.cov_latents = B @ B.T / d + .1 * np.eye(d)
    pm, pv = a.approx_pred_means_vars()
    _print('headline MN grids: %.0f s' % (time.perf_counter() - t0))
    pick = a.pick_query_point(key=MN.pred_variance)
    _print('headline MN pick: %.0f s' % (time.perf_counter() - t0))
    rows = np.array([0, 1, 470, 471, 941, 942])
    rs = np.random.RandomState(81)
    si, sj = rs.randint(0, n, 20000), rs.randint(0, m, 20000)
    save('mn_c2_943x1682_d10', N=ratings.shape[0], ratings_sum=ratings.sum(0), Mf=Mf, A=A, B=B,
         rows=rows, p_mean_rows=pm[rows], p_var_rows=pv[rows], si=si, sj=sj, p_mean_s=pm[si, sj],
         p_var_s=pv[si, sj], total_variance=a._total_variance(), var_sum=pv.sum(), mean_sum=pm.sum(),
         pick=np.array(pick), pick_var=pv[pick])


def round2(np, pmf_cy, mn_active_pmf, bayes_pmf, save, seed, fake, _print):
    """Fixtures added in round 2: BASELINE.json configs[3] and configs[2] at their real sizes, and
    the minibatch fit."""
    from itertools import islice
    MN = mn_active_pmf.MNActivePMF

    # ---------------- G. configs[3]: DrugBank 94 x 425, rank 5, 500 known entries
    # (results/drugbank-94x425/Makefile: choose_training.py --drugbank --n-pick 500; the matrix is
    # the reference's own drugbank/subset_94x425.npy, read here, its values stored as data)
    t0 = time.perf_counter()
    real = np.load(os.path.join(REF, 'drugbank', 'subset_94x425.npy')).astype(float)
    n, m = real.shape
    seed(60)
    # every row and column known at least once (the Gibbs and the fit need no empty rows), then
    # uniformly up to 500
    mask = np.zeros((n, m), bool)
    mask[np.arange(n), np.random.randint(0, m, n)] = True
    mask[np.random.randint(0, n, m), np.arange(m)] = True
    free = np.flatnonzero(~mask.ravel())
    mask.ravel()[np.random.choice(free, max(0, 500 - int(mask.sum())), replace=False)] = True
    ii, jj = mask.nonzero()
    perm = np.random.permutation(ii.size)
    ii, jj = ii[perm], jj[perm]
    ratings = np.column_stack([ii, jj, real[ii, jj]]).astype(float)
    seed(61)
    a = MN(ratings.copy(), latent_d=5, rating_values=(0., 1.), discrete_expectations=True)
    a.do_fit()
    a.initialize_approx()
    out = dict(real=real.astype(np.int8), ratings=ratings, users=a.users, items=a.items,
               sigma=np.array([a.sigma_sq, a.sigma_u_sq, a.sigma_v_sq]), kl0=a.kl_divergence())
    pm, pv = a.approx_pred_means_vars()
    out.update(p_mean0=pm, p_var0=pv, pick0=np.array(a.pick_query_point(key=MN.pred_variance)))
    kls = list(islice(a.fit_normal_kls(), 3))
    pm, pv = a.approx_pred_means_vars()
    out.update(kl_trace3=np.array(kls), mean3=a.mean.copy(), cov_latents3=a.cov_latents.copy(),
               cov_diag3=np.diag(a.cov_useritems).copy(), p_mean3=pm, p_var3=pv)
    _print('c4: 3 steps done, %.0f s' % (time.perf_counter() - t0))
    rest = list(a.fit_normal_kls())          # to convergence (a fresh search from the 3-step state)
    pm, pv = a.approx_pred_means_vars()
    out.update(fit_steps=len(rest), kl_fit=rest[-1], kl_trace_fit=np.array(rest), mean_fit=a.mean.copy(),
               cov_useritems_fit=a.cov_useritems.copy(), cov_latents_fit=a.cov_latents.copy(),
               p_mean_fit=pm, p_var_fit=pv, total_variance_fit=a._total_variance(),
               pick_fit=np.array(a.pick_query_point(key=MN.pred_variance)),
               entropy_fit=a._approx_entropy())
    _print('c4: fitted in %d more steps, %.0f s' % (len(rest), time.perf_counter() - t0))
    cand = sorted(a.unrated)[11]
    out.update(cand=np.array(cand), exp_total_variance=a.exp_total_variance(cand))
    _print('c4: lookahead done, %.0f s' % (time.perf_counter() - t0))
    save('mn_c4_94x425_d5', **out)

    # ---------------- H. minibatch fit (pmf_cy.pyx:308-381)
    seed(70)
    r, _, _ = fake(50, 40, 3, 8)
    seed(71)
    p = pmf_cy.ProbabilisticMatrixFactorization(r.copy(), latent_d=3)
    U0, V0 = p.users.copy(), p.items.copy()
    seed(72)
    errs = list(islice(p.fit_minibatches(64), 2))
    out = dict(ratings=r, U0=U0, V0=V0, batch_size=64, seed_fit=72, train_errs=np.array(errs, np.float64),
               users=p.users.copy(), items=p.items.copy(), ratings_after=p.ratings.copy())
    seed(73)
    q = pmf_cy.ProbabilisticMatrixFactorization(r.copy(), latent_d=3)
    q.users, q.items = U0.copy(), V0.copy()
    seed(74)
    q.fit_minibatches_until_validation(64, 60)
    out.update(seed_valid=74, valid_size=60, users_valid=q.users.copy(), items_valid=q.items.copy())
    seed(75)
    q2 = pmf_cy.ProbabilisticMatrixFactorization(r.copy(), latent_d=3)
    q2.users, q2.items = U0.copy(), V0.copy()
    seed(76)
    pairs = list(islice(q2.fit_minibatches_validation(64, 60), 3))
    out.update(seed_pairs=76, valid_pairs=np.array(pairs, np.float64))
    save('pmf_minibatch_50x40_d3', **out)

    # ---------------- I. configs[2]: rank-15 Gibbs at 943 x 1682 (inputs regenerated from seed 0 by
    # the test, exactly as c2_inputs() in tests/conftest.py; only sampled outputs are stored)
    seed(0)
    n, m, d, N = 943, 1682, 10, 100000
    Ut = np.random.normal(0, 2, (n, d))
    Vt = np.random.normal(0, 2, (m, d))
    mask = np.zeros((n, m), bool)
    mask[np.arange(n), np.random.randint(0, m, n)] = True
    mask[np.random.randint(0, n, m), np.arange(m)] = True
    free = np.flatnonzero(~mask.ravel())
    extra = np.random.choice(free, N - int(mask.sum()), replace=False)
    mask.ravel()[extra] = True
    ii, jj = mask.nonzero()
    perm = np.random.permutation(ii.size)
    ii, jj = ii[perm], jj[perm]
    vals = (Ut[ii] * Vt[jj]).sum(1) + np.random.normal(0, .25, ii.size)
    ratings = np.column_stack([ii, jj, vals]).astype(float)
    seed(80)
    b = bayes_pmf.BayesianPMF(ratings.copy(), latent_d=15)
    U0, V0 = b.users.copy(), b.items.copy()
    seed(81)
    samps = list(islice(b.samples(), 3))
    rows_u = np.arange(0, n, 37)
    rows_v = np.arange(0, m, 53)
    pv = b.pred_variance(samps)
    pr = b.predict(samps)
    pi, pj = np.arange(0, n, 41), np.arange(0, m, 67)
    save('bayes_c3_943x1682_d15', U0_sum=U0.sum(), V0_sum=V0.sum(), seed_ctor=80, seed_chain=81,
         rows_u=rows_u, rows_v=rows_v,
         chain_U_rows=np.array([u[rows_u] for u, v in samps]),
         chain_V_rows=np.array([v[rows_v] for u, v in samps]),
         chain_U_sum=np.array([u.sum() for u, v in samps]),
         chain_V_sum=np.array([v.sum() for u, v in samps]),
         pi=pi, pj=pj, pred_variance_sub=pv[np.ix_(pi, pj)], predict_sub=pr[np.ix_(pi, pj)],
         pred_variance_sum=pv.sum(), pick_argmax=np.array(np.unravel_index(np.argmax(pv), pv.shape)))


if __name__ == '__main__':
    main()
