"""The CPU oracle (oracle/apmf_oracle.c) pinned against golden vectors produced by the
REAL reference (tests/golden/make_golden.py).  CPU-only; this is what makes the oracle
trustworthy as the checker of the HIP path."""
import numpy as np
import pytest

from conftest import load_golden, c2_inputs
from oracle import oracle as orc

EPS = np.finfo(np.float64).eps

PMF_CASES = ['pmf_c1_10x10_d2', 'pmf_50x40_d3', 'pmf_50x40_d3_submean', 'pmf_30x70_d16']


def _pmf_args(g):
    sig, sig_u, sig_v = g['sigma']
    return dict(sig=sig, sig_u=sig_u, sig_v=sig_v, subtract_mean=bool(g['subtract_mean']),
                mean_rating=float(g['mean_rating']))


@pytest.mark.parametrize('name', PMF_CASES)
def test_pmf_gradient_and_ll(name):
    g = load_golden(name)
    kw = _pmf_args(g)
    gU, gV = orc.pmf_gradient(g['ratings'], g['U0'], g['V0'], **kw)
    # reference: pmf_cy.pyx:204-223.  Only np.dot's internal summation order differs.
    np.testing.assert_allclose(gU, g['gU'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(gV, g['gV'], rtol=1e-12, atol=1e-12)
    ll = orc.pmf_log_likelihood(g['ratings'], g['U0'], g['V0'], **kw)
    assert abs(ll - g['ll']) <= 1e-13 * abs(g['ll'])
    lo, hi = int(g['batch_lo']), int(g['batch_hi'])
    gUb, gVb = orc.pmf_gradient(g['ratings'][lo:hi], g['U0'], g['V0'], **kw)
    np.testing.assert_allclose(gUb, g['gU_batch'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(gVb, g['gV_batch'], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('name', PMF_CASES)
def test_pmf_fit_lls_trace(name):
    g = load_golden(name)
    U, V, trace = orc.pmf_fit_lls(g['ratings'], g['U0'], g['V0'], **_pmf_args(g))
    # same number of accepted steps = same accept/reject control flow (pmf_cy.pyx:265-291)
    assert trace.size == g['ll_trace'].size
    np.testing.assert_allclose(trace, g['ll_trace'], rtol=1e-10)
    np.testing.assert_allclose(U, g['U_fit'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(V, g['V_fit'], rtol=1e-8, atol=1e-10)


def test_pmf_headline_shape():
    g = load_golden('pmf_c2_943x1682_d10')
    ratings, U0, V0 = c2_inputs()
    assert ratings.shape[0] == int(g['N'])
    np.testing.assert_array_equal(ratings[:64], g['ratings_head'])
    np.testing.assert_allclose(ratings.sum(0), g['ratings_sum'], rtol=1e-13)
    gU, gV = orc.pmf_gradient(ratings, U0, V0)
    np.testing.assert_allclose(gU, g['gU'], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(gV, g['gV'], rtol=1e-11, atol=1e-11)
    ll = orc.pmf_log_likelihood(ratings, U0, V0)
    assert abs(ll - g['ll']) <= 1e-13 * abs(g['ll'])


def test_pmf_headline_fit_trace():
    """fit_lls at BASELINE.json configs[1]'s shape (943 x 1682, rank 10, N = 100 000): the reference's
    log-likelihood after each of its first 20 accepted steps, and its factors there."""
    g = load_golden('pmf_c2_fit_943x1682_d10')
    ratings, U0, V0 = c2_inputs()
    assert abs(U0.sum() - g['U0_sum']) <= 1e-9 and abs(V0.sum() - g['V0_sum']) <= 1e-9
    U, V, trace = orc.pmf_fit_lls(ratings, U0, V0, lr=float(g['learning_rate']), max_steps=20)
    assert len(trace) == 20
    np.testing.assert_allclose(trace, g['ll_trace'], rtol=1e-11)
    np.testing.assert_allclose(U[g['rows_u']], g['U_rows'], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(V[g['rows_v']], g['V_rows'], rtol=1e-9, atol=1e-11)
    assert abs(U.sum() - g['U_sum']) <= 1e-9 * abs(g['U_sum']) and abs(V.sum() - g['V_sum']) <= 1e-9 * abs(g['V_sum'])


MN_CASES = ['mn_c1_10x10_d2_eye', 'mn_c1_10x10_d2_rand', 'mn_24x17_d5_rand', 'mn_12x20_d10_rand']


@pytest.mark.parametrize('name', MN_CASES)
def test_mn_grids(name):
    g = load_golden(name)
    n, m = g['p_mean'].shape
    pm, pv, e2 = orc.mn_pred_means_vars(n, m, g['mean'], g['cov_useritems'], g['cov_latents'],
                                        want_e2=True)
    # exp_dotprod_sq is pure C arithmetic in the reference (matrix_normal_exps_cy.pyx:126-154):
    # the restatement reproduces it bit for bit.
    np.testing.assert_array_equal(e2, g['e2'])
    # mean/var go through numpy scalar ops (pow, sum) in mn_active_pmf.py:311-315
    assert np.max(np.abs(pm - g['p_mean'])) <= 4 * EPS * np.max(np.abs(g['p_mean']))
    assert np.all(np.abs(pv - g['p_var']) <= 8 * EPS * np.abs(g['e2']))
    # compact state (cross block + diagonal) gives the same numbers
    R = g['cov_useritems']
    pm2, pv2 = orc.mn_pred_means_vars_compact(n, m, g['mean'], R[:n, n:], np.diag(R).copy(),
                                              g['cov_latents'])
    np.testing.assert_array_equal(pm2, pm)
    np.testing.assert_array_equal(pv2, pv)
    # the selection the reference made (first max in pool order, mn_active_pmf.py:728-729)
    pool = [tuple(x) for x in g['pool']]
    vals = [pv[i, j] for i, j in pool]
    np.testing.assert_allclose(vals, g['pool_vals'], rtol=0, atol=8 * EPS * np.abs(g['e2']).max())
    pick = max(zip(pool, vals), key=lambda t: t[1])[0]
    assert pick == tuple(g['pick_max'])
    assert abs(pv.sum() - g['total_variance']) <= 1e-12 * abs(g['total_variance'])


def test_mn_moments():
    g = load_golden('moments_mn')
    a = (g['mean'], g['cov_rows'], g['cov_cols'])
    assert orc.mn_moment('tripexpect', *a, 0, 0, 1, 1, 2, 2) == g['tripexpect']
    assert orc.mn_moment('quadexpect', *a, 0, 0, 1, 1, 2, 2, 3, 0) == g['quadexpect']
    assert orc.mn_moment('exp_squared', *a, 0, 1, 4, 2) == g['exp_squared']
    assert orc.mn_moment('exp_a2bc', *a, 1, 1, 2, 0, 4, 2) == g['exp_a2bc']
    assert orc.mn_exp_dotprod_sq(2, *a, 1, 2) == g['exp_dotprod_sq']


def test_full_moments():
    g = load_golden('moments_full')
    a = (g['mean'], g['cov'])
    assert orc.full_moment('tripexpect', *a, 0, 1, 2) == pytest.approx(g['tripexpect'], rel=1e-15)
    assert orc.full_moment('quadexpect', *a, 0, 1, 2, 3) == g['quadexpect']
    assert orc.full_moment('exp_squared', *a, 1, 4) == pytest.approx(g['exp_squared'], rel=1e-15)
    assert orc.full_moment('exp_a2bc', *a, 2, 3, 5) == g['exp_a2bc']
    assert orc.full_exp_dotprod_sq(1, 3, *a, 0, 0) == pytest.approx(g['exp_dotprod_sq'], rel=1e-15)


def test_moments_monte_carlo():
    """What the reference's own test_normal_exps.py pins: the closed forms agree with a
    Monte-Carlo estimate to 2 % (test_normal_exps.py:22-63), here for the oracle."""
    rs = np.random.RandomState(7)
    for _ in range(2):
        dim = 6
        mn = rs.normal(0, 10, dim)
        s = rs.normal(0, 5, (dim, dim))
        w, v = np.linalg.eigh((s + s.T) / 2)
        cov = (v * np.maximum(w, 1e-5)) @ v.T
        cov = (cov + cov.T) / 2
        x = rs.multivariate_normal(mn, cov, 400000)
        u = x[:, :3]
        vv = x[:, 3:]
        monte = np.mean(np.sum(u * vv, axis=1) ** 2)
        e = orc.full_exp_dotprod_sq(1, 3, mn, cov, 0, 0)
        assert abs(monte - e) / e < .02
        monte4 = np.mean(x[:, 0] * x[:, 1] * x[:, 2] * x[:, 3])
        e4 = orc.full_moment('quadexpect', mn, cov, 0, 1, 2, 3)
        assert abs(monte4 - e4) / abs(e4) < .05


@pytest.mark.parametrize('name', ['full_6x5_d3', 'full_12x9_d5', 'full_11x14_d10', 'full_7x9_d16'])
def test_full_grids(name):
    g = load_golden(name)
    n, m = g['p_mean'].shape
    d = int(g['d'])
    pm, pv, e2 = orc.full_pred_means_vars(n, m, d, g['mean'], g['cov'], want_e2=True)
    scale = np.abs(e2)
    assert np.all(np.abs(pm - g['p_mean']) <= 8 * EPS * (np.abs(g['p_mean']) + 1))
    assert np.all(np.abs(pv - g['p_var']) <= 16 * EPS * scale)
    pool = [tuple(x) for x in g['pool']]
    vals = [pv[i, j] for i, j in pool]
    pick = max(zip(pool, vals), key=lambda t: t[1])[0]
    assert pick == tuple(g['pick_max'])


def test_movielens_100k_pmf():
    """The oracle's K1 restatement on the REAL MovieLens-100k matrix (fixture section AE: the reference's constructor
    under seed 91, rank 10, subtract_mean): gradient and log-likelihood at the seeded start."""
    import random
    g = load_golden('ml100k_pmf_d10')
    real = load_golden('ml100k_real')['real']
    ii, jj = real.nonzero()
    ratings = np.column_stack([ii, jj, real[ii, jj]]).astype(float)
    n, m = real.shape
    d = int(g['d'])
    np.random.seed(int(g['seed_ctor']))
    random.seed(int(g['seed_ctor']))
    U0, V0 = np.random.random((n, d)), np.random.random((m, d))          # pmf_cy.pyx:74-75
    assert abs(U0.sum() - g['U0_sum']) <= 1e-12 * abs(g['U0_sum'])
    kw = dict(subtract_mean=True, mean_rating=float(g['mean_rating']))
    gU, gV = orc.pmf_gradient(ratings, U0, V0, **kw)
    np.testing.assert_allclose(gU, g['gU'], rtol=1e-12, atol=1e-13 * np.abs(g['gU']).max())
    np.testing.assert_allclose(gV, g['gV'], rtol=1e-12, atol=1e-13 * np.abs(g['gV']).max())
    ll = orc.pmf_log_likelihood(ratings, U0, V0, **kw)
    assert abs(ll - float(g['ll0'])) <= 1e-12 * abs(float(g['ll0']))


def test_full_grid_configs3_shape():
    """The full-covariance scoring at BASELINE.json configs[3]'s shape (K = 2595; fixture section AD): the 54 MB
    covariance is not stored - it is rebuilt here the way the reference built it (seeded K x K normal draw, symmetrised,
    eigenvalues clipped at 1e-5: active_pmf.py:190-200, mn_active_pmf.py:42-67), checked against the recorded diagonal
    and samples, and the oracle's approx_pred_means_vars over all 39 950 entries compared with the reference's."""
    import random
    g = load_golden('full_c4_94x425_d5')
    K, d = int(g['K']), int(g['d'])
    np.random.seed(int(g['seed_init']))
    random.seed(int(g['seed_init']))
    s = np.random.normal(0, 2, (K, K))
    s += s.T
    s /= 2
    vals, vecs = np.linalg.eigh(s)
    cov = vecs @ np.diag(np.maximum(vals, 1e-5)) @ vecs.T
    cov += cov.T
    cov /= 2
    scale = np.abs(g['cov_diag0']).max()
    assert np.abs(np.diag(cov) - g['cov_diag0']).max() <= 1e-10 * scale
    assert np.abs(cov[g['samp_i'], g['samp_j']] - g['cov_samp0']).max() <= 1e-10 * scale
    n, m = g['p_mean0'].shape
    pm, pv, e2 = orc.full_pred_means_vars(n, m, d, g['mean0'], cov, want_e2=True)
    # (cov is the rebuilt one: its eigenvectors differ from the reference's LAPACK build in the last places)
    assert np.all(np.abs(pm - g['p_mean0']) <= 1e-9 * (np.abs(g['p_mean0']) + 1))
    assert np.all(np.abs(pv - g['p_var0']) <= 1e-9 * np.abs(e2))
    unrated = np.ones((n, m), bool)
    unrated[g['ratings'][:, 0].astype(int), g['ratings'][:, 1].astype(int)] = False
    best = np.unravel_index(np.argmax(np.where(unrated, pv, -np.inf)), pv.shape)
    assert tuple(int(x) for x in best) == tuple(g['pick0'])


BAYES_CASES = ['bayes_20x15_d4', 'bayes_c1_10x10_d2', 'bayes_100x90_d15']


@pytest.mark.parametrize('name', BAYES_CASES)
def test_sample_feature(name):
    g = load_golden(name)
    sub = float(g['mean_rating']) if int(g['subtract_mean']) else 0.
    out = orc.sample_feature(g['items'], g['sf_rated'], g['sf_ratings'], g['sf_alpha'],
                             g['sf_mu'], 2.0, sub, g['sf_z'])
    # bayes_pmf.py:189-216 (LAPACK inv/cholesky vs plain LU/Cholesky: rounding only)
    np.testing.assert_allclose(out, g['sf_out'], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('name', BAYES_CASES)
def test_sample_stats(name):
    g = load_golden(name)
    add = float(g['mean_rating']) if int(g['subtract_mean']) else 0.
    mean, var, _ = orc.sample_stats(g['chain_U'], g['chain_V'], add=add)
    np.testing.assert_allclose(mean, g['predict_full'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(var, g['pred_variance_full'], rtol=1e-9, atol=1e-13)
    which = tuple(g['which'])
    np.testing.assert_allclose(var[which], g['pred_variance'], rtol=1e-9, atol=1e-13)
    assert tuple(g['which'][:, np.argmax(var[which])]) == tuple(g['pick_argmax'])
    _, _, p35 = orc.sample_stats(g['chain_U'], g['chain_V'], add=add, cutoff=3.5)
    np.testing.assert_array_equal(p35[which], g['prob_ge'])
    _, _, p05 = orc.sample_stats(g['chain_U'], g['chain_V'], add=add, cutoff=0.5)
    np.testing.assert_array_equal(p05, g['prob_ge_full'])


def _fit_with_sigmas_trace(g, count):
    """fit_with_sigmas_lls (pmf_cy.pyx:384-403) composed from the oracle's gradient / LL: the
    generator keeps the LL of the previous step as `old_ll` even after the sigmas change."""
    ratings, d = g['ratings'], int(g['d'])
    sub, mr = bool(g['subtract_mean']), float(g['mean_rating'])
    st = dict(zip(('sig', 'sig_u', 'sig_v'), (float(x) for x in g['sigma'])))
    U, V = g['U0'].copy(), g['V0'].copy()
    n, m = U.shape[0], V.shape[0]

    def ll(A, B):
        return orc.pmf_log_likelihood(ratings, A, B, subtract_mean=sub, mean_rating=mr, **st)

    def fit_lls():
        nonlocal U, V
        lr, old_ll, conv = 1e-4, ll(U, V), False
        while not conv:
            gU, gV = orc.pmf_gradient(ratings, U, V, subtract_mean=sub, mean_rating=mr, **st)
            while True:
                nU, nV = U + lr * gU, V + lr * gV
                new_ll = ll(nU, nV)
                if new_ll > old_ll:
                    U, V = nU, nV
                    lr *= 1.25
                    if new_ll - old_ll < 1e-2:
                        conv = True
                    yield new_ll
                    old_ll = new_ll
                    break
                lr *= .5
                if lr < 1e-10:
                    conv = True
                    break

    out, cont = [], True
    while cont and len(out) < count:
        cont = False
        for i, v in enumerate(fit_lls()):
            if i % 5 == 0:
                st['sig'] = orc.pmf_sq_error(ratings, U, V, sub, mr) / ratings.shape[0]
            if i % 2 == 0:
                st['sig_u'] = (U * U).sum() / n / d
                st['sig_v'] = (V * V).sum() / m / d
            out.append(v)
            cont = True
            if len(out) >= count:
                return np.array(out), U, V, st
        st['sig'] = orc.pmf_sq_error(ratings, U, V, sub, mr) / ratings.shape[0]
        st['sig_u'] = (U * U).sum() / n / d
        st['sig_v'] = (V * V).sum() / m / d
    return np.array(out), U, V, st


@pytest.mark.parametrize('name', PMF_CASES)
def test_pmf_fit_with_sigmas_trace(name):
    g = load_golden(name)
    want = g['sigmas_ll_trace']
    got, U, V, st = _fit_with_sigmas_trace(g, want.size)
    assert got.size == want.size
    np.testing.assert_allclose(got, want, rtol=1e-9)


# ---------------------------------------------------------------- round-2 fixtures (real sizes)
def test_configs3_fixture_oracle():
    """BASELINE.json configs[3] (DrugBank 94 x 425, rank 5, 500 known), recorded from the reference:
    the oracle's per-candidate path reproduces the reference's grids at the identity start and at
    the converged approximation, and the arg-max the reference picked."""
    g = load_golden('mn_c4_94x425_d5')
    n, m = g['users'].shape[0], g['items'].shape[0]
    # 518 known entries: every one of the 94 rows and 425 columns at least once (the 500 of the
    # Makefile's --n-pick would leave empty rows, which the reference's own Gibbs path rejects)
    assert (n, m) == (94, 425) and g['ratings'].shape == (518, 3)
    pm, pv = orc.mn_pred_means_vars(n, m, g['mean_fit'], g['cov_useritems_fit'], g['cov_latents_fit'])
    np.testing.assert_allclose(pv, g['p_var_fit'], rtol=1e-9, atol=1e-12 * np.abs(g['p_var_fit']).max())
    np.testing.assert_allclose(pm, g['p_mean_fit'], rtol=1e-10, atol=1e-12)
    assert abs(pv.sum() - g['total_variance_fit']) <= 1e-10 * abs(g['total_variance_fit'])
    mean0 = np.vstack((g['users'], g['items']))
    pm0, pv0 = orc.mn_pred_means_vars(n, m, mean0, np.eye(n + m), np.eye(5))
    np.testing.assert_allclose(pv0, g['p_var0'], rtol=1e-10)
    # the reference's pick = first maximum over its unrated set; with distinct values that is the
    # masked arg-max of the grid
    rated = np.zeros((n, m), bool)
    rated[g['ratings'][:, 0].astype(int), g['ratings'][:, 1].astype(int)] = True
    for pv_, pick in ((pv0, g['pick0']), (pv, g['pick_fit'])):
        cand = np.where(rated, -np.inf, pv_)
        assert cand[tuple(pick)] == cand.max()


def headline_state(g):
    """The posterior state of the headline-shape fixture from its small factors (make_golden.py, section J)."""
    n, m, d = 943, 1682, 10
    mean = g['Mf']
    R = g['A'] @ g['A'].T + .1 * np.eye(n + m)
    Cl = g['B'] @ g['B'].T / d + .1 * np.eye(d)
    return n, m, d, mean, R, Cl


def test_headline_shape_fixture_oracle():
    """BASELINE.json's metric shape (943 x 1682, rank 10), recorded from the reference: the oracle's
    per-candidate path reproduces the reference's mean / variance grid on six whole rows and 20 000
    sampled candidates, its totals over all 1 586 126 entries, and the entry the reference picked
    over the 1 486 126 unrated ones."""
    from conftest import c2_inputs
    g = load_golden('mn_c2_943x1682_d10')
    n, m, d, mean, R, Cl = headline_state(g)
    pm, pv = orc.mn_pred_means_vars(n, m, mean, R, Cl)
    scale = np.abs(g['p_var_rows']).max()
    np.testing.assert_allclose(pv[g['rows']], g['p_var_rows'], rtol=1e-10, atol=1e-13 * scale)
    np.testing.assert_allclose(pm[g['rows']], g['p_mean_rows'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(pv[g['si'], g['sj']], g['p_var_s'], rtol=1e-10, atol=1e-13 * scale)
    np.testing.assert_allclose(pm[g['si'], g['sj']], g['p_mean_s'], rtol=1e-10, atol=1e-12)
    assert abs(pv.sum() - g['var_sum']) <= 1e-11 * abs(g['var_sum'])
    assert abs(pv.sum() - g['total_variance']) <= 1e-11 * abs(g['total_variance'])
    ratings, _, _ = c2_inputs()
    assert ratings.shape[0] == int(g['N']) and np.allclose(ratings.sum(0), g['ratings_sum'])
    rated = np.zeros((n, m), bool)
    rated[ratings[:, 0].astype(int), ratings[:, 1].astype(int)] = True
    cand = np.where(rated, -np.inf, pv)
    pick = tuple(int(x) for x in g['pick'])
    assert cand[pick] == cand.max() and np.argmax(cand) == pick[0] * m + pick[1]
    assert abs(pv[pick] - float(g['pick_var'])) <= 1e-10 * abs(float(g['pick_var']))


def test_minibatch_fixture_oracle():
    """fit_minibatches (pmf_cy.pyx:308-351) recorded from the reference: two epochs of the momentum
    recipe composed from the oracle's gradient on the same shuffles give the reference's factors."""
    g = load_golden('pmf_minibatch_50x40_d3')
    import random as _random
    r_host, U, V = g['ratings'].copy(), g['U0'].copy(), g['V0'].copy()
    ui, vi = np.zeros_like(U), np.zeros_like(V)
    np.random.seed(int(g['seed_fit']))
    _random.seed(int(g['seed_fit']))
    bs = int(g['batch_size'])
    errs = []
    for _ in range(2):
        np.random.shuffle(r_host)
        for s in range(0, r_host.shape[0], bs):
            b = r_host[s:s + bs]
            gU, gV = orc.pmf_gradient(b, U, V)
            sc = float(np.float32(1.) / np.float32(b.shape[0]))     # lr = 1 default, C float arithmetic
            ui = ui * float(np.float32(.8)) + gU * sc
            U = U + ui
            vi = vi * float(np.float32(.8)) + gV * sc
            V = V + vi
        ij = r_host[:, :2].astype(int)
        pred = U @ V.T
        errs.append(float(np.float32(np.sqrt(np.mean((pred[ij[:, 0], ij[:, 1]] - r_host[:, 2]) ** 2)))))
    np.testing.assert_array_equal(r_host, g['ratings_after'])
    np.testing.assert_allclose(U, g['users'], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(V, g['items'], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(errs, g['train_errs'], rtol=1e-6)
