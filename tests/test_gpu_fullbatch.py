"""The batched variational-fit engine of the FULL-covariance class (csrc/k8_fullbatch.hip, apmf_full_batch_*) through
the C-ABI: its KL and its first step against the single-matrix routines on a container rebuilt with the hypothesis'
rating, and - what the lookahead relies on - a batch of H hypotheses giving bit for bit what each gives when fitted
alone (deepcopy -> add_rating -> fit_normal, active_pmf.py:660-668).  Needs an MI355X: run with -m gpu."""
import copy
import random

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from active_matrix_factorization_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _problem(n, m, d, seed):
    rs = np.random.RandomState(seed)
    K = (n + m) * d
    ii = np.concatenate([np.arange(n), rs.randint(0, n, m), rs.randint(0, n, 2 * (n + m))])
    jj = np.concatenate([rs.randint(0, m, n), np.arange(m), rs.randint(0, m, 2 * (n + m))])
    keep = np.sort(np.unique(ii * m + jj, return_index=True)[1])
    ii, jj = ii[keep], jj[keep]
    ratings = np.column_stack([ii, jj, rs.randint(1, 6, ii.size)]).astype(float)
    mean = rs.normal(size=K)
    A = rs.normal(size=(K, K)) / np.sqrt(K)
    cov = A @ A.T + .3 * np.eye(K)
    rated = set(zip(ii.tolist(), jj.tolist()))
    free = [(i, j) for i in range(n) for j in range(m) if (i, j) not in rated]
    return rs, ratings, mean, cov, free


@pytest.mark.parametrize('shape', [(4, 3, 2), (6, 7, 5), (9, 8, 4), (10, 10, 2), (11, 6, 9)])
def test_full_batch_kl_and_first_step(ctx, shape):
    """KL of the batch at H = 3 (same state, two different extra ratings and none) against apmf_full_kl_value on
    containers REBUILT with the extra rating, and one accepted step at a tiny learning rate against
    sym(cov - lr g), mean - lr g_mean with the gradient of apmf_full_kl_gradient on the rebuilt container
    (K = (n+m) d = 14 ... 153: one to three 64-column Cholesky panels)."""
    n, m, d = shape
    K = (n + m) * d
    rs, ratings, mean, cov, free = _problem(n, m, d, 100 * n + d)
    sig, su, sv = .8, 5., 7.
    ex = [free[0], free[-1]]
    vals = [2., 5.]
    rt = ctx.ratings(ratings, n, m)
    b = ctx.full_batch(rt, d, 3)
    dm, dc = ctx.array(mean), ctx.array(cov)
    b.set_state(dm, dc)
    b.set_extra([ex[0][0], ex[1][0], -1], [ex[0][1], ex[1][1], -1], vals + [0.])
    kls = b.kl(sig, su, sv)
    lr0 = 1e-7
    lr, kl, steps, conv, trace = b.fit(sig, su, sv, 1e-5, lr0, None, max_steps=1, trace_cap=1)
    assert list(steps) == [1, 1, 1]
    for h in range(3):
        rr = ratings if h == 2 else np.vstack([ratings, [ex[h][0], ex[h][1], vals[h]]])
        rth = ctx.ratings(rr, n, m)
        ref_kl = ctx.full_kl_value(rth, d, dm, dc, sig, su, sv)
        assert abs(kls[h] - ref_kl) <= 1e-12 * abs(ref_kl), (h, kls[h], ref_kl)
        gm, gc = ctx.empty(K), ctx.empty((K, K))
        ctx.full_kl_gradient(rth, d, dm, dc, sig, su, sv, gm, gc)
        gm, gc = gm.to_host(), gc.to_host()
        hm, hc = ctx.empty(K), ctx.empty((K, K))
        b.get_state(h, hm, hc)
        cc = cov - lr0 * gc
        scale = np.abs(gc).max()
        np.testing.assert_allclose(hc.to_host(), (cc + cc.T) / 2, rtol=0, atol=1e-9 * lr0 * scale + 1e-15)
        np.testing.assert_allclose(hm.to_host(), mean - lr0 * gm, rtol=0, atol=1e-9 * lr0 * np.abs(gm).max() + 1e-15)
        assert trace[h, 0] == kl[h] and kl[h] < kls[h]
        rth.free()
    b.free()
    rt.free()


@pytest.mark.parametrize('shape', [(6, 7, 3), (10, 10, 2), (9, 8, 8)])
def test_full_batched_equals_alone_bitwise(ctx, shape):
    """H hypotheses fitted to convergence as one batch == each fitted ALONE (a batch of one on a container rebuilt
    with its rating, no extra): states, KL, learning rates, step counts - bit for bit."""
    n, m, d = shape
    K = (n + m) * d
    rs, ratings, mean, cov, free = _problem(n, m, d, 7 * n + d)
    sig, su, sv = 1., 10., 10.
    hyps = [(free[k][0], free[k][1], float(1 + k % 5)) for k in (0, 1, len(free) // 2, -1)] + [(free[0][0], free[0][1], 5.)]
    H = len(hyps)
    rt = ctx.ratings(ratings, n, m)
    dm, dc = ctx.array(mean), ctx.array(cov)
    b = ctx.full_batch(rt, d, H)
    b.set_state(dm, dc)
    b.set_extra([h[0] for h in hyps], [h[1] for h in hyps], [h[2] for h in hyps])
    lr, kl, steps, conv, _ = b.fit(sig, su, sv, 1e-5, 1e-4, None, max_steps=60)
    tv = b.total_variance()
    assert steps.min() >= 3
    for k, (i, j, v) in enumerate(hyps):
        rth = ctx.ratings(np.vstack([ratings, [i, j, v]]), n, m)
        a = ctx.full_batch(rth, d, 1)
        a.set_state(dm, dc)
        lr1, kl1, st1, cv1, _ = a.fit(sig, su, sv, 1e-5, 1e-4, None, max_steps=60)
        assert st1[0] == steps[k] and lr1[0] == lr[k] and kl1[0] == kl[k] and cv1[0] == conv[k], (k, st1, steps[k])
        hm, hc, am, ac = ctx.empty(K), ctx.empty((K, K)), ctx.empty(K), ctx.empty((K, K))
        b.get_state(k, hm, hc)
        a.get_state(0, am, ac)
        assert np.array_equal(hm.to_host(), am.to_host()) and np.array_equal(hc.to_host(), ac.to_host()), k
        assert a.total_variance()[0] == tv[k]
        a.free()
        rth.free()
    # resuming: three calls of 20 steps == one of 60
    c = ctx.full_batch(rt, d, H)
    c.set_state(dm, dc)
    c.set_extra([h[0] for h in hyps], [h[1] for h in hyps], [h[2] for h in hyps])
    lr_c, kl_c = 1e-4, None
    total = np.zeros(H, int)
    for _ in range(3):
        lr_c, kl_c, st_c, cv_c, _ = c.fit(sig, su, sv, 1e-5, lr_c, kl_c, max_steps=20)
        total += st_c
        if cv_c.all():
            break
    done = conv | (steps == 60)
    assert np.array_equal(total[done], steps[done]) and np.array_equal(kl_c[done], kl[done])
    b.free()
    c.free()
    rt.free()


def _seed(s):
    np.random.seed(s)
    random.seed(s)


@pytest.mark.parametrize('key', ['total-variance', 'uv-entropy', '1step-ge-3.5'])
def test_full_lookahead_batched_equals_per_hypothesis(monkeypatch, key):
    """The lookahead criteria of the full-covariance class (active_pmf.py:560-704) evaluate all hypotheses of a pool
    as ONE device batch; with APMF_LOOKAHEAD_BATCH=0 every hypothesis is the reference's deepcopy -> add_rating ->
    fit_normal -> fn on its own.  Same values bit for bit (the entropy comes from the batch's own factors in one
    path and from a fresh factorisation in the other: 1e-12)."""
    import active_matrix_factorization_amd.active_pmf as full
    g = load_golden('lookahead_c1_full')
    a = full.ActivePMF(g['ratings'].copy(), latent_d=2, rating_values=(1, 2, 3, 4, 5), discrete_expectations=True)
    a.users, a.items = g['users'].copy(), g['items'].copy()
    a.mean, a.cov = g['mean'].copy(), g['cov'].copy()
    pool = sorted(a.unrated)[5:8]
    k = full.KEY_FUNCS[key]
    _seed(5)
    batched = a._get_key_vals(pool, k)
    monkeypatch.setenv('APMF_LOOKAHEAD_BATCH', '0')
    _seed(5)
    alone = a._get_key_vals(pool, k, procs=1)
    if key == 'uv-entropy':
        np.testing.assert_allclose(batched, alone, rtol=1e-12)
    else:
        assert batched == alone
    # and the reference's own value for the fixture's candidate (make_golden.py section N)
    name = {'total-variance': 'exp_total_variance', 'uv-entropy': 'exp_approx_entropy', '1step-ge-3.5': 'onestep_ge_3_5'}[key]
    cand = tuple(int(x) for x in g['cand'])
    monkeypatch.delenv('APMF_LOOKAHEAD_BATCH')
    got = a._get_key_vals([cand], k)[0]
    np.testing.assert_allclose(got, float(g[name]), rtol=1e-10)
