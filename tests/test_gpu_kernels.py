"""Parity of the HIP path (through the C-ABI, libapmf_hip.so) against the CPU oracle and the
golden vectors produced by the real reference.  Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

from conftest import load_golden, c2_inputs
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps


@pytest.fixture(scope='module')
def ctx():
    from active_matrix_factorization_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _pmf_kw(g):
    sig, sig_u, sig_v = (float(x) for x in g['sigma'])
    return dict(sig=sig, sig_u=sig_u, sig_v=sig_v, subtract_mean=bool(g['subtract_mean']),
                mean_rating=float(g['mean_rating']))


def _grad(ctx, ratings, U, V, kw):
    n, d = U.shape
    m = V.shape[0]
    rt = ctx.ratings(ratings, n, m)
    dU, dV = ctx.array(U), ctx.array(V)
    gU, gV = ctx.empty((n, d)), ctx.empty((m, d))
    ctx.pmf_gradient(rt, dU, dV, d, kw['sig'], kw['sig_u'], kw['sig_v'], kw['subtract_mean'],
                     kw['mean_rating'], gU, gV)
    ll, sq = ctx.pmf_log_likelihood(rt, dU, dV, d, kw['sig'], kw['sig_u'], kw['sig_v'],
                                    kw['subtract_mean'], kw['mean_rating'])
    return gU.to_host(), gV.to_host(), ll, sq


PMF_CASES = ['pmf_c1_10x10_d2', 'pmf_50x40_d3', 'pmf_50x40_d3_submean', 'pmf_30x70_d16']


# ---------------------------------------------------------------------------- K1
@pytest.mark.parametrize('name', PMF_CASES)
def test_pmf_gradient_ll_golden(ctx, name):
    g = load_golden(name)
    kw = _pmf_kw(g)
    gU, gV, ll, sq = _grad(ctx, g['ratings'], g['U0'], g['V0'], kw)
    # float64, fixed summation tree; tolerance = reordering of <= a few hundred adds
    np.testing.assert_allclose(gU, g['gU'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(gV, g['gV'], rtol=1e-12, atol=1e-12)
    assert abs(ll - g['ll']) <= 1e-12 * abs(g['ll'])
    assert abs(sq - orc.pmf_sq_error(g['ratings'], g['U0'], g['V0'], kw['subtract_mean'],
                                     kw['mean_rating'])) <= 1e-12 * sq
    # a minibatch slice (pmf_cy.pyx:335-336)
    lo, hi = int(g['batch_lo']), int(g['batch_hi'])
    gUb, gVb, _, _ = _grad(ctx, g['ratings'][lo:hi], g['U0'], g['V0'], kw)
    np.testing.assert_allclose(gUb, g['gU_batch'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(gVb, g['gV_batch'], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('d', [1, 2, 5, 8, 10, 15, 16, 17, 33, 64])
def test_pmf_gradient_all_d(ctx, d):
    rs = np.random.RandomState(d)
    n, m, N = 37, 29, 400
    ii, jj = rs.randint(0, n, N), rs.randint(0, m, N)
    ratings = np.column_stack([ii, jj, rs.normal(size=N)]).astype(float)
    ratings[0, :2] = [n - 1, m - 1]
    U, V = rs.normal(size=(n, d)), rs.normal(size=(m, d))
    kw = dict(sig=0.8, sig_u=3., sig_v=5., subtract_mean=True, mean_rating=0.3)
    gU, gV, ll, _ = _grad(ctx, ratings, U, V, kw)
    oU, oV = orc.pmf_gradient(ratings, U, V, **kw)
    np.testing.assert_allclose(gU, oU, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(gV, oV, rtol=1e-11, atol=1e-11)
    assert abs(ll - orc.pmf_log_likelihood(ratings, U, V, **kw)) <= 1e-12 * abs(ll)


def test_pmf_empty_rows_and_no_ratings(ctx):
    # users/items without ratings only get the prior term (pmf_cy.pyx:214-215)
    U, V = np.arange(12.).reshape(4, 3), np.arange(15.).reshape(5, 3) / 7
    ratings = np.array([[1, 3, 2.5]])
    kw = dict(sig=1., sig_u=10., sig_v=10., subtract_mean=False, mean_rating=0.)
    gU, gV, ll, _ = _grad(ctx, ratings, U, V, kw)
    oU, oV = orc.pmf_gradient(ratings, U, V, **kw)
    np.testing.assert_allclose(gU, oU, rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(gV, oV, rtol=1e-14, atol=1e-14)
    gU, gV, ll, sq = _grad(ctx, np.zeros((0, 3)), U, V, kw)
    np.testing.assert_allclose(gU, -U / 10., rtol=1e-15)
    assert sq == 0.


def test_pmf_bad_ratings(ctx):
    with pytest.raises(ValueError):
        ctx.ratings(np.array([[5, 0, 1.]]), 3, 3)
    with pytest.raises(TypeError):
        ctx.ratings(np.zeros((3, 2)), 3, 3)


@pytest.mark.parametrize('name', PMF_CASES)
def test_pmf_fit_lls_golden(ctx, name):
    g = load_golden(name)
    kw = _pmf_kw(g)
    n, d = g['U0'].shape
    m = g['V0'].shape[0]
    rt = ctx.ratings(g['ratings'], n, m)
    dU, dV = ctx.array(g['U0']), ctx.array(g['V0'])
    trace, steps, lr, conv = ctx.pmf_fit_lls(rt, dU, dV, d, kw['sig'], kw['sig_u'], kw['sig_v'],
                                             kw['subtract_mean'], kw['mean_rating'], 1e-4, 1e-10, 1e-2)
    # same number of accepted steps = same accept/reject control flow as pmf_cy.pyx:265-291
    assert steps == g['ll_trace'].size and conv
    np.testing.assert_allclose(trace, g['ll_trace'], rtol=1e-10)
    np.testing.assert_allclose(dU.to_host(), g['U_fit'], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(dV.to_host(), g['V_fit'], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize('shape', ['skewed', 'many_rows', 'uniform_wpr4'])
def test_pmf_fit_lls_block_shapes(ctx, shape, monkeypatch):
    """The one-launch search iteration against the oracle on the layouts its two block shapes meet:
    a few very long rows next to empty ones (4 waves per row), more rows than the grid cap covers
    in one trip (the blocks stride), and the 4-waves-per-row shape forced on short rows."""
    rs = np.random.RandomState(11)
    if shape == 'skewed':
        n, m, d = 40, 700, 5
        pairs = {(0, j) for j in range(650)} | {(1, j) for j in range(0, 700, 2)}
        pairs |= {(int(i), int(j)) for i, j in zip(rs.randint(2, n - 3, 300), rs.randint(0, m - 10, 300))}
    elif shape == 'many_rows':
        n, m, d = 3000, 2500, 3
        pairs = {(int(i), int(j)) for i, j in zip(rs.randint(0, n, 9000), rs.randint(0, m, 9000))}
    else:
        n, m, d = 60, 45, 10
        pairs = {(int(i), int(j)) for i, j in zip(rs.randint(0, n, 900), rs.randint(0, m, 900))}
        monkeypatch.setenv('APMF_K1_WPR', '4')
    pairs = np.array(sorted(pairs))
    ratings = np.column_stack([pairs, rs.randint(1, 6, len(pairs))]).astype(float)
    U0, V0 = rs.normal(size=(n, d)), rs.normal(size=(m, d))
    kw = dict(sig=.7, sig_u=10., sig_v=5., subtract_mean=True, mean_rating=float(ratings[:, 2].mean()))
    steps = 25
    Uo, Vo, tro = orc.pmf_fit_lls(ratings, U0, V0, lr=1e-4, max_steps=steps, **kw)
    rt = ctx.ratings(ratings, n, m)
    dU, dV = ctx.array(U0), ctx.array(V0)
    trace, nst, lr, conv = ctx.pmf_fit_lls(rt, dU, dV, d, kw['sig'], kw['sig_u'], kw['sig_v'], True,
                                           kw['mean_rating'], 1e-4, 1e-10, 1e-2, max_steps=steps)
    assert nst == len(tro)
    np.testing.assert_allclose(trace, tro, rtol=1e-11)
    np.testing.assert_allclose(dU.to_host(), Uo, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dV.to_host(), Vo, rtol=1e-10, atol=1e-13)
    # bit-reproducible run to run
    dU2, dV2 = ctx.array(U0), ctx.array(V0)
    trace2, *_ = ctx.pmf_fit_lls(rt, dU2, dV2, d, kw['sig'], kw['sig_u'], kw['sig_v'], True,
                                 kw['mean_rating'], 1e-4, 1e-10, 1e-2, max_steps=steps)
    assert np.array_equal(trace, trace2) and np.array_equal(dU.to_host(), dU2.to_host())


def test_pmf_fit_lls_resume(ctx):
    """max_steps + returned lr let a caller drive the generator step by step (fit_lls yields)."""
    g = load_golden('pmf_50x40_d3')
    kw = _pmf_kw(g)
    n, d = g['U0'].shape
    m = g['V0'].shape[0]
    rt = ctx.ratings(g['ratings'], n, m)
    dU, dV = ctx.array(g['U0']), ctx.array(g['V0'])
    lr, out, conv = 1e-4, [], False
    while not conv and len(out) < 10000:
        trace, steps, lr, conv = ctx.pmf_fit_lls(rt, dU, dV, d, kw['sig'], kw['sig_u'], kw['sig_v'],
                                                 False, 0., lr, 1e-10, 1e-2, max_steps=7)
        out.extend(trace)
    assert len(out) == g['ll_trace'].size
    np.testing.assert_allclose(out, g['ll_trace'], rtol=1e-10)


def test_pmf_headline_shape(ctx):
    g = load_golden('pmf_c2_943x1682_d10')
    ratings, U0, V0 = c2_inputs()
    kw = dict(sig=1., sig_u=10., sig_v=10., subtract_mean=False, mean_rating=0.)
    gU, gV, ll, _ = _grad(ctx, ratings, U0, V0, kw)
    np.testing.assert_allclose(gU, g['gU'], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(gV, g['gV'], rtol=1e-11, atol=1e-11)
    assert abs(ll - g['ll']) <= 1e-12 * abs(g['ll'])
    # run-to-run bit reproducibility (fixed summation tree, no atomics)
    gU2, gV2, ll2, _ = _grad(ctx, ratings, U0, V0, kw)
    assert np.array_equal(gU, gU2) and np.array_equal(gV, gV2) and ll == ll2


def test_pmf_headline_fit_trace(ctx):
    """The one-launch search iteration at BASELINE.json configs[1]'s shape against the REAL reference:
    20 accepted steps of fit_lls from the recorded starting point - same accept / reject sequence
    (the trace has exactly the reference's values), same factors."""
    g = load_golden('pmf_c2_fit_943x1682_d10')
    ratings, U0, V0 = c2_inputs()
    n, m, d = 943, 1682, 10
    rt = ctx.ratings(ratings, n, m)
    dU, dV = ctx.array(U0), ctx.array(V0)
    trace, steps, lr, conv = ctx.pmf_fit_lls(rt, dU, dV, d, 1., 10., 10., False, 0., float(g['learning_rate']),
                                             1e-10, 1e-2, max_steps=20)
    assert steps == 20
    np.testing.assert_allclose(trace, g['ll_trace'], rtol=1e-11)
    U, V = dU.to_host(), dV.to_host()
    np.testing.assert_allclose(U[g['rows_u']], g['U_rows'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(V[g['rows_v']], g['V_rows'], rtol=1e-12, atol=1e-13)
    assert abs(U.sum() - g['U_sum']) <= 1e-9 * abs(g['U_sum']) and abs(V.sum() - g['V_sum']) <= 1e-9 * abs(g['V_sum'])


def test_pmf_minibatch_epoch(ctx):
    rs = np.random.RandomState(5)
    n, m, d, N, bs = 23, 31, 6, 500, 64
    ratings = np.column_stack([rs.randint(0, n, N), rs.randint(0, m, N), rs.normal(size=N)]).astype(float)
    U, V = rs.normal(size=(n, d)), rs.normal(size=(m, d))
    ui, vi = rs.normal(size=(n, d)) * .01, rs.normal(size=(m, d)) * .01
    kw = dict(sig=1., sig_u=10., sig_v=10., subtract_mean=False, mean_rating=0.)
    dU, dV, dui, dvi = ctx.array(U), ctx.array(V), ctx.array(ui), ctx.array(vi)
    ctx.pmf_minibatch_epoch(ratings, n, m, bs, 1.0, .8, dU, dV, dui, dvi, d, 1., 10., 10., False, 0.)
    # reference recipe, pmf_cy.pyx:329-344 (float lr / int n, float momentum)
    mom = float(np.float32(.8))
    for s in range(0, N, bs):
        b = ratings[s:s + bs]
        gU, gV = orc.pmf_gradient(b, U, V, **kw)
        sc = float(np.float32(1.0) / np.float32(b.shape[0]))
        ui = ui * mom + gU * sc
        U = U + ui
        vi = vi * mom + gV * sc
        V = V + vi
    np.testing.assert_allclose(dU.to_host(), U, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(dV.to_host(), V, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(dui.to_host(), ui, rtol=1e-11, atol=1e-12)


def test_predicted_matrix(ctx):
    g = load_golden('pmf_50x40_d3_submean')
    n, d = g['U0'].shape
    m = g['V0'].shape[0]
    from active_matrix_factorization_amd._lib import grid_ld
    ld = grid_ld(m)
    out = ctx.empty((n, ld))
    ctx.pmf_predicted_matrix(ctx.array(g['U0']), ctx.array(g['V0']), n, m, d, float(g['mean_rating']),
                             out, ld)
    np.testing.assert_allclose(ctx.download_grid(out, m), g['pred0'], rtol=1e-13, atol=1e-13)


# ---------------------------------------------------------------------------- K2 (matrix normal)
def _mn_state(ctx, mean, R, Om, dtype, n, m):
    cross = ctx.upload_grid(R[:n, n:], dtype)
    return (ctx.array(mean, dtype), cross, ctx.array(np.diag(R).copy(), dtype), ctx.array(Om, dtype),
            cross.shape[1])


def _mn_run(ctx, mean, R, Om, n, m, dtype, mask=None, want=1):
    d = mean.shape[1]
    dm, dc, dd, do, ld = _mn_state(ctx, mean, R, Om, dtype, n, m)
    var, pm = ctx.empty((n, ld), dtype), ctx.empty((n, ld), dtype)
    dsum, bv, bi = ctx.empty(1, np.float64), ctx.empty(1, dtype), ctx.empty(1, np.int64)
    dmask = None
    if mask is not None:
        dmask = ctx.upload_grid(mask.astype(np.uint8), np.uint8, ld=ld)
    ctx.mn_score_grid(dtype, 1, n, m, d, ld, dm, dc, dd, do, mask=dmask, var=var, pmean=pm, dsum=dsum,
                      want=want, best_val=bv, best_idx=bi)
    return (ctx.download_grid(var, m), ctx.download_grid(pm, m), dsum.to_host()[0], bv.to_host()[0],
            int(bi.to_host()[0]))


MN_CASES = ['mn_c1_10x10_d2_eye', 'mn_c1_10x10_d2_rand', 'mn_24x17_d5_rand', 'mn_12x20_d10_rand']


@pytest.mark.parametrize('name', MN_CASES)
def test_mn_grid_f64_golden(ctx, name):
    g = load_golden(name)
    n, m = g['p_mean'].shape
    mask = np.zeros((n, m), bool)
    mask[tuple(g['pool'].T)] = True
    var, pm, tot, bv, bi = _mn_run(ctx, g['mean'], g['cov_useritems'], g['cov_latents'], n, m,
                                   np.float64, mask=mask)
    # The reference forms E[(u.v)^2] - mean^2 (mn_active_pmf.py:311-315): its own rounding
    # floor is a few eps * E2.  The closed form evaluated here has no cancellation.
    scale = np.abs(g['e2'])
    assert np.all(np.abs(var - g['p_var']) <= 64 * EPS * scale)
    assert np.all(np.abs(pm - g['p_mean']) <= 64 * EPS * (np.abs(g['p_mean']) + np.sqrt(scale)))
    # the reference's selection: first max over the unrated pool (mn_active_pmf.py:728-729)
    assert (bi // m, bi % m) == tuple(g['pick_max'])
    assert abs(bv - g['pool_vals'].max()) <= 64 * EPS * scale.max()
    assert abs(tot - g['pool_vals'].sum()) <= 1e-12 * abs(g['pool_vals'].sum())
    # no mask -> _total_variance over the whole grid (mn_active_pmf.py:597-598)
    _, _, tot_all, _, _ = _mn_run(ctx, g['mean'], g['cov_useritems'], g['cov_latents'], n, m,
                                  np.float64)
    assert abs(tot_all - g['total_variance']) <= 1e-12 * abs(g['total_variance'])


@pytest.mark.parametrize('name', MN_CASES)
def test_mn_grid_f32_golden(ctx, name):
    g = load_golden(name)
    n, m = g['p_mean'].shape
    mask = np.zeros((n, m), bool)
    mask[tuple(g['pool'].T)] = True
    var, pm, tot, bv, bi = _mn_run(ctx, g['mean'], g['cov_useritems'], g['cov_latents'], n, m,
                                   np.float32, mask=mask)
    # stated f32 tolerance: 1e-5 relative to the magnitude of the terms being summed
    scale = np.abs(g['e2'])
    assert np.all(np.abs(var - g['p_var']) <= 1e-5 * np.maximum(scale, np.abs(g['p_var'])))
    assert np.all(np.abs(pm - g['p_mean']) <= 1e-5 * (np.abs(g['p_mean']) + np.sqrt(scale)))
    assert abs(tot - g['pool_vals'].sum()) <= 1e-5 * abs(g['pool_vals'].sum())
    # exact selection = f32 sweep + f64 re-score of the near-extremum short list
    i, j = bi // m, bi % m
    best_ref = g['pool_vals'].max()
    assert g['p_var'][i, j] >= best_ref - 1e-5 * abs(best_ref)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('shape', [(1, 1, 1), (3, 63, 2), (5, 64, 3), (7, 65, 4), (64, 257, 10),
                                   (65, 300, 15), (130, 70, 16), (9, 40, 24), (11, 33, 32),
                                   (6, 130, 47), (4, 20, 64)])
def test_mn_grid_shapes_vs_oracle(ctx, dtype, shape):
    n, m, d = shape
    rs = np.random.RandomState(n * 1000 + m + d)
    mean = rs.normal(size=(n + m, d))
    a = rs.normal(size=(n + m, n + m)) / np.sqrt(n + m)
    R = a @ a.T + 0.1 * np.eye(n + m)
    b = rs.normal(size=(d, d)) / np.sqrt(d)
    Om = b @ b.T + 0.1 * np.eye(d)
    mask = rs.rand(n, m) < 0.7
    mask[0, 0] = True
    for want in (1, -1):
        var, pm, tot, bv, bi = _mn_run(ctx, mean, R, Om, n, m, dtype, mask=mask, want=want)
        opm, opv, oe2 = orc.mn_pred_means_vars(n, m, mean, R, Om, want_e2=True)
        tol = 1e-5 if dtype == np.float32 else 1e-12
        scale = np.maximum(np.abs(oe2), np.abs(opv))
        assert np.all(np.abs(var - opv) <= tol * scale)
        assert np.all(np.abs(pm - opm) <= tol * (np.abs(opm) + np.sqrt(scale)))
        assert abs(tot - opv[mask].sum()) <= tol * np.abs(opv[mask]).sum()
        cand = np.where(mask, opv, -np.inf if want > 0 else np.inf)
        ref = cand.max() if want > 0 else cand.min()
        i, j = bi // m, bi % m
        assert mask[i, j]
        assert abs(opv[i, j] - ref) <= 2 * tol * scale.max()
        if dtype == np.float64:
            flat = int(np.argmax(cand)) if want > 0 else int(np.argmin(cand))
            assert bi == flat or abs(opv[i, j] - ref) <= 64 * EPS * scale.max()


def test_mn_grid_batched_states(ctx):
    """B posterior states per launch == B single launches (what the lookahead batches)."""
    rs = np.random.RandomState(3)
    B, n, m, d = 5, 33, 70, 10
    from active_matrix_factorization_amd._lib import grid_ld
    ld = grid_ld(m)
    means = rs.normal(size=(B, n + m, d))
    cross = np.zeros((B, n, ld))
    cross[:, :, :m] = rs.normal(size=(B, n, m)) * .1
    diag = rs.rand(B, n + m) + .5
    om = np.array([(lambda x: x @ x.T / d + .1 * np.eye(d))(rs.normal(size=(d, d))) for _ in range(B)])
    for dtype in (np.float32, np.float64):
        dm, dc, dd, do = (ctx.array(x, dtype) for x in (means, cross, diag, om))
        var = ctx.empty((B, n, ld), dtype)
        dsum, bv, bi = ctx.empty(B, np.float64), ctx.empty(B, dtype), ctx.empty(B, np.int64)
        ctx.mn_score_grid(dtype, B, n, m, d, ld, dm, dc, dd, do, var=var, dsum=dsum, want=-1,
                          best_val=bv, best_idx=bi)
        v = ctx.download_grid(var, m)
        sums, bis = dsum.to_host(), bi.to_host()
        for b in range(B):
            R = np.zeros((n + m, n + m))
            R[np.arange(n + m), np.arange(n + m)] = diag[b]
            R[:n, n:] = cross[b, :, :m]
            R[n:, :n] = cross[b, :, :m].T
            _, opv = orc.mn_pred_means_vars(n, m, means[b], R, om[b])
            tol = 1e-5 if dtype == np.float32 else 1e-12
            assert np.all(np.abs(v[b] - opv) <= tol * np.abs(opv).max())
            assert abs(sums[b] - opv.sum()) <= tol * np.abs(opv).sum()
            i, j = int(bis[b]) // m, int(bis[b]) % m
            assert abs(opv[i, j] - opv.min()) <= 2 * tol * np.abs(opv).max()


@pytest.mark.parametrize('kernel', ['valu', 'mfma'])
@pytest.mark.parametrize('shape', [(5, 64, 3), (65, 300, 10), (130, 70, 16), (97, 257, 24), (33, 130, 64)])
def test_mn_grid_f32_kernel_variants(ctx, monkeypatch, kernel, shape):
    """Both float32 sweeps (streaming VALU kernel, matrix-core kernel) at every rank bucket:
    values against the oracle, fused sum / arg-extremum against the kernel's OWN written grid
    (first occurrence in flat order, exact), with and without the grid output, and a state full
    of exact ties (identical users, zero cross-covariance)."""
    monkeypatch.setenv('APMF_K2_KERNEL', kernel)
    n, m, d = shape
    rs = np.random.RandomState(n + 7 * m + d)
    from active_matrix_factorization_amd._lib import grid_ld
    ld = grid_ld(m)
    b = rs.normal(size=(d, d)) / np.sqrt(d)
    Om = b @ b.T + 0.1 * np.eye(d)
    mask = rs.rand(n, m) < 0.8
    mask[-1, -1] = True
    dmask = ctx.upload_grid(mask.astype(np.uint8), np.uint8, ld=ld)
    for ties in (False, True):
        mean = rs.normal(size=(n + m, d))
        a = rs.normal(size=(n + m, n + m)) / np.sqrt(n + m)
        R = a @ a.T + 0.1 * np.eye(n + m)
        if ties:
            mean[:n] = mean[0]
            mean[n:] = mean[n]
            R = np.eye(n + m) * 1.5
        dm, dc, dd, do, _ = _mn_state(ctx, mean, R, Om, np.float32, n, m)
        _, opv, oe2 = orc.mn_pred_means_vars(n, m, mean, R, Om, want_e2=True)
        scale = np.maximum(np.abs(oe2), np.abs(opv))
        for want in (1, -1):
            var = ctx.empty((n, ld), np.float32)
            dsum, bv, bi = ctx.empty(1), ctx.empty(1, np.float32), ctx.empty(1, np.int64)
            ctx.mn_score_grid(np.float32, 1, n, m, d, ld, dm, dc, dd, do, mask=dmask, var=var, dsum=dsum,
                              want=want, best_val=bv, best_idx=bi)
            v = ctx.download_grid(var, m)
            assert np.all(np.abs(v - opv) <= 1e-5 * scale)
            cand = np.where(mask, v, -np.inf if want > 0 else np.inf)
            flat = int(np.argmax(cand)) if want > 0 else int(np.argmin(cand))
            assert int(bi.to_host()[0]) == flat
            assert bv.to_host()[0] == cand.ravel()[flat]
            tot = v[mask].astype(np.float64).sum()
            assert abs(dsum.to_host()[0] - tot) <= 1e-6 * abs(tot)
            # reductions only (no grid written): same selection, same total
            dsum2, bv2, bi2 = ctx.empty(1), ctx.empty(1, np.float32), ctx.empty(1, np.int64)
            ctx.mn_score_grid(np.float32, 1, n, m, d, ld, dm, dc, dd, do, mask=dmask, dsum=dsum2,
                              want=want, best_val=bv2, best_idx=bi2)
            if kernel == 'valu' or d >= 13:     # same kernel as above: bit-identical
                assert int(bi2.to_host()[0]) == flat and bv2.to_host()[0] == bv.to_host()[0]
            else:
                i2, j2 = divmod(int(bi2.to_host()[0]), m)
                assert mask[i2, j2] and abs(opv[i2, j2] - opv.ravel()[flat]) <= 2e-5 * scale.max()
            assert abs(dsum2.to_host()[0] - tot) <= 1e-5 * abs(tot)


def test_mn_pool_and_exact_selection(ctx):
    """f32 sweep -> near-extremum short list -> f64 re-score = the f64 argmax (exact index)."""
    g = load_golden('mn_24x17_d5_rand')
    n, m = g['p_mean'].shape
    d = int(g['d'])
    R, Om, mean = g['cov_useritems'], g['cov_latents'], g['mean']
    mask = np.zeros((n, m), bool)
    mask[tuple(g['pool'].T)] = True
    dm, dc, dd, do, ld = _mn_state(ctx, mean, R, Om, np.float32, n, m)
    dmask = ctx.upload_grid(mask.astype(np.uint8), np.uint8, ld=ld)
    var, bv, bi = ctx.empty((n, ld), np.float32), ctx.empty(1, np.float32), ctx.empty(1, np.int64)
    ctx.mn_score_grid(np.float32, 1, n, m, d, ld, dm, dc, dd, do, mask=dmask, var=var, want=1,
                      best_val=bv, best_idx=bi)
    cap = 64
    pi, pj = ctx.empty(cap, np.int32), ctx.empty(cap, np.int32)
    cnt = ctx.grid_collect_near(n, m, ld, var, dmask, 1, float(bv.to_host()[0]), 1e-4, pi, pj, cap)
    assert 1 <= cnt <= cap
    dm64, dc64, dd64, do64, _ = _mn_state(ctx, mean, R, Om, np.float64, n, m)
    pm, pv = ctx.empty(cnt), ctx.empty(cnt)
    ctx.mn_score_pool(n, m, d, ld, dm64, dc64, dd64, do64, pi, pj, cnt, pm, pv)
    ii, jj, vals = pi.to_host()[:cnt], pj.to_host()[:cnt], pv.to_host()
    k = int(np.argmax(vals))
    assert (int(ii[k]), int(jj[k])) == tuple(g['pick_max'])
    scale = np.abs(g['e2']).max()
    assert np.all(np.abs(vals - g['p_var'][ii, jj]) <= 64 * EPS * scale)
    # whole pool through the pool kernel, in the reference's pool order; first-max tie rule
    P = g['pool'].shape[0]
    dpi, dpj = ctx.array(g['pool'][:, 0], np.int32), ctx.array(g['pool'][:, 1], np.int32)
    pm, pv = ctx.empty(P), ctx.empty(P)
    ctx.mn_score_pool(n, m, d, ld, dm64, dc64, dd64, do64, dpi, dpj, P, pm, pv)
    np.testing.assert_allclose(pv.to_host(), g['pool_vals'], rtol=0, atol=64 * EPS * scale)
    pos, val = ctx.pool_argext(pv, P, 1)
    assert tuple(g['pool'][pos]) == tuple(g['pick_max'])


def test_pool_argext_tie_rule(ctx):
    v = np.array([1., 5., 3., 5., -2., -2., np.nan, 5.])
    d = ctx.array(v)
    assert ctx.pool_argext(d, v.size, 1)[0] == 1      # first max, like python max()/np.argmax
    assert ctx.pool_argext(d, v.size, -1)[0] == 4     # first min
    big = np.zeros(300000)
    big[[123456, 250000]] = 7.
    assert ctx.pool_argext(ctx.array(big), big.size, 1)[0] == 123456
    with pytest.raises(ValueError):                   # empty pool: mn_active_pmf.py:723-724
        ctx.pool_argext(d, 0, 1)


def test_grid_gather(ctx):
    rs = np.random.RandomState(0)
    n, m = 13, 70
    from active_matrix_factorization_amd._lib import grid_ld
    ld = grid_ld(m)
    host = rs.normal(size=(n, m))
    g = ctx.upload_grid(host)
    pi, pj = rs.randint(0, n, 500).astype(np.int32), rs.randint(0, m, 500).astype(np.int32)
    out = ctx.empty(500)
    ctx.grid_gather(g, ld, ctx.array(pi), ctx.array(pj), 500, out)
    np.testing.assert_array_equal(out.to_host(), host[pi, pj])


def test_mn_grid_headline_size_properties(ctx):
    """943 x 1682, d = 10: f32 sweep vs f64 sweep of the same state, fused reductions vs the
    written grid (size-independent properties; the oracle would take minutes here)."""
    rs = np.random.RandomState(11)
    n, m, d, B = 943, 1682, 10, 3
    from active_matrix_factorization_amd._lib import grid_ld
    ld = grid_ld(m)
    means = rs.normal(0, 1, (B, n + m, d))
    cross = np.zeros((B, n, ld))
    cross[:, :, :m] = rs.normal(0, .05, (B, n, m))
    diag = rs.rand(B, n + m) + .5
    om = np.array([(lambda x: x @ x.T / d + .1 * np.eye(d))(rs.normal(size=(d, d))) for _ in range(B)])
    mask = rs.rand(n, m) < 0.9
    dmask = ctx.upload_grid(mask.astype(np.uint8), np.uint8, ld=ld)
    res = {}
    for dtype in (np.float32, np.float64):
        dm, dc, dd, do = (ctx.array(x, dtype) for x in (means, cross, diag, om))
        var = ctx.empty((B, n, ld), dtype)
        dsum, bv, bi = ctx.empty(B, np.float64), ctx.empty(B, dtype), ctx.empty(B, np.int64)
        ctx.mn_score_grid(dtype, B, n, m, d, ld, dm, dc, dd, do, mask=dmask, var=var, dsum=dsum,
                          want=1, best_val=bv, best_idx=bi)
        res[dtype] = (ctx.download_grid(var, m), dsum.to_host(), bv.to_host(), bi.to_host())
    v32, s32, b32, i32 = res[np.float32]
    v64, s64, b64, i64 = res[np.float64]
    assert np.all(np.abs(v32 - v64) <= 1e-5 * np.abs(v64).max())
    for b in range(B):
        cand = np.where(mask, v64[b], -np.inf)
        assert int(i64[b]) == int(np.argmax(cand))             # first max, flat order
        assert b64[b] == cand.max()
        assert abs(s64[b] - v64[b][mask].sum()) <= 1e-12 * v64[b][mask].sum()
        assert abs(s32[b] - s64[b]) <= 1e-5 * s64[b]
        assert cand.ravel()[int(i32[b])] >= cand.max() * (1 - 1e-5)
    # spot-check a few rows against the oracle
    b = 1
    R = np.zeros((n + m, n + m))
    R[np.arange(n + m), np.arange(n + m)] = diag[b]
    R[:n, n:] = cross[b, :, :m]
    R[n:, :n] = cross[b, :, :m].T
    ov = orc.mn_score_rows(n, m, means[b], R, om[b], 500, 503)
    assert np.all(np.abs(v64[b][500:503] - ov) <= 1e-11 * np.abs(ov).max())


# ---------------------------------------------------------------------------- K2-full
@pytest.mark.parametrize('name', ['full_6x5_d3', 'full_12x9_d5', 'full_11x14_d10', 'full_7x9_d16'])
def test_full_grid_golden(ctx, name):
    g = load_golden(name)
    n, m = g['p_mean'].shape
    d = int(g['d'])
    from active_matrix_factorization_amd._lib import grid_ld
    ld = grid_ld(m)
    mask = np.zeros((n, m), bool)
    mask[tuple(g['pool'].T)] = True
    dmask = ctx.upload_grid(mask.astype(np.uint8), np.uint8, ld=ld)
    var, pm = ctx.empty((n, ld)), ctx.empty((n, ld))
    dsum, bv, bi = ctx.empty(1), ctx.empty(1), ctx.empty(1, np.int64)
    ctx.full_score_grid(n, m, d, ld, ctx.array(g['mean']), ctx.array(g['cov']), mask=dmask, var=var,
                        pmean=pm, dsum=dsum, want=1, best_val=bv, best_idx=bi)
    _, _, e2 = orc.full_pred_means_vars(n, m, d, g['mean'], g['cov'], want_e2=True)
    v, p = ctx.download_grid(var, m), ctx.download_grid(pm, m)
    assert np.all(np.abs(v - g['p_var']) <= 256 * EPS * np.abs(e2))
    assert np.all(np.abs(p - g['p_mean']) <= 256 * EPS * (np.abs(g['p_mean']) + np.sqrt(np.abs(e2))))
    b = int(bi.to_host()[0])
    assert (b // m, b % m) == tuple(g['pick_max'])
    assert abs(dsum.to_host()[0] - g['pool_vals'].sum()) <= 1e-11 * abs(g['pool_vals'].sum())


@pytest.mark.parametrize('shape', [(9, 17, 9), (23, 70, 10), (8, 16, 10), (1, 15, 12), (17, 65, 13), (10, 129, 16),
                                   (33, 31, 11), (64, 64, 10)])
def test_full_grid_quad_kernel(ctx, shape, monkeypatch):
    """The sweep for 9 <= d <= 16 (k_full_sweep_q: four lanes per candidate, cross block contracted from LDS; d = 10
    is the headline rank) against the oracle's restatement of normal_exps_cy.exp_dotprod_sq (normal_exps_cy.pyx:111-135)
    and against the general one-thread-per-candidate kernel: values, masked sum, arg-max and arg-min; item counts
    that leave short and padding-only groups of 16, row counts that leave a short last group of 8."""
    from active_matrix_factorization_amd._lib import grid_ld
    n, m, d = shape
    rs = np.random.RandomState(n * 1000 + m * 10 + d)
    K = (n + m) * d
    a = rs.normal(size=(K, K)) / np.sqrt(K)
    cov = a @ a.T + .05 * np.eye(K)
    mean = rs.normal(size=K)
    ld = grid_ld(m)
    mask = rs.rand(n, m) < .7
    mask[0, 0] = True
    dmask = ctx.upload_grid(mask.astype(np.uint8), np.uint8, ld=ld)
    dm, dc = ctx.array(mean), ctx.array(cov)
    pm0, pv0, e2 = orc.full_pred_means_vars(n, m, d, mean, cov, want_e2=True)
    res = {}
    for kern in ('quad', 'general'):
        if kern == 'general':
            monkeypatch.setenv('APMF_K2FULL_KERNEL', 'general')
        for want in (1, -1):
            var, pm = ctx.empty((n, ld)), ctx.empty((n, ld))
            dsum, bv, bi = ctx.empty(1), ctx.empty(1), ctx.empty(1, np.int64)
            ctx.full_score_grid(n, m, d, ld, dm, dc, mask=dmask, var=var, pmean=pm, dsum=dsum, want=want, best_val=bv,
                                best_idx=bi)
            v, p = ctx.download_grid(var, m), ctx.download_grid(pm, m)
            assert np.all(np.abs(v - pv0) <= 256 * EPS * np.abs(e2)), kern
            assert np.all(np.abs(p - pm0) <= 256 * EPS * (np.abs(pm0) + np.sqrt(np.abs(e2)))), kern
            assert not var.to_host()[:, m:].any() and not pm.to_host()[:, m:].any()        # padding written as zeros
            b = int(bi.to_host()[0])
            cand = np.where(mask, v, -np.inf if want > 0 else np.inf)
            k = int(np.argmax(cand)) if want > 0 else int(np.argmin(cand))
            assert b == k and bv.to_host()[0] == v.ravel()[k], (kern, want)
            assert abs(dsum.to_host()[0] - v[mask].sum()) <= 1e-12 * abs(v[mask].sum())
            res[kern, want] = (b, v)
        # sum only (no grid written)
        dsum = ctx.empty(1)
        ctx.full_score_grid(n, m, d, ld, dm, dc, dsum=dsum)
        assert abs(dsum.to_host()[0] - pv0.sum()) <= 1e-11 * abs(pv0.sum())
    assert res['quad', 1][0] == res['general', 1][0] and res['quad', -1][0] == res['general', -1][0]


# ---------------------------------------------------------------------------- K3 / K4
BAYES_CASES = ['bayes_20x15_d4', 'bayes_c1_10x10_d2', 'bayes_100x90_d15']


@pytest.mark.parametrize('name', BAYES_CASES)
def test_gibbs_half_sweep(ctx, name):
    g = load_golden(name)
    n, d = g['users'].shape
    m = g['items'].shape[0]
    sub = float(g['mean_rating']) if int(g['subtract_mean']) else 0.
    rt = ctx.ratings(g['ratings'], n, m)
    rs = np.random.RandomState(1)
    for side, rows, oth in ((0, n, g['items']), (1, m, g['users'])):
        z = rs.normal(size=(rows, d))
        out = ctx.empty((rows, d))
        ctx.gibbs_half_sweep(rt, side, ctx.array(oth), d, ctx.array(g['sf_alpha']),
                             ctx.array(g['sf_mu']), 2.0, sub, ctx.array(z), out)
        ptr, idx, rat = orc.csr_by_row(g['ratings'], side, rows)
        ref = orc.gibbs_half_sweep(ptr, idx, rat, oth, g['sf_alpha'], g['sf_mu'], 2.0, sub, z)
        # Cholesky-based inverse here vs LU in numpy/the oracle: rounding only
        np.testing.assert_allclose(out.to_host(), ref, rtol=2e-10, atol=1e-12)
    # the recorded sample_feature call of the reference (bayes_pmf.py:189-216), user 0
    z = np.zeros((n, d))
    z[0] = g['sf_z']
    out = ctx.empty((n, d))
    ctx.gibbs_half_sweep(rt, 0, ctx.array(g['items']), d, ctx.array(g['sf_alpha']), ctx.array(g['sf_mu']),
                         2.0, sub, ctx.array(z), out)
    np.testing.assert_allclose(out.to_host()[0], g['sf_out'], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize('d', [1, 3, 10, 11, 15, 16, 17, 22, 23, 32, 33, 46, 47, 64])
def test_gibbs_all_d(ctx, d):
    rs = np.random.RandomState(d)
    n, m, N = 9, 12, 150
    ii = np.concatenate([np.arange(n), rs.randint(0, n, N)])   # every user has >= 1 rating
    jj = rs.randint(0, m, ii.size)
    ratings = np.column_stack([ii, jj, rs.normal(size=ii.size)]).astype(float)
    items = rs.normal(size=(m, d))
    a = rs.normal(size=(d, d))
    alpha, mu, z = a @ a.T / d + np.eye(d), rs.normal(size=d), rs.normal(size=(n, d))
    rt = ctx.ratings(ratings, n, m)
    out = ctx.empty((n, d))
    ctx.gibbs_half_sweep(rt, 0, ctx.array(items), d, ctx.array(alpha), ctx.array(mu), 2.0, 0.1,
                         ctx.array(z), out)
    ptr, idx, rat = orc.csr_by_row(ratings, 0, n)
    ref = orc.gibbs_half_sweep(ptr, idx, rat, items, alpha, mu, 2.0, 0.1, z)
    np.testing.assert_allclose(out.to_host(), ref, rtol=5e-10, atol=1e-11)


def test_gibbs_errors(ctx):
    ratings = np.array([[0, 0, 1.], [1, 0, 2.]])
    rt = ctx.ratings(ratings, 3, 1)       # user 2 has no rating: KeyError in bayes_pmf.py:287
    d = 2
    args = (ctx.array(np.ones((1, d))), d, ctx.array(np.eye(d)), ctx.array(np.zeros(d)), 2.0, 0.,
            ctx.array(np.zeros((3, d))), ctx.empty((3, d)))
    with pytest.raises(ValueError):
        ctx.gibbs_half_sweep(rt, 0, *args)
    rt2 = ctx.ratings(ratings, 2, 1)
    with pytest.raises(np.linalg.LinAlgError):
        ctx.gibbs_half_sweep(rt2, 0, ctx.array(np.ones((1, d))), d, ctx.array(-5 * np.eye(d)),
                             ctx.array(np.zeros(d)), 2.0, 0., ctx.array(np.zeros((2, d))),
                             ctx.empty((2, d)))


@pytest.mark.parametrize('name', BAYES_CASES)
def test_sample_stats_golden(ctx, name):
    g = load_golden(name)
    S, n, d = g['chain_U'].shape
    m = g['chain_V'].shape[1]
    from active_matrix_factorization_amd._lib import grid_ld
    ld = grid_ld(m)
    add = float(g['mean_rating']) if int(g['subtract_mean']) else 0.
    mean, var, prob = ctx.empty((n, ld)), ctx.empty((n, ld)), ctx.empty((n, ld))
    dU, dV = ctx.array(g['chain_U']), ctx.array(g['chain_V'])
    ctx.sample_stats(S, n, m, d, ld, dU, dV, add, 3.5, mean=mean, var=var, prob=prob)
    mu, v, p = (ctx.download_grid(x, m) for x in (mean, var, prob))
    np.testing.assert_allclose(mu, g['predict_full'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(v, g['pred_variance_full'], rtol=1e-10, atol=1e-11)
    which = tuple(g['which'])
    np.testing.assert_array_equal(p[which], g['prob_ge'])
    # selection as bayes_pmf.full_test does it: np.argmax over the unrated list (bayes_pmf.py:702-712)
    assert tuple(g['which'][:, np.argmax(v[which])]) == tuple(g['pick_argmax'])
    ctx.sample_stats(S, n, m, d, ld, dU, dV, add, 0.5, prob=prob)
    np.testing.assert_array_equal(ctx.download_grid(prob, m), g['prob_ge_full'])


@pytest.mark.parametrize('shape', [(1, 1, 1, 1), (3, 5, 70, 2), (8, 33, 65, 5), (16, 47, 130, 15), (5, 32, 64, 16),
                                   (4, 70, 31, 17), (3, 20, 40, 33), (6, 95, 129, 64), (4, 40, 70, 12), (3, 50, 66, 24),
                                   (2, 33, 65, 49)])
def test_sample_stats_shapes_vs_oracle(ctx, shape):
    """K4 over every staging bucket (d = 1 .. 64, power-of-two d included: rotated LDS rows),
    ragged tiles and S = 1, against the oracle's sample-by-sample restatement."""
    S, n, m, d = shape
    rs = np.random.RandomState(S + n + m + d)
    Us, Vs = rs.normal(size=(S, n, d)), rs.normal(size=(S, m, d))
    from active_matrix_factorization_amd._lib import grid_ld
    ld = grid_ld(m)
    mean, var, prob = ctx.empty((n, ld)), ctx.empty((n, ld)), ctx.empty((n, ld))
    ctx.sample_stats(S, n, m, d, ld, ctx.array(Us), ctx.array(Vs), 1.25, 1.5, mean=mean, var=var, prob=prob)
    om, ov, op = orc.sample_stats(Us, Vs, 1.25, 1.5)
    scale = np.abs(om) + np.sqrt(ov) + 1.
    assert np.all(np.abs(ctx.download_grid(mean, m) - om) <= 1e-12 * scale)
    assert np.all(np.abs(ctx.download_grid(var, m) - ov) <= 1e-9 * (ov + 1e-3 * scale ** 2))
    # exceedance counts: exact except where a prediction sits within rounding of the cutoff
    x = np.einsum('sik,sjk->sij', Us, Vs) + 1.25
    safe = np.all(np.abs(x - 1.5) > 1e-9, axis=0)
    assert np.array_equal(ctx.download_grid(prob, m)[safe], op[safe])
    # the variant compiled without the exceedance counts (pred_variance / predict alone): the same bits
    mean2, var2 = ctx.empty((n, ld)), ctx.empty((n, ld))
    ctx.sample_stats(S, n, m, d, ld, ctx.array(Us), ctx.array(Vs), 1.25, 1.5, mean=mean2, var=var2)
    assert np.array_equal(mean2.to_host(), mean.to_host()) and np.array_equal(var2.to_host(), var.to_host())
    # padding columns of the grids are written as zeros
    assert not mean.to_host()[:, m:].any() and not var.to_host()[:, m:].any() and not prob.to_host()[:, m:].any()


# ---------------------------------------------------------------------------- dense (rocSOLVER)
@pytest.mark.parametrize('N', [3, 17, 63, 64, 65, 129, 200, 519])
def test_dense_project_psd_and_inverse(ctx, N):
    rs = np.random.RandomState(N)
    a = rs.normal(size=(N, N))
    # indefinite, asymmetric input: the projection branch (mn_active_pmf.py:61-66)
    d = ctx.array(a)
    assert ctx.dense_project_psd(d, N, 1e-5) is True
    out = d.to_host()
    sym = (a + a.T) / 2
    w, v = np.linalg.eigh(sym)
    ref = (v * np.maximum(w, 1e-5)) @ v.T
    ref = (ref + ref.T) / 2
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-10 * np.abs(ref).max())
    assert np.array_equal(out, out.T)
    # already PD: only symmetrised
    spd = a @ a.T + N * np.eye(N)
    skew = spd + 1e-3 * (a - a.T)
    d2 = ctx.array(skew)
    assert ctx.dense_project_psd(d2, N, 1e-5) is False
    np.testing.assert_allclose(d2.to_host(), (skew + skew.T) / 2, rtol=1e-15)
    # inverse + log|det| (np.linalg.inv / slogdet)
    inv = ctx.empty((N, N))
    ld = ctx.dense_inv_logdet(ctx.array(spd), N, inv)
    np.testing.assert_allclose(inv.to_host(), np.linalg.inv(spd), rtol=1e-9, atol=1e-12)
    assert abs(ld - np.linalg.slogdet(spd)[1]) <= 1e-11 * abs(ld) + 1e-12
    gen = a + N * np.eye(N)                      # general (non-symmetric) matrix
    inv2 = ctx.empty((N, N))
    ld2 = ctx.dense_inv_logdet(ctx.array(gen), N, inv2)
    np.testing.assert_allclose(inv2.to_host(), np.linalg.inv(gen), rtol=1e-9, atol=1e-12)
    assert abs(ld2 - np.linalg.slogdet(gen)[1]) <= 1e-11 * abs(ld2) + 1e-12
    with pytest.raises(np.linalg.LinAlgError):
        ctx.dense_inv_logdet(ctx.array(np.zeros((N, N))), N, inv2)


# ---------------------------------------------------------------------------- K5 (variational fit)
@pytest.mark.parametrize('shape', [(7, 5, 1), (13, 9, 2), (30, 41, 5), (25, 60, 10), (40, 33, 16), (12, 10, 23)])
def test_kl_gradient_and_value_vs_numpy(ctx, shape):
    """K5 kernels vs an independent NumPy restatement of matrix_normal_exps_cy.pyx:395-485
    (tests/mn_grad_numpy.py, itself pinned to the reference's golden gradients elsewhere)."""
    import mn_grad_numpy as ref
    n, m, d = shape
    rs = np.random.RandomState(n * 100 + d)
    N = n + m
    pairs = rs.permutation(n * m)[:min(n * m, 4 * N)]
    ii, jj = np.concatenate([np.arange(n), pairs // m]), np.concatenate([rs.randint(0, m, n), pairs % m])
    keep = np.unique(ii * m + jj, return_index=True)[1]          # unique (i, j)
    ii, jj = ii[np.sort(keep)], jj[np.sort(keep)]
    ratings = np.column_stack([ii, jj, rs.normal(2, 1, ii.size)]).astype(float)
    mean = rs.normal(size=(N, d))
    A = rs.normal(size=(N, N)) / np.sqrt(N)
    S = A @ A.T + .3 * np.eye(N)
    Bm = rs.normal(size=(d, d)) / np.sqrt(d)
    O = Bm @ Bm.T + .2 * np.eye(d)
    sig, su, sv = 0.7, 4.0, 6.0
    rt = ctx.ratings(ratings, n, m)
    gm, gs, go = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
    ctx.mn_kl_gradient(rt, d, ctx.array(mean), ctx.array(S), ctx.array(O), sig, su, sv, gm, gs, go)
    rm, rsg, ro = ref._mnormal_grad(mean, S, O, ratings, n, d, sig, su, sv)
    np.testing.assert_allclose(gm.to_host(), rm, rtol=1e-10, atol=1e-10 * np.abs(rm).max())
    np.testing.assert_allclose(gs.to_host(), rsg, rtol=1e-10, atol=1e-10 * np.abs(rsg).max())
    np.testing.assert_allclose(go.to_host(), ro, rtol=1e-11, atol=1e-10 * np.abs(ro).max())
    # KL value against the oracle's per-rating E[(u.v)^2] and NumPy log-determinants
    kl = ctx.mn_kl_value(rt, d, ctx.array(mean), ctx.array(S), ctx.array(O), sig, su, sv)
    trO = np.trace(O)
    bit = 0.
    for i, j, r in ratings:
        i, j = int(i), int(j)
        e2 = orc.mn_exp_dotprod_sq(n, mean, S, O, i, j)
        bit += e2 - 2 * r * ((mean[i] * mean[n + j]).sum() + S[i, n + j] * trO) + r * r
    ref_kl = (-(np.linalg.slogdet(S)[1] * d + np.linalg.slogdet(O)[1] * N) / 2.
              + ((mean[:n] ** 2).sum() + np.trace(S[:n, :n]) * trO) / (2 * su)
              + ((mean[n:] ** 2).sum()) / (2 * su) + bit / (2 * sig))
    assert abs(kl - ref_kl) <= 1e-10 * abs(ref_kl)


# ---------------------------------------------------------------------------- rank-64 shape (configs[4] family)
def test_rank64_paths(ctx):
    """The d = 64 code paths (BASELINE.json configs[4] uses rank 64) at a size the oracle still
    checks in seconds: K1 gradient/LL, K2 f32 sweep with fused reductions, K3 half-sweep."""
    rs = np.random.RandomState(64)
    n, m, d, N = 1500, 1100, 64, 120000
    flat = rs.choice(n * m, N, replace=False)
    ii, jj = flat // m, flat % m
    ii[:n], jj[:m] = np.arange(n), np.arange(m)                  # every row / column rated
    ratings = np.column_stack([ii, jj, rs.normal(size=N)]).astype(float)
    U, V = rs.normal(size=(n, d)) / 4, rs.normal(size=(m, d)) / 4
    kw = dict(sig=1., sig_u=10., sig_v=10., subtract_mean=False, mean_rating=0.)
    gU, gV, ll, _ = _grad(ctx, ratings, U, V, kw)
    oU, oV = orc.pmf_gradient(ratings, U, V, **kw)
    np.testing.assert_allclose(gU, oU, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(gV, oV, rtol=1e-10, atol=1e-10)
    assert abs(ll - orc.pmf_log_likelihood(ratings, U, V, **kw)) <= 1e-12 * abs(ll)
    # K2, f32, two states, compact inputs
    from active_matrix_factorization_amd._lib import grid_ld
    ld, B = grid_ld(m), 2
    mean = rs.normal(size=(B, n + m, d)) / 4
    cross = np.zeros((B, n, ld))
    cross[:, :, :m] = rs.normal(0, .05, (B, n, m))
    diag = rs.rand(B, n + m) + .5
    om = np.array([(lambda x: x @ x.T / d + .1 * np.eye(d))(rs.normal(size=(d, d))) for _ in range(B)])
    var = ctx.empty((B, n, ld), np.float32)
    dsum, bv, bi = ctx.empty(B), ctx.empty(B, np.float32), ctx.empty(B, np.int64)
    ctx.mn_score_grid(np.float32, B, n, m, d, ld, *(ctx.array(x, np.float32) for x in (mean, cross, diag, om)),
                      var=var, dsum=dsum, want=1, best_val=bv, best_idx=bi)
    v = ctx.download_grid(var, m)
    for b in range(B):
        _, ov = orc.mn_pred_means_vars_compact(40, m, np.vstack([mean[b, :40], mean[b, n:]]),
                                               cross[b, :40, :m], np.concatenate([diag[b, :40], diag[b, n:]]),
                                               om[b])
        assert np.all(np.abs(v[b, :40] - ov) <= 2e-5 * np.abs(ov).max())
        k = int(bi.to_host()[b])
        assert v[b].ravel()[k // m * m + k % m] == v[b].max()
        assert abs(dsum.to_host()[b] - v[b].astype(np.float64).sum()) <= 1e-6 * dsum.to_host()[b]
    # K3 at d = 64
    rt = ctx.ratings(ratings, n, m)
    a = rs.normal(size=(d, d))
    alpha, mu, z = a @ a.T / d + np.eye(d), rs.normal(size=d), rs.normal(size=(n, d))
    out = ctx.empty((n, d))
    ctx.gibbs_half_sweep(rt, 0, ctx.array(V), d, ctx.array(alpha), ctx.array(mu), 2.0, 0., ctx.array(z), out)
    ptr, idx, rat = orc.csr_by_row(ratings, 0, n)
    ref = orc.gibbs_half_sweep(ptr[:51], idx, rat, V, alpha, mu, 2.0, 0., z[:50])
    np.testing.assert_allclose(out.to_host()[:50], ref, rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize('case', ['tail_rows_empty', 'short_tail_block', 'long_row_4_waves'])
def test_pmf_gradient_ll_edge_shapes(ctx, case):
    """K1's row kernels issue every load of a trip unconditionally, entries past the end of a row read at the row's
    LAST entry and masked afterwards: that address is inside the array because a wave only enters
    the loop when its share of the row is non-empty (span > 0, so end - 1 >= beg).  Shapes whose last block of 4
    row slots is mostly padding, whose trailing rows have no rating at all, and a row split over 4 waves of which
    the last ones get nothing: gradient and log-likelihood against the oracle, under the suite's memory poisoning."""
    rs = np.random.RandomState(len(case))
    if case == 'tail_rows_empty':
        n, m, d = 6, 9, 3                      # users 4, 5 and items 7, 8 have no rating
        ii, jj = rs.randint(0, 4, 25), rs.randint(0, 7, 25)
    elif case == 'short_tail_block':
        n, m, d = 5, 2, 10                     # 7 row slots: the second block of 4 holds 3 rows
        ii, jj = rs.randint(0, n, 9), rs.randint(0, m, 9)
    else:
        n, m, d = 2, 150, 4                    # 120 ratings over 2 users: 60 per row -> 4 waves per row (>= 48)
        ii, jj = np.repeat([0, 1], [117, 3]), np.concatenate([rs.permutation(m)[:117], [0, 1, 2]])
    keep = np.sort(np.unique(ii * m + jj, return_index=True)[1])
    ii, jj = ii[keep], jj[keep]
    ratings = np.column_stack([ii, jj, rs.normal(size=ii.size)]).astype(float)
    U, V = rs.normal(size=(n, d)), rs.normal(size=(m, d))
    rt = ctx.ratings(ratings, n, m)
    dU, dV, gU, gV = ctx.array(U), ctx.array(V), ctx.empty((n, d)), ctx.empty((m, d))
    ctx.pmf_gradient(rt, dU, dV, d, 1., 10., 10., False, 0., gU, gV)
    oU, oV = orc.pmf_gradient(ratings, U, V)
    np.testing.assert_allclose(gU.to_host(), oU, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(gV.to_host(), oV, rtol=1e-12, atol=1e-13)
    ll, _ = ctx.pmf_log_likelihood(rt, dU, dV, d, 1., 10., 10., False, 0.)
    assert abs(ll - orc.pmf_log_likelihood(ratings, U, V)) <= 1e-12 * abs(ll)
    tr, steps, lr, conv = ctx.pmf_fit_lls(rt, dU, dV, d, 1., 10., 10., False, 0., 1e-4, 1e-10, 1e-2, max_steps=5)
    oUf, oVf, otr = orc.pmf_fit_lls(ratings, U, V, max_steps=5)
    assert steps == otr.size
    np.testing.assert_allclose(tr, otr, rtol=1e-11)
    np.testing.assert_allclose(dU.to_host(), oUf, rtol=1e-10, atol=1e-12)
    rt.free()


def _batch_fit_case(ctx, ratings, n, m, d, hyps, subtract_mean=False, max_steps=40, seed=0):
    """H batched re-fits (apmf_pmf_batch_fit_lls) against H separate apmf_pmf_fit_lls calls on containers REBUILT
    with the hypothesis' rating appended: factors, log-likelihood and step counts must agree bit for bit."""
    rs = np.random.RandomState(seed)
    H = len(hyps)
    U0, V0 = rs.random_sample((n, d)), rs.random_sample((m, d))
    rt = ctx.ratings(ratings, n, m)
    dU = ctx.array(np.broadcast_to(U0, (H, n, d)).copy())
    dV = ctx.array(np.broadcast_to(V0, (H, m, d)).copy())
    ei, ej, ev = [h[0] for h in hyps], [h[1] for h in hyps], [float(h[2]) for h in hyps]
    means = [float(np.mean(np.append(ratings[:, 2], v))) for v in ev] if subtract_mean else None
    ll, steps, conv = ctx.pmf_batch_fit_lls(rt, dU, dV, d, H, ei, ej, ev, 1., 10., 10., subtract_mean, means, 1e-4,
                                            1e-10, 1e-2, max_steps=max_steps)
    bU, bV = dU.to_host(), dV.to_host()
    for h, (i, j, v) in enumerate(hyps):
        plus = np.vstack([ratings, [i, j, v]])
        rth = ctx.ratings(plus, n, m)
        aU, aV = ctx.array(U0), ctx.array(V0)
        mr = float(np.mean(plus[:, 2])) if subtract_mean else 0.
        _, st, lr, cv, ll1 = ctx.pmf_fit_lls(rth, aU, aV, d, 1., 10., 10., subtract_mean, mr, 1e-4, 1e-10, 1e-2,
                                             max_steps=max_steps, max_trace=0, return_ll=True)
        assert st == steps[h] and cv == conv[h], (h, st, steps[h])
        assert ll1 == ll[h], (h, ll1, ll[h])
        assert np.array_equal(aU.to_host(), bU[h]) and np.array_equal(aV.to_host(), bV[h]), h
        rth.free()
    rt.free()
    return steps


@pytest.mark.parametrize('d', [2, 5, 10, 20])
def test_pmf_batch_fit_lls_equals_rebuilt(ctx, d):
    """The MAP re-fits of a lookahead with refit_lookahead (mn_active_pmf.py:662-665) as one batch: hypotheses whose
    rating lands in an empty-ish row, in the heaviest row (it moves up in the descending-nnz row order the partial
    sums follow), in rows tied with others, twice the same entry with different values; with and without
    subtract_mean."""
    rs = np.random.RandomState(40 + d)
    n, m, N = 23, 31, 180
    flat = rs.choice(n * m, N, replace=False)
    ratings = np.column_stack([flat // m, flat % m, rs.randint(1, 6, N)]).astype(float)
    rated = set(zip(ratings[:, 0].astype(int), ratings[:, 1].astype(int)))
    cnt_u = np.bincount(ratings[:, 0].astype(int), minlength=n)
    cnt_v = np.bincount(ratings[:, 1].astype(int), minlength=m)
    free = [(i, j) for i in range(n) for j in range(m) if (i, j) not in rated]
    heavy_u, light_u = int(np.argmax(cnt_u)), int(np.argmin(cnt_u))
    heavy_v, light_v = int(np.argmax(cnt_v)), int(np.argmin(cnt_v))
    pick = lambda i=None, j=None: next(e for e in free if (i is None or e[0] == i) and (j is None or e[1] == j))
    cand = [pick(heavy_u), pick(light_u), pick(j=heavy_v), pick(j=light_v), free[0], free[-1], free[len(free) // 2]]
    hyps = [(i, j, float(1 + (k % 5))) for k, (i, j) in enumerate(cand)] + [(cand[0][0], cand[0][1], 5.), (cand[0][0], cand[0][1], 1.)]
    steps = _batch_fit_case(ctx, ratings, n, m, d, hyps, max_steps=60)
    assert steps.min() > 3
    _batch_fit_case(ctx, ratings, n, m, d, hyps[:4], subtract_mean=True, max_steps=25, seed=1)


def test_pmf_batch_fit_lls_rows_without_ratings(ctx):
    """Hypotheses about a user and an item that have NO rating yet (the extra rating is the row's only entry: the row
    moves from the tail of the descending-nnz order the partial sums follow to the one-rating rows), about an entry
    whose user AND item are both empty, and H = 1; against fits on rebuilt containers, bit for bit."""
    rs = np.random.RandomState(77)
    n, m, d, N = 19, 26, 4, 110
    flat = rs.choice((n - 2) * (m - 3), N, replace=False)          # users n-2, n-1 and items m-3 .. m-1 stay empty
    ratings = np.column_stack([flat // (m - 3), flat % (m - 3), rs.randint(1, 6, N)]).astype(float)
    hyps = [(n - 1, 0, 4.), (0, m - 1, 2.), (n - 2, m - 2, 5.), (n - 1, m - 1, 1.), (3, 4, 3.) if (3, 4) not in
            set(zip(ratings[:, 0].astype(int), ratings[:, 1].astype(int))) else (n - 2, 1, 3.)]
    steps = _batch_fit_case(ctx, ratings, n, m, d, hyps, max_steps=50)
    assert steps.min() > 3
    _batch_fit_case(ctx, ratings, n, m, d, hyps[2:3], max_steps=30, seed=3)                 # H = 1
    _batch_fit_case(ctx, ratings, n, m, d, hyps[:2], subtract_mean=True, max_steps=20, seed=4)


def test_pmf_batch_fit_lls_headline_shape(ctx):
    """The batched MAP re-fit at the metric's own shape - 943 x 1682, rank 10, N = 100 000 (106 ratings per user: the
    4-waves-per-row kernels, some rows beyond 192 ratings) - six hypotheses against six fits on rebuilt containers,
    bit for bit, and against the reference's recorded log-likelihood trace of this very fit (fixture
    pmf_c2_fit_943x1682_d10) for the hypothesis-free case."""
    from conftest import c2_inputs
    ratings, U0, V0 = c2_inputs()
    n, m, d = 943, 1682, 10
    rated = set(zip(ratings[:, 0].astype(int).tolist(), ratings[:, 1].astype(int).tolist()))
    rs = np.random.RandomState(3)
    hyps = []
    while len(hyps) < 6:
        i, j = int(rs.randint(n)), int(rs.randint(m))
        if (i, j) not in rated:
            hyps.append((i, j, float(rs.randint(1, 6))))
    H = len(hyps)
    rt = ctx.ratings(ratings, n, m)
    dU = ctx.array(np.broadcast_to(U0, (H, n, d)).copy())
    dV = ctx.array(np.broadcast_to(V0, (H, m, d)).copy())
    ll, steps, conv = ctx.pmf_batch_fit_lls(rt, dU, dV, d, H, [h[0] for h in hyps], [h[1] for h in hyps],
                                            [h[2] for h in hyps], 1., 10., 10., False, None, 1e-4, 1e-10, 1e-2, max_steps=8)
    bU, bV = dU.to_host(), dV.to_host()
    assert (steps == 8).all()
    for h, (i, j, v) in enumerate(hyps[:3]):
        rth = ctx.ratings(np.vstack([ratings, [i, j, v]]), n, m)
        aU, aV = ctx.array(U0), ctx.array(V0)
        _, st, lr, cv, ll1 = ctx.pmf_fit_lls(rth, aU, aV, d, 1., 10., 10., False, 0., 1e-4, 1e-10, 1e-2, max_steps=8,
                                             max_trace=0, return_ll=True)
        assert st == 8 and ll1 == ll[h]
        assert np.array_equal(aU.to_host(), bU[h]) and np.array_equal(aV.to_host(), bV[h])
        rth.free()
    rt.free()


def test_pmf_batch_fit_lls_block_shapes(ctx):
    """A hypothesis can change the block shape its re-fit runs with - a row growing past 192 ratings (4 waves per
    row in k_fit_iter), N + 1 reaching 48 ratings per row (4 waves per row in the gradient / log-likelihood
    kernels) - and a batch may hold both kinds: each hypothesis must get what its rebuilt container would."""
    rs = np.random.RandomState(9)
    n, m, d = 5, 260, 4
    cols = rs.choice(m, 192, replace=False)
    rows = [np.column_stack([np.zeros(192), cols, rs.randint(1, 6, 192)])]
    for i in range(1, n):
        c = rs.choice(m, 9, replace=False)
        rows.append(np.column_stack([np.full(9, i), c, rs.randint(1, 6, 9)]))
    ratings = np.vstack(rows).astype(float)
    free0 = [j for j in range(m) if j not in set(cols)]
    rated1 = set(ratings[ratings[:, 0] == 1][:, 1].astype(int))
    free1 = [j for j in range(m) if j not in rated1]
    hyps = [(0, free0[0], 3.), (1, free1[0], 2.), (0, free0[1], 5.), (2, 7 if (2, 7) not in set(map(tuple, ratings[:, :2].astype(int))) else 8, 1.)]
    _batch_fit_case(ctx, ratings, n, m, d, hyps, max_steps=12)
    # N = 191 over min(n, m) = 4 rows: the rebuilt containers have 192 / 4 = 48 ratings per row
    n, m, N = 4, 70, 191
    flat = rs.choice(n * m, N, replace=False)
    ratings = np.column_stack([flat // m, flat % m, rs.randint(1, 6, N)]).astype(float)
    rated = set(zip(ratings[:, 0].astype(int), ratings[:, 1].astype(int)))
    free = [(i, j) for i in range(n) for j in range(m) if (i, j) not in rated]
    _batch_fit_case(ctx, ratings, n, m, 3, [(i, j, 4.) for i, j in free[:3]], max_steps=12)


def test_config5_shard_scale(ctx):
    """BASELINE.json configs[4] (100 000 x 20 000, rank 64, 5e7 observed) at the size ONE of its
    8 row shards has: 12 500 user rows x 20 000 items, 6.25e6 ratings.  Size-independent
    properties only (tools/c5_scale.py: sampled gradient rows and the log-likelihood against
    NumPy, sampled grid rows against the float64 closed form, fused sum / arg-max against the
    written grid); the full single-GPU run of the whole configuration is the tool itself."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'c5_scale.py')
    spec = importlib.util.spec_from_file_location('c5_scale', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out, ok = mod.run(n=12500, m=20000, d=64, N=6250000, ctx=ctx)
    assert ok, out


def test_config5_full_scale(ctx):
    """BASELINE.json configs[4] at its FULL size on one MI355X - 100 000 x 20 000, rank 64, 5e7 observed ratings,
    2e9 candidates: the whole 8-GPU configuration fits a single 288 GB device (ratings CSR x 2 1.2 GB, one f32
    cross block 8 GB in + 8 GB out).  Same size-independent properties as the shard-sized case above
    (tools/c5_scale.py): sampled gradient rows and the log-likelihood against NumPy, sampled grid rows against the
    float64 closed form, fused sum / arg-max against the written grid, an increasing fit_lls trace."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'c5_scale.py')
    spec = importlib.util.spec_from_file_location('c5_scale', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out, ok = mod.run(n=100000, m=20000, d=64, N=50000000, ctx=ctx)
    assert ok, out
    assert out['k2_candidates_per_s'] > 1e11 and out['k1_grad_steps_per_s'] > 10, out      # seconds, not minutes


@pytest.mark.parametrize('shape', [(4, 3, 1, 5), (6, 5, 3, 12), (9, 14, 5, 40), (7, 6, 8, 20), (5, 4, 17, 9)])
def test_full_kl_gradient_vs_numpy(ctx, shape):
    """K5-full (apmf_full_kl_gradient) against the NumPy restatement of normal_exps_cy.pyx:217-303
    (tests/full_grad_numpy.py), rows without ratings included."""
    from full_grad_numpy import normal_grad_numpy
    n, m, d, N = shape
    rs = np.random.RandomState(n * 100 + m * 10 + d)
    flat = rs.choice(n * m, N, replace=False)
    ratings = np.c_[flat // m, flat % m, rs.randint(1, 6, N)].astype(float)
    K = (n + m) * d
    mean = rs.normal(size=K)
    a = rs.normal(size=(K, K)) / np.sqrt(K)
    cov = a @ a.T + .3 * np.eye(K)
    u = np.arange(n * d).reshape(n, d).T
    v = n * d + np.arange(m * d).reshape(m, d).T
    rt = ctx.ratings(ratings, n, m)
    gm, gc = ctx.empty(K), ctx.empty((K, K))
    ctx.full_kl_gradient(rt, d, ctx.array(mean), ctx.array(cov), 0.7, 9., 11., gm, gc)
    rm, rc = normal_grad_numpy(mean, cov, ratings, d, u, v, 0.7, 9., 11.)
    np.testing.assert_allclose(gm.to_host(), rm, rtol=1e-10, atol=1e-10 * np.abs(rm).max())
    np.testing.assert_allclose(gc.to_host(), rc, rtol=1e-10, atol=1e-10 * np.abs(rc).max())


def test_fill_normal_device_generator(ctx):
    """Philox4x32-10 + Box-Muller: reproducible for (seed, offset), different across offsets, and
    standard normal to sampling accuracy (moments, tail mass, no correlation between neighbours)."""
    n = 1 << 20
    a, b, c = ctx.empty(n), ctx.empty(n), ctx.empty(n + 1)
    ctx.fill_normal(a, 12345, 0)
    ctx.fill_normal(b, 12345, 0)
    ctx.fill_normal(c, 12345, n // 2)
    x, y, z = a.to_host(), b.to_host(), c.to_host()
    assert np.array_equal(x, y) and np.all(np.isfinite(x)) and np.all(np.isfinite(z))
    assert not np.array_equal(x[:1000], z[:1000])
    assert abs(x.mean()) < 5 / np.sqrt(n) and abs(x.var() - 1) < 5 * np.sqrt(2. / n)
    assert abs((x ** 4).mean() - 3) < 0.05 and abs((np.abs(x) > 3).mean() - 0.0027) < 3e-4
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 5 / np.sqrt(n)
    assert abs(np.corrcoef(x[0::2], x[1::2])[0, 1]) < 5 / np.sqrt(n / 2)     # the two outputs of a pair
    odd = ctx.empty(7)
    ctx.fill_normal(odd, 1, 0)
    assert np.all(np.isfinite(odd.to_host()))


# ---------------------------------------------------------------------------- K7 (Gibbs hyper-parameters)
@pytest.mark.parametrize('shape', [(40, 3), (943, 15), (300, 32)])
def test_device_hyperparam_step_vs_numpy(ctx, shape):
    """apmf_gibbs_sample_hyperparam with SUPPLIED variates == the host arithmetic of
    BayesianPMF.sample_hyperparam + the Bartlett branch of sample_wishart (bayes_pmf.py:41-59,
    157-186) on the same chi-square / normal draws - including the reference's quirk of adding the
    scalar <mu0 - x_bar, mu0 - x_bar> to every element."""
    N, d = shape
    rs = np.random.RandomState(N + d)
    F = rs.normal(0.3, 1.5, (N, d)) @ (np.eye(d) + .2 * rs.normal(size=(d, d)))
    wi, b0, df, mu0 = np.eye(d) + .1 * np.ones((d, d)), 2., float(d), rs.normal(size=d) * .1
    dof = df + N
    chi = rs.chisquare(dof - np.arange(d))
    low = rs.normal(size=d * (d - 1) // 2)
    z = rs.normal(size=d)
    # host restatement
    x_bar = F.mean(0)
    S_bar = np.cov(F, rowvar=0).reshape(d, d)
    diff = mu0 - x_bar
    WI = np.linalg.inv(np.linalg.inv(wi) + N * S_bar + (b0 * N) / (b0 + N) * np.dot(diff, diff.T))
    WI /= 2
    WI = WI + WI.T
    A = np.diag(np.sqrt(chi))
    A[np.tri(d, k=-1, dtype=bool)] = low
    X = np.linalg.cholesky(WI) @ A
    alpha = X @ X.T
    lam = np.linalg.cholesky(np.linalg.inv((b0 + N) * alpha))
    mu = lam @ z + (b0 * mu0 + N * x_bar) / (b0 + N)
    d_mu, d_alpha = ctx.empty(d), ctx.empty((d, d))
    ctx.gibbs_sample_hyperparam(ctx.array(F), N, d, ctx.array(wi), b0, df, ctx.array(mu0), d_mu, d_alpha,
                                variates=ctx.array(np.concatenate([chi, low, z])))
    np.testing.assert_allclose(d_alpha.to_host(), alpha, rtol=1e-9, atol=1e-12 * np.abs(alpha).max())
    np.testing.assert_allclose(d_mu.to_host(), mu, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize('shape', [(10, 2, 2.), (40, 5, 5.), (300, 16, 16.), (90, 15, 15.)])
def test_device_hyperparam_step_batched(ctx, shape):
    """apmf_gibbs_sample_hyperparam_batched: H chains, one block each, supplied variates - against the package's own
    host arithmetic (BayesianPMF._hyper_from_variates = bayes_pmf.py:157-186 + sample_wishart's two branches,
    :41-59): the small-dof DIRECT scheme (dof = df + N <= 81 + d: what the 10 x 10 configs take) and Bartlett's."""
    from active_matrix_factorization_amd import bayes_pmf as bp
    N, d, df = shape
    H = 5
    rs = np.random.RandomState(int(N + d))
    np.random.seed(N * 7 + d)
    wi, b0, mu0 = np.eye(d) + .1 * np.ones((d, d)), 2., rs.normal(size=d) * .1
    b = bp.BayesianPMF.__new__(bp.BayesianPMF)
    b.latent_d = d
    b.u_hyperparams = (wi, b0, df, mu0)
    F = rs.normal(0.3, 1.5, (H, N, d))
    draws = [b._hyper_variates(N, True) for _ in range(H)]
    packed = [bp._pack_variates(v) for v in draws]
    direct = packed[0][1]
    assert (direct > 0) == (df + N <= 81 + d)
    d_mu, d_al, status = ctx.empty((H, d)), ctx.empty((H, d, d)), ctx.zeros(1, np.int32)
    ctx.gibbs_sample_hyperparam_batched(H, ctx.array(F), N, d, ctx.array(wi), b0, df, ctx.array(mu0),
                                        ctx.array(np.stack([p[0] for p in packed])), direct, d_mu, d_al, status)
    assert int(status.to_host()[0]) == 0
    mu, al = d_mu.to_host(), d_al.to_host()
    for h in range(H):
        rm, ra = b._hyper_from_variates(F[h], True, draws[h])
        np.testing.assert_allclose(al[h], ra, rtol=2e-11, atol=1e-12 * np.abs(ra).max())
        np.testing.assert_allclose(mu[h], rm, rtol=1e-13, atol=1e-14)


def test_device_hyperparam_generator_moments(ctx):
    """The same step drawing its own variates (Philox: Box-Muller normals, Marsaglia-Tsang
    chi-squares): E[alpha] = dof * WI for a Wishart draw, mu centred on the posterior mean;
    reproducible per (seed, offset), different across offsets."""
    N, d = 200, 4
    rs = np.random.RandomState(3)
    F = rs.normal(0.5, 1., (N, d))
    wi, b0, df, mu0 = np.eye(d), 2., float(d), np.zeros(d)
    x_bar = F.mean(0)
    S_bar = np.cov(F, rowvar=0)
    diff = mu0 - x_bar
    WI = np.linalg.inv(np.linalg.inv(wi) + N * S_bar + (b0 * N) / (b0 + N) * np.dot(diff, diff))
    WI = WI / 2 + (WI / 2).T
    dF, dW, dm0 = ctx.array(F), ctx.array(wi), ctx.array(mu0)
    d_mu, d_alpha = ctx.empty(d), ctx.empty((d, d))
    alphas, mus = [], []
    for k in range(600):
        ctx.gibbs_sample_hyperparam(dF, N, d, dW, b0, df, dm0, d_mu, d_alpha, seed=1234, offset=1024 * k)
        alphas.append(d_alpha.to_host())
        mus.append(d_mu.to_host())
    alphas, mus = np.array(alphas), np.array(mus)
    mean_alpha = alphas.mean(0)
    expect = (df + N) * WI
    # Wishart: Var[alpha_ij] = dof (WI_ij^2 + WI_ii WI_jj); 600 draws -> ~4 sigma band
    sd = np.sqrt((df + N) * (WI ** 2 + np.outer(np.diag(WI), np.diag(WI))) / 600)
    assert np.all(np.abs(mean_alpha - expect) < 4.5 * sd), (mean_alpha, expect)
    mu_mean = (b0 * mu0 + N * x_bar) / (b0 + N)
    assert np.all(np.abs(mus.mean(0) - mu_mean) < 5 * mus.std(0) / np.sqrt(600))
    assert np.all(np.array([np.linalg.eigvalsh(a).min() for a in alphas[:20]]) > 0)
    ctx.gibbs_sample_hyperparam(dF, N, d, dW, b0, df, dm0, d_mu, d_alpha, seed=1234, offset=1024 * 7)
    assert np.array_equal(d_alpha.to_host(), alphas[7]) and not np.array_equal(alphas[7], alphas[8])
