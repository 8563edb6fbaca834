"""The batched variational-fit engine (csrc/k6_batch.hip, apmf_mn_batch_*) through the C-ABI:
its KL / gradient against the reference's recorded values and an independent NumPy
restatement, and - the property the lookahead relies on - a batch of H hypotheses giving
bit-for-bit what each hypothesis gives when fitted alone on a ratings container that was
rebuilt with its extra rating (deepcopy -> add_rating -> fit_normal, mn_active_pmf.py:660-668).
Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from active_matrix_factorization_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _problem(n, m, d, seed, per_row=3):
    rs = np.random.RandomState(seed)
    N = n + m
    pairs = rs.permutation(n * m)[:min(n * m // 2, per_row * N)]
    ii = np.concatenate([np.arange(n), pairs // m])
    jj = np.concatenate([rs.randint(0, m, n), pairs % m])
    # every item rated at least once too
    ii = np.concatenate([ii, rs.randint(0, n, m)])
    jj = np.concatenate([jj, np.arange(m)])
    keep = np.sort(np.unique(ii * m + jj, return_index=True)[1])
    ii, jj = ii[keep], jj[keep]
    ratings = np.column_stack([ii, jj, rs.randint(1, 6, ii.size)]).astype(float)
    mean = rs.normal(size=(N, d))
    A = rs.normal(size=(N, N)) / np.sqrt(N)
    S = A @ A.T + .3 * np.eye(N)
    B = rs.normal(size=(d, d)) / np.sqrt(d)
    O = B @ B.T + .2 * np.eye(d)
    return rs, ratings, mean, S, O


def _unrated(rs, ratings, n, m, count):
    rated = set(zip(ratings[:, 0].astype(int), ratings[:, 1].astype(int)))
    out = []
    while len(out) < count:
        i, j = int(rs.randint(n)), int(rs.randint(m))
        if (i, j) not in rated and (i, j) not in out:
            out.append((i, j))
    return out


@pytest.mark.parametrize('shape', [(7, 5, 1), (13, 9, 2), (30, 41, 5), (40, 33, 16), (70, 90, 5), (100, 131, 10)])
def test_batch_kl_and_gradient_vs_numpy(ctx, shape):
    """KL value and gradient of the batch engine at H = 3 (same state, three different extra
    ratings) vs tests/mn_grad_numpy.py on the ratings with the extra one appended; sizes on both
    sides of one 64-column Cholesky panel (N = 12 ... 231: 1 to 4 panels)."""
    import mn_grad_numpy as ref
    n, m, d = shape
    N = n + m
    rs, ratings, mean, S, O = _problem(n, m, d, 1000 * n + d)
    sig, su, sv = 0.7, 4.0, 6.0
    extras = _unrated(rs, ratings, n, m, 2)
    vals = [3.0, 1.0]
    rt = ctx.ratings(ratings, n, m)
    b = ctx.mn_batch(rt, d, 3)
    b.set_state(ctx.array(mean), ctx.array(S), ctx.array(O))
    b.set_extra([extras[0][0], extras[1][0], -1], [extras[0][1], extras[1][1], -1], vals + [0.])
    kls = b.kl(sig, su, sv)
    trO = np.trace(O)
    for h in range(3):
        rr = ratings if h == 2 else np.vstack([ratings, [extras[h][0], extras[h][1], vals[h]]])
        bit = 0.
        for i, j, r in rr:
            i, j = int(i), int(j)
            e2 = orc.mn_exp_dotprod_sq(n, mean, S, O, i, j)
            bit += e2 - 2 * r * ((mean[i] * mean[n + j]).sum() + S[i, n + j] * trO) + r * r
        ref_kl = (-(np.linalg.slogdet(S)[1] * d + np.linalg.slogdet(O)[1] * N) / 2.
                  + ((mean[:n] ** 2).sum() + np.trace(S[:n, :n]) * trO) / (2 * su)
                  + ((mean[n:] ** 2).sum()) / (2 * su) + bit / (2 * sig))
        assert abs(kls[h] - ref_kl) <= 1e-10 * abs(ref_kl), (h, kls[h], ref_kl)
        gm, gs, go = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
        b.gradient(h, sig, su, sv, gm, gs, go)
        rm, rsg, ro = ref._mnormal_grad(mean, S, O, rr, n, d, sig, su, sv)
        np.testing.assert_allclose(gm.to_host(), rm, rtol=1e-10, atol=1e-10 * np.abs(rm).max())
        np.testing.assert_allclose(gs.to_host(), rsg, rtol=1e-9, atol=1e-10 * np.abs(rsg).max())
        np.testing.assert_allclose(go.to_host(), ro, rtol=1e-11, atol=1e-10 * np.abs(ro).max())
    # entropy = (d log|S| + N log|O|) / 2  (mn_active_pmf.py:513-521)
    ent = b.entropy()
    ref_ent = 0.5 * (d * np.linalg.slogdet(S)[1] + N * np.linalg.slogdet(O)[1])
    np.testing.assert_allclose(ent, ref_ent, rtol=1e-11)
    # total variance = sum of the oracle's variance grid
    _, pv = orc.mn_pred_means_vars(n, m, mean, S, O)
    np.testing.assert_allclose(b.total_variance(), pv.sum(), rtol=1e-11)
    b.free()


@pytest.mark.parametrize('shape', [(20, 13, 3), (40, 23, 2), (17, 15, 4), (50, 15, 5), (2, 31, 2)])
def test_batch_trial_point_edge_shapes(ctx, shape):
    """kb_trial_S forms the trial point in 32 x 32 tiles whose loads are all issued up front at addresses CLAMPED to
    the matrix (min(row, N - 1), min(col, N - 1): always inside the N x N array, masked afterwards - VERDICT r2 #9):
    matrix sizes one past / one short of / exactly on a tile edge (N = 33, 63, 32, 65, 33; the last with two users only).  One accepted step at a
    tiny learning rate must land on  sym(S - lr g_S), mean - lr g_mean, sym(Omega - lr g_Omega)  with the gradient the
    engine itself reports (checked against NumPy above), element by element - under the suite's memory poisoning a
    stray read would show as NaN."""
    n, m, d = shape
    N = n + m
    rs, ratings, mean, S, O = _problem(n, m, d, 77 * n + m)
    sig, su, sv = 1., 10., 10.
    rt = ctx.ratings(ratings, n, m)
    b = ctx.mn_batch(rt, d, 2)
    ex = _unrated(rs, ratings, n, m, 1)[0]
    b.set_state(ctx.array(mean), ctx.array(S), ctx.array(O))
    b.set_extra([ex[0], -1], [ex[1], -1], [2., 0.])
    lr0 = 1e-7
    grads = []
    for h in range(2):
        gm, gs, go = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
        b.gradient(h, sig, su, sv, gm, gs, go)
        grads.append((gm.to_host(), gs.to_host(), go.to_host()))
    lr, kl, nsteps, conv, _ = b.fit(sig, su, sv, 1e-5, lr0, max_steps=1)
    assert list(nsteps) == [1, 1] and np.allclose(lr, 1.25 * lr0)          # first trial accepted
    for h in range(2):
        dm, ds, do = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
        b.get_state(h, dm, ds, do)
        gm, gs, go = grads[h]
        cs = S - lr0 * gs
        co = O - lr0 * go
        assert np.isfinite(ds.to_host()).all()
        np.testing.assert_allclose(ds.to_host(), (cs + cs.T) / 2, rtol=1e-14, atol=1e-16)
        np.testing.assert_allclose(dm.to_host(), mean - lr0 * gm, rtol=1e-14, atol=1e-16)
        np.testing.assert_allclose(do.to_host(), (co + co.T) / 2, rtol=1e-14, atol=1e-16)
    b.free()


@pytest.mark.parametrize('name', ['mn_c1_10x10_d2_eye', 'mn_c1_10x10_d2_rand', 'mn_24x17_d5_rand'])
def test_batch_fit_golden_trace(ctx, name):
    """Three accepted fit_normal_kls steps of the reference (mn_active_pmf.py:242-288), run as
    hypothesis 1 of a batch of 3 whose other members carry extra ratings."""
    g = load_golden(name)
    ratings = g['ratings']
    n, m = g['users'].shape[0], g['items'].shape[0]
    d = int(g['d'])
    sig, su, sv = (float(x) for x in g['sigma'])
    rs = np.random.RandomState(5)
    ex = _unrated(rs, ratings, n, m, 2)
    steps = g['kl_trace'].size
    rt = ctx.ratings(ratings, n, m)
    b = ctx.mn_batch(rt, d, 3, trace_cap=steps)
    b.set_state(ctx.array(g['mean']), ctx.array(g['cov_useritems']), ctx.array(g['cov_latents']))
    b.set_extra([ex[0][0], -1, ex[1][0]], [ex[0][1], -1, ex[1][1]], [2., 0., 4.])
    lr, kl, nsteps, conv, trace = b.fit(sig, su, sv, 1e-5, 1e-4, max_steps=steps)
    assert nsteps[1] == steps
    np.testing.assert_allclose(trace[1], g['kl_trace'], rtol=1e-12)
    N = n + m
    dm, ds, do = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
    b.get_state(1, dm, ds, do)
    np.testing.assert_allclose(dm.to_host(), g['mean_fit'], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ds.to_host(), g['cov_useritems_fit'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(do.to_host(), g['cov_latents_fit'], rtol=1e-12, atol=1e-14)
    b.free()


def _fit_alone(ctx, ratings, n, m, d, mean, S, O, sig, su, sv, max_steps, cap):
    rt = ctx.ratings(ratings, n, m)
    dm, ds, do = ctx.array(mean), ctx.array(S), ctx.array(O)
    trace, steps, lr, conv, kl = ctx.mn_fit_normal(rt, d, dm, ds, do, sig, su, sv, 1e-5, 1e-4,
                                                   max_steps=max_steps, max_trace=cap)
    return dm.to_host(), ds.to_host(), do.to_host(), trace, steps, lr, conv, kl


@pytest.mark.parametrize('shape,max_steps,H', [((24, 17, 5), 25, 5), ((94, 425, 5), 6, 5), ((60, 75, 3), -1, 5),
                                                ((70, 90, 4), 12, 20), ((94, 425, 5), 4, 20)])
def test_batched_equals_alone_bitwise(ctx, shape, max_steps, H):
    """H hypotheses fitted in one batch == each fitted alone on a container rebuilt with its extra
    rating appended (what deepcopy + add_rating + fit_normal does): states, KL traces, step counts
    and learning rates bit-for-bit.  (94, 425, 5) is BASELINE.json configs[3]'s shape.  H = 20 puts 40
    matrices in a factorisation launch: the chain then factors diagonal blocks in launches of their own
    (csrc/chol.h) and updates the factor block column by block column (each tile takes all earlier panels in one
    go, from registers), while a fit run alone updates panel by panel and factors the diagonal blocks inside the
    update launch - same bits (at N = 519 a tile takes up to 8 panels)."""
    n, m, d = shape
    N = n + m
    rs, ratings, mean, S, O = _problem(n, m, d, 77 + n, per_row=2)
    # identity-like start, as initialize_approx() gives: the fit then runs many accepted steps
    S = np.eye(N) + 0.01 * S
    O = np.eye(d)
    sig, su, sv = 1.0, 10.0, 10.0
    ex = _unrated(rs, ratings, n, m, H - 1)
    vals = [[1., 5., 3., 2.][q % 4] for q in range(H - 1)]
    hi = [e[0] for e in ex] + [-1]
    hj = [e[1] for e in ex] + [-1]
    hv = vals + [0.]
    cap = 64
    rt = ctx.ratings(ratings, n, m)
    b = ctx.mn_batch(rt, d, H, trace_cap=cap)
    b.set_state(ctx.array(mean), ctx.array(S), ctx.array(O))
    b.set_extra(hi, hj, hv)
    lr, kl, nsteps, conv, trace = b.fit(sig, su, sv, 1e-5, 1e-4, max_steps=max_steps)
    tv = b.total_variance()
    dm, ds, do = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
    assert len(set(nsteps.tolist())) >= 1
    for h in range(H):
        rr = ratings if hi[h] < 0 else np.vstack([ratings, [hi[h], hj[h], hv[h]]])
        am, as_, ao, atrace, asteps, alr, aconv, akl = _fit_alone(ctx, rr, n, m, d, mean, S, O, sig, su, sv,
                                                                  max_steps, cap)
        b.get_state(h, dm, ds, do)
        assert asteps == nsteps[h] and aconv == conv[h]
        assert alr == lr[h] and akl == kl[h]
        k = min(cap, asteps)
        assert np.array_equal(atrace[:k], trace[h, :k])
        assert np.array_equal(dm.to_host(), am)
        assert np.array_equal(ds.to_host(), as_)
        assert np.array_equal(do.to_host(), ao)
        # and the sweep of the batch's resident state == the f64 sweep of the stand-alone result
        _, pv = orc.mn_pred_means_vars(n, m, am, as_, ao)
        assert abs(tv[h] - pv.sum()) <= 1e-11 * abs(pv.sum())
    b.free()


@pytest.mark.parametrize('shape', [(20, 25, 3), (70, 95, 3)])
@pytest.mark.parametrize('route', ['eigenpairs', 'full'])
def test_batch_slow_path_projection(ctx, route, shape, monkeypatch):
    """A trial point whose covariance really needs its spectrum clipped (the certificate fails):
    the hypothesis leaves the batch for the projection - through its eigenpairs below min_eig (the
    default) or through the full eigendecomposition - and comes back; the others are unaffected.
    Checked against a NumPy restatement of one search iteration (eigh + clip + rebuild)."""
    import mn_grad_numpy as ref
    if route == 'full':
        monkeypatch.setenv('APMF_BATCH_EIG', 'full')
    else:
        monkeypatch.delenv('APMF_BATCH_EIG', raising=False)
    n, m, d = shape              # N = 45: one 64-column panel; N = 165: three, the last one short
    N = n + m
    rs, ratings, mean, S, O = _problem(n, m, d, 4242)
    sig, su, sv = 1.0, 10.0, 10.0
    # hypothesis 0: a covariance with tiny eigenvalues, so that a gradient step makes it indefinite
    w, v = np.linalg.eigh(S)
    S_bad = (v * np.concatenate([np.full(5, 2e-5), w[5:]])) @ v.T
    S_bad = (S_bad + S_bad.T) / 2
    rt = ctx.ratings(ratings, n, m)
    b = ctx.mn_batch(rt, d, 2, trace_cap=4)
    b.set_state(ctx.array(np.stack([mean, mean])), ctx.array(np.stack([S_bad, S])), ctx.array(np.stack([O, O])),
                per_hypothesis=7)
    lr0 = 1e-4
    lr, kl, nsteps, conv, trace = b.fit(sig, su, sv, 1e-5, lr0, max_steps=1)

    def project(mat, min_eig=1e-5):
        mat = (mat + mat.T) / 2
        vals, vecs = np.linalg.eigh(mat)
        if vals.min() < min_eig:
            mat = (vecs * np.maximum(vals, min_eig)) @ vecs.T
            mat = (mat + mat.T) / 2
        return mat

    def kl_of(mean_, S_, O_):
        trO = np.trace(O_)
        bit = 0.
        for i, j, r in ratings:
            i, j = int(i), int(j)
            e2 = orc.mn_exp_dotprod_sq(n, mean_, S_, O_, i, j)
            bit += e2 - 2 * r * ((mean_[i] * mean_[n + j]).sum() + S_[i, n + j] * trO) + r * r
        return (-(np.linalg.slogdet(S_)[1] * d + np.linalg.slogdet(O_)[1] * N) / 2.
                + ((mean_[:n] ** 2).sum() + np.trace(S_[:n, :n]) * trO) / (2 * su)
                + ((mean_[n:] ** 2).sum()) / (2 * su) + bit / (2 * sig))

    clipped = False
    for h, S0 in enumerate((S_bad, S)):
        old = kl_of(mean, S0, O)
        gm, gs, go = ref._mnormal_grad(mean, S0, O, ratings, n, d, sig, su, sv)
        lr_h = lr0
        while True:
            raw = S0 - lr_h * gs
            if h == 0 and np.linalg.eigvalsh((raw + raw.T) / 2).min() < 1e-5:
                clipped = True
            cand = (mean - lr_h * gm, project(raw), project(O - lr_h * go))
            new = kl_of(*cand)
            if new < old:
                break
            lr_h *= .5
        assert nsteps[h] == 1
        assert abs(trace[h, 0] - new) <= 1e-9 * abs(new)
        dS = ctx.empty((N, N))
        b.get_state(h, None, dS, None)
        np.testing.assert_allclose(dS.to_host(), cand[1], rtol=1e-8, atol=1e-10)
    assert clipped, 'the test problem no longer exercises the projection branch'
    st = b.fit_stats()
    # (the first trial point of hypothesis 0 has more eigenvalues below min_eig than the eigenpair route takes
    # on and goes through the full decomposition either way; the halved step's five go through the eigenpairs)
    assert st['proj_low' if route == 'eigenpairs' else 'proj_full'] >= 1
    if route == 'full':
        assert st['proj_low'] == 0
    b.free()


@pytest.mark.parametrize('lanes', ['1', '3'])
def test_batch_slow_lanes_same_states(ctx, lanes, monkeypatch):
    """Hypotheses that keep leaving the fast path (covariances with eigenvalues at the clipping
    threshold) go through the slow lanes - streams, rocBLAS handles and worker threads of their own -
    while the others iterate: however many lanes there are, every hypothesis ends in the same state,
    bit for bit, as when it is fitted in a batch of one."""
    monkeypatch.setenv('APMF_BATCH_LANES', lanes)
    n, m, d, H = 30, 40, 3, 6
    N = n + m
    rs, ratings, mean, S, O = _problem(n, m, d, 991)
    sig, su, sv = 1.0, 10.0, 10.0
    w, v = np.linalg.eigh(S)
    states = []
    for h in range(H):
        k = 3 + h                       # k eigenvalues sit just above min_eig = 1e-5
        Sh = (v * np.concatenate([np.full(k, 1.2e-5 + 1e-6 * h), w[k:]])) @ v.T
        states.append((Sh + Sh.T) / 2)
    rt = ctx.ratings(ratings, n, m)

    def run(idx):
        b = ctx.mn_batch(rt, d, len(idx), trace_cap=32)
        b.set_state(ctx.array(np.stack([mean] * len(idx))), ctx.array(np.stack([states[h] for h in idx])),
                    ctx.array(np.stack([O] * len(idx))), per_hypothesis=7)
        lr, kl, nsteps, conv, trace = b.fit(sig, su, sv, 1e-5, 1e-3, max_steps=12)
        out = []
        dm, ds, do = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
        for q in range(len(idx)):
            b.get_state(q, dm, ds, do)
            out.append((dm.to_host(), ds.to_host(), do.to_host(), float(kl[q]), int(nsteps[q]), float(lr[q])))
        b.free()
        return out

    together = run(list(range(H)))
    for h in range(H):
        alone = run([h])[0]
        for x, y in zip(together[h], alone):
            assert np.array_equal(x, y)


def test_batch_projection_by_eigenpairs_equals_full_decomposition(ctx, monkeypatch):
    """project_psd (mn_active_pmf.py:42-67) rebuilds V max(L, min_eig) V' from a full eigh; the batched fit
    finds only the eigenpairs below min_eig (shift bracketed by one batched Cholesky, inverse iteration,
    deflation + certificate) and adds (min_eig - l) v v'.  Same trial matrices either way: a search that clips
    at many of its trial points must go through the same accept / reject decisions and end in the same states
    up to the rounding of the two routes, and the eigenpair route must have done the work (the ladder of
    shifts does not reach: full decomposition, which APMF_BATCH_EIG=full forces everywhere)."""
    n, m, d, H = 30, 40, 3, 5
    N = n + m
    rs, ratings, mean, S, O = _problem(n, m, d, 4242)
    sig, su, sv = 1.0, 10.0, 10.0
    w, v = np.linalg.eigh(S)
    states = []
    for h in range(H):
        # one to three eigenvalues near the threshold, the rest of the spectrum well above it
        k = 1 + h % 3
        Sh = (v * np.concatenate([np.full(k, 1.5e-5 + 2e-6 * h), w[k:]])) @ v.T
        states.append((Sh + Sh.T) / 2)
    rt = ctx.ratings(ratings, n, m)

    def run(mode):
        if mode:
            monkeypatch.setenv('APMF_BATCH_EIG', mode)
        else:
            monkeypatch.delenv('APMF_BATCH_EIG', raising=False)
        b = ctx.mn_batch(rt, d, H, trace_cap=64)
        b.set_state(ctx.array(np.stack([mean] * H)), ctx.array(np.stack(states)), ctx.array(np.stack([O] * H)),
                    per_hypothesis=7)
        lr, kl, nsteps, conv, trace = b.fit(sig, su, sv, 1e-5, 1e-3, max_steps=25)
        st = b.fit_stats()
        out = []
        dm, ds, do = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
        for q in range(H):
            b.get_state(q, dm, ds, do)
            out.append((dm.to_host(), ds.to_host(), do.to_host(), float(kl[q]), int(nsteps[q]), float(lr[q])))
        b.free()
        return out, st

    low, st_low = run(None)
    full, st_full = run('full')
    assert st_full['proj_low'] == 0 and st_full['proj_full'] > 0
    assert st_low['proj_low'] + st_low['proj_full'] == st_full['proj_full']          # the same trial points clipped
    # (the first trial points of this search are wildly indefinite - more eigenvalues below min_eig than the
    # eigenpair route takes on, LP_KMAX = 6 - so both routes are exercised; at the shape of configs[3] every
    # projection of a lookahead goes through the eigenpairs: profiles/r02_lookahead_*.json)
    assert st_low['proj_low'] > 0 and st_low['proj_full'] > 0
    assert st_low['trials'] == st_full['trials']
    for a, b_ in zip(low, full):
        assert a[4] == b_[4] and a[5] == b_[5]                  # accepted steps, learning rate
        np.testing.assert_allclose(a[3], b_[3], rtol=1e-10)     # KL
        for x, y in zip(a[:3], b_[:3]):
            np.testing.assert_allclose(x, y, rtol=0, atol=1e-10 * np.abs(y).max())
        assert np.linalg.eigvalsh(a[1]).min() >= 1e-5 * (1 - 1e-6)


@pytest.mark.parametrize('shape,H,steps', [((70, 90, 4), 20, 14), ((94, 425, 5), 24, 5), ((21, 44, 3), 18, 30)])
def test_batch_trial_point_routes_same_bits(ctx, monkeypatch, shape, H, steps):
    """The trial point of a search step is written ONCE (kb_trial_S) and, when a launch factors many matrices, both of
    its factorisations read it from there on their first touch of every tile (ChJob::src, csrc/chol.h) instead of
    from copies F / W; and once the current point and the gradient are exactly symmetric - after the first accepted
    step - the mirrored tiles are not read, (x + x) / 2 being x.  APMF_BATCH_FROM_CAND=0 / APMF_BATCH_SYM=0 switch the
    two off: states, KL traces, step counts and learning rates are the same bits in all four combinations (N = 160,
    519 and 65: tiles on, past and one short of a block edge)."""
    n, m, d = shape
    N = n + m
    rs, ratings, mean, S, O = _problem(n, m, d, 5 * n + m, per_row=2)
    S = np.eye(N) + 0.01 * S
    ex = _unrated(rs, ratings, n, m, H)
    rt = ctx.ratings(ratings, n, m)

    def run(cand, sym):
        monkeypatch.setenv('APMF_BATCH_FROM_CAND', cand)
        monkeypatch.setenv('APMF_BATCH_SYM', sym)
        b = ctx.mn_batch(rt, d, H, trace_cap=64)
        b.set_state(ctx.array(mean), ctx.array(S), ctx.array(np.eye(d)))
        b.set_extra([e[0] for e in ex], [e[1] for e in ex], [[1., 5., 3., 2.][q % 4] for q in range(H)])
        lr, kl, nsteps, conv, trace = b.fit(1., 10., 10., 1e-5, 1e-4, max_steps=steps)
        out = [lr.copy(), kl.copy(), nsteps.copy(), trace.copy()]
        dm, ds, do = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
        for q in (0, H // 2, H - 1):
            b.get_state(q, dm, ds, do)
            out += [dm.to_host(), ds.to_host(), do.to_host()]
        st = b.fit_stats()
        b.free()
        return out, st

    ref, st0 = run('1', '1')
    assert int(ref[2].min()) >= 2                       # accepted steps: the symmetric path was taken
    assert np.array_equal(ref[5], ref[5].T)             # the state it leaves is symmetric bit for bit
    for cand, sym in (('0', '1'), ('1', '0'), ('0', '0')):
        got, st = run(cand, sym)
        assert st['trials'] == st0['trials']
        for x, y in zip(ref, got):
            assert np.array_equal(x, y, equal_nan=True), (cand, sym)      # (trace slots beyond the steps taken: poison)


def test_batch_certificate_by_eigenvalue_bound_same_bits(ctx, monkeypatch):
    """A trial point needs project_psd's eigh only to learn that nothing is clipped; the batched fit certifies
    that by factoring cand - (min_eig + delta) I - or, when the step is short, by
    l_min(cand) >= 1 / |cur^-1|_F - |lr g|_F >= min_eig from the inverse the gradient has just computed, without
    that factorisation.  Either way the trial's arithmetic is the same: states, KL traces, steps and learning
    rates bit for bit (APMF_BATCH_CERT=factor switches the bound off)."""
    n, m, d, H = 40, 55, 3, 4
    N = n + m
    rs, ratings, mean, S, O = _problem(n, m, d, 321, per_row=2)
    ex = _unrated(rs, ratings, n, m, H)
    rt = ctx.ratings(ratings, n, m)

    def run(mode):
        if mode:
            monkeypatch.setenv('APMF_BATCH_CERT', mode)
        else:
            monkeypatch.delenv('APMF_BATCH_CERT', raising=False)
        b = ctx.mn_batch(rt, d, H, trace_cap=64)
        b.set_state(ctx.array(.1 * mean), ctx.array(np.eye(N)), ctx.array(np.eye(d)))
        b.set_extra([e[0] for e in ex], [e[1] for e in ex], [1., 5., 3., 2.])
        lr, kl, nsteps, conv, trace = b.fit(1., 10., 10., 1e-5, 1e-4, max_steps=60)
        st = b.fit_stats()
        out = []
        dm, ds, do = ctx.empty((N, d)), ctx.empty((N, N)), ctx.empty((d, d))
        for q in range(H):
            b.get_state(q, dm, ds, do)
            out.append((dm.to_host(), ds.to_host(), do.to_host(), trace[q, :int(nsteps[q])].copy(), int(nsteps[q]),
                        float(lr[q])))
        b.free()
        return out, st

    bound, st_b = run(None)
    fact, st_f = run('factor')
    assert st_f['cert_by_bound'] == 0
    assert 0 < st_b['cert_by_bound'] <= st_b['trials'] and st_b['trials'] == st_f['trials']
    for a, b_ in zip(bound, fact):
        for x, y in zip(a, b_):
            assert np.array_equal(x, y)

