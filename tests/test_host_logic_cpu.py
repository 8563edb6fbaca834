"""Host-side logic that needs no GPU: the quadrature predictor of the continuous lookahead and the packing of
the Gibbs hyper-parameter variates."""
import os
import warnings

import numpy as np
import pytest
from scipy import stats


def test_quadrature_first_pass_and_bisections_are_predicted():
    """The continuous lookahead (mn_active_pmf.py:686-691) integrates with stats.norm.expect = adaptive 21-point
    Gauss-Kronrod quadrature.  _Quadrature must know, before any integrand value exists, every point of the first
    pass (so that the pool's hypotheses can be fitted as one batch) and, when the routine bisects an interval, the 42
    points of its two halves: run the real quadrature on a noisy integrand (it never converges: 50 subdivisions per
    interval) and check that no point it asks for was unpredicted."""
    from active_matrix_factorization_amd._active_base import _Quadrature, _gk21_points
    for mean, std in ((3.0, 1.3), (-0.4, 0.25), (120., 40.)):
        left, right = mean - 2 * std, mean + 2 * std
        q = _Quadrature(mean, std, left, right)
        assert q.intervals is not None, 'the QUADPACK node formula does not reproduce the dry run'
        assert len(q.first_pass) == 21 * len(q.intervals)
        assert q.first_pass[:21] == _gk21_points(*q.intervals[0])
        assert abs(q.intervals[0][0] - left) < 1e-12 and abs(q.intervals[-1][1] - right) < 1e-12
        rs = np.random.RandomState(0)
        known = {v: 1. + rs.rand() for v in q.first_pass}
        stats_ = {'batches': 0, 'stray': 0}

        def integrand(v):
            v = float(v)
            if v not in known:
                more = q.bisection_points(v)
                if more and v in more:
                    stats_['batches'] += 1
                    for x in more:
                        known.setdefault(x, 1. + rs.rand())
                else:
                    stats_['stray'] += 1
                    known[v] = 1. + rs.rand()
            return known[v]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            val = stats.norm.expect(integrand, loc=mean, scale=std, lb=left, ub=right, epsrel=.02)
        assert np.isfinite(val) and stats_['stray'] == 0 and stats_['batches'] > 10, stats_
    # a smooth integrand converges on the first pass: nothing beyond it is asked for
    q = _Quadrature(0., 1., -2., 2.)
    seen = []
    stats.norm.expect(lambda v: seen.append(float(v)) or float(v) ** 2, loc=0., scale=1., lb=-2., ub=2., epsrel=.02)
    assert seen == q.first_pass


def test_pack_variates_layouts():
    """k7_hyper.hip's supplied-variates layouts: Bartlett [chi (d) | strict lower triangle row-major | z (d)] and the
    reference's small-dof direct scheme [G (d x dof) row-major | z (d)] (bayes_pmf.py:41-59)."""
    from active_matrix_factorization_amd import bayes_pmf as bp
    d = 4
    np.random.seed(1)
    w = bp._wishart_variates(d, 10)                 # dof <= 81 + d, integral: direct
    assert w[0] == 'direct' and w[1].shape == (d, 10)
    z = np.arange(d, dtype=float)
    flat, dof = bp._pack_variates((w, z))
    assert dof == 10 and flat.size == d * 10 + d and np.array_equal(flat[:d * 10].reshape(d, 10), w[1])
    assert np.array_equal(flat[-d:], z)
    w = bp._wishart_variates(d, 200.)               # Bartlett
    assert w[0] == 'bartlett'
    flat, dof = bp._pack_variates((w, z))
    assert dof == 0 and flat.size == d + d * (d - 1) // 2 + d
    A = np.diag(np.sqrt(w[1]))
    A[np.tri(d, k=-1, dtype=bool)] = w[2]
    low = flat[d:d + d * (d - 1) // 2]
    assert [A[1, 0], A[2, 0], A[2, 1], A[3, 0], A[3, 1], A[3, 2]] == list(low)
    # the two host branches agree with what they replace
    s = np.eye(d) + .1
    np.random.seed(2)
    a = bp.sample_wishart(s, 10)
    np.random.seed(2)
    b = bp._wishart_from_variates(s, bp._wishart_variates(d, 10))
    assert np.array_equal(a, b)


def test_lookahead_streams_knob(monkeypatch):
    """Stream groups of a batched lookahead: 3 by default, APMF_LOOKAHEAD_STREAMS overrides - and nothing else does:
    a run under a profiler executes the code an unprofiled run executes (tools/prof_*.sh pass
    APMF_LOOKAHEAD_STREAMS=1 to their kernel-trace passes explicitly)."""
    from active_matrix_factorization_amd import mn_active_pmf as mn
    for k in [k for k in os.environ if k.startswith('ROCPROF_')] + ['APMF_LOOKAHEAD_STREAMS']:
        monkeypatch.delenv(k, raising=False)
    assert mn.lookahead_streams() == 3
    monkeypatch.setenv('ROCPROF_KERNEL_TRACE', '1')
    assert mn.lookahead_streams() == 3
    monkeypatch.setenv('APMF_LOOKAHEAD_STREAMS', '2')
    assert mn.lookahead_streams() == 2
    monkeypatch.setenv('APMF_LOOKAHEAD_STREAMS', '1')
    assert mn.lookahead_streams() == 1


def test_structured_eigenvalue_bound_is_a_lower_bound():
    """The bound that lets a trial of the batched variational fit skip its shifted factorisation (cert_by_bound,
    csrc/k6_batch.hip) restated in NumPy and checked against exact eigenvalues: for
    cand = cur - lr g,  g = Rs - (d/2)(2 cur^-1 - diag cur^-1)  (matrix_normal_exps_cy.pyx:395-485: Rs = the rating / prior
    part, symmetric, on the diagonal and at the rated (user, item) pairs),
        l_min(cand) >= 2 sqrt(2c) - lr (max_t q_t + |cross entries of Rs|_F),   c = lr d / 2,  q_t = Rs_tt + (d/2)(cur^-1)_tt,
    and >= L + 2c / L - ... when every eigenvalue of cur is >= L >= sqrt(2c).  Random positive definite `cur` with
    eigenvalues down to 1e-2, random sparse Rs of either sign, learning rates over five decades."""
    rs = np.random.RandomState(3)
    checked = tight = 0
    for trial in range(60):
        N, d = int(rs.randint(8, 60)), int(rs.randint(1, 8))
        Q, _ = np.linalg.qr(rs.normal(size=(N, N)))
        lam = 10. ** rs.uniform(-2, .2, N)
        cur = (Q * lam) @ Q.T
        cur = (cur + cur.T) / 2
        inv = np.linalg.inv(cur)
        Rs = np.diag(rs.normal(0, 30, N) + 40)
        n_u = N // 3
        for _ in range(int(rs.randint(1, 3 * N))):
            i, j = int(rs.randint(0, n_u)), int(rs.randint(n_u, N))
            v = rs.normal(0, 10)
            Rs[i, j] += v
            Rs[j, i] += v
        g = Rs - d / 2. * (2 * inv - np.diag(np.diag(inv)))
        off = Rs - np.diag(np.diag(Rs))
        q = np.diag(Rs) + d / 2. * np.diag(inv)
        up_unit = max(q.max(), 0.) + np.sqrt((off[:n_u] ** 2).sum())      # (each cross entry once: |off|_2 <= that norm)
        L = 1. / np.linalg.norm(inv, 'fro')
        assert L <= np.linalg.eigvalsh(cur).min() * (1 + 1e-12)
        for lr in 10. ** rs.uniform(-7, -2, 6):
            c = lr * d / 2.
            low = 2 * np.sqrt(2 * c)
            if L >= np.sqrt(2 * c):
                low = max(low, L + 2 * c / L)
            bound = low - lr * up_unit
            cand = cur - lr * g
            true = np.linalg.eigvalsh((cand + cand.T) / 2).min()
            assert bound <= true + 1e-12 * max(1., abs(true)), (trial, lr, bound, true)
            checked += 1
            tight += bound > 0
    assert checked == 360 and tight > 60        # (and it is not vacuous: positive - certifying - in a good share of the cases)
