"""Shared body of the sharding tests: runs every sharded algorithm on this rank and checks it
against the unsharded CPU oracle.  Used with OracleEngine under gloo on CPU (world 2) and with
HipEngine on the GPU box."""
import numpy as np

from oracle import oracle as orc
from active_matrix_factorization_amd import sharding as sh


def problem(seed=0, n=23, m=31, d=5, N=260):
    rs = np.random.RandomState(seed)
    ii = np.concatenate([np.arange(n), rs.randint(0, n, N)])
    jj = np.concatenate([rs.randint(0, m, n), rs.randint(0, m, N)])
    jj[:m] = np.arange(m)                       # every item rated at least once
    ratings = np.column_stack([ii, jj, rs.normal(size=ii.size)]).astype(float)
    U, V = rs.normal(size=(n, d)), rs.normal(size=(m, d))
    return rs, ratings, U, V, n, m, d


def check_all(make_engine, comm, tol=1e-10, dims=None):
    rs, ratings, U, V, n, m, d = problem(**(dims or {}))
    eng = make_engine(ratings, n, m, d)
    # --- K1: sharded fit == single-process fit
    U1, V1, tr1 = orc.pmf_fit_lls(ratings, U, V, max_steps=12)
    U2, V2, tr2 = sh.sharded_fit_lls(eng, comm, U, V, n, m, max_steps=12)
    assert tr2.size == tr1.size
    np.testing.assert_allclose(tr2, tr1, rtol=1e-11)
    np.testing.assert_allclose(U2, U1, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(V2, V1, rtol=1e-9, atol=1e-11)
    ll = sh.sharded_log_likelihood(eng, comm, U, V, n, m, 1., 10., 10.)
    assert abs(ll - orc.pmf_log_likelihood(ratings, U, V)) <= 1e-12 * abs(ll)
    # --- K3: sharded sweep == row-by-row sweep
    a = rs.normal(size=(d, d))
    alpha_u, alpha_v = a @ a.T / d + np.eye(d), np.eye(d) * 1.5
    mu_u, mu_v = rs.normal(size=d), rs.normal(size=d)
    zu, zv = rs.normal(size=(n, d)), rs.normal(size=(m, d))
    nu, nv = sh.sharded_gibbs_sweep(eng, comm, U, V, n, m, alpha_u, mu_u, alpha_v, mu_v, 2.0, 0.2, zu, zv)
    pu, iu, ru = orc.csr_by_row(ratings, 0, n)
    pv, iv, rv = orc.csr_by_row(ratings, 1, m)
    ref_u = orc.gibbs_half_sweep(pu, iu, ru, V, alpha_u, mu_u, 2.0, 0.2, zu)
    ref_v = orc.gibbs_half_sweep(pv, iv, rv, ref_u, alpha_v, mu_v, 2.0, 0.2, zv)
    np.testing.assert_allclose(nu, ref_u, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(nv, ref_v, rtol=1e-8, atol=1e-10)
    # --- K2: sharded selection == global arg-extremum, ties -> lowest flat index
    mean = rs.normal(size=(n + m, d))
    A = rs.normal(size=(n + m, n + m)) / np.sqrt(n + m)
    R = A @ A.T + .1 * np.eye(n + m)
    b = rs.normal(size=(d, d))
    Om = b @ b.T / d + .1 * np.eye(d)
    mask = rs.rand(n, m) < .8
    _, pvar = orc.mn_pred_means_vars(n, m, mean, R, Om)
    for want in (1, -1):
        tot, bv, ij = sh.sharded_select(eng, comm, n, m, mean, R[:n, n:], np.diag(R).copy(), Om, mask, want)
        cand = np.where(mask, pvar, -np.inf if want > 0 else np.inf)
        k = int(np.argmax(cand)) if want > 0 else int(np.argmin(cand))
        assert abs(tot - pvar[mask].sum()) <= 1e-11 * abs(pvar[mask].sum())
        assert abs(pvar[ij] - cand.ravel()[k]) <= 1e-12 * abs(cand.ravel()[k])
        assert ij == (k // m, k % m)
    # the same selection with this rank's rows RESIDENT and only the item-side blocks exchanged (ShardedGrid):
    # the state changes between steps (here: twice), the cross block and the mask stay where they are
    lo, hi = sh.block_range(n, comm.rank, comm.world)
    v_lo, v_hi = sh.block_range(m, comm.rank, comm.world)
    grid = sh.ShardedGrid(eng, comm, n, m, R[:n, n:][lo:hi], mask[lo:hi])
    diag = np.diag(R).copy()
    for scale in (1., .5):
        mean_s = mean * scale
        _, pv_s = orc.mn_pred_means_vars(n, m, mean_s, R, Om)
        for want in (1, -1):
            tot, bv, ij = grid.select(mean_s[lo:hi], mean_s[n + v_lo:n + v_hi], diag[lo:hi], diag[n + v_lo:n + v_hi],
                                      Om, want)
            cand = np.where(mask, pv_s, -np.inf if want > 0 else np.inf)
            k = int(np.argmax(cand)) if want > 0 else int(np.argmin(cand))
            assert abs(tot - pv_s[mask].sum()) <= 1e-11 * abs(pv_s[mask].sum())
            assert ij == (k // m, k % m) and abs(bv - cand.ravel()[k]) <= 1e-12 * abs(cand.ravel()[k])
    # exact ties across the shard boundary resolve to the lowest flat index
    mean_t = np.tile(mean[:1], (n + m, 1))
    R_t = np.eye(n + m)
    tot, bv, ij = sh.sharded_select(eng, comm, n, m, mean_t, R_t[:n, n:], np.diag(R_t).copy(), Om, None, 1)
    assert ij == (0, 0)
    m2 = np.zeros((n, m), bool)
    m2[n - 1, 3] = m2[n - 2, m - 1] = True      # both near the end of the grid
    tot, bv, ij = sh.sharded_select(eng, comm, n, m, mean_t, R_t[:n, n:], np.diag(R_t).copy(), Om, m2, 1)
    assert ij == (n - 2, m - 1)
    # block layout
    los = [sh.block_range(n, r, comm.world) for r in range(comm.world)]
    assert los[0][0] == 0 and los[-1][1] == n and all(a[1] == b[0] for a, b in zip(los, los[1:]))
    full = comm.all_gather_rows(U[slice(*sh.block_range(n, comm.rank, comm.world))], n)
    np.testing.assert_array_equal(full, U)
    return True
