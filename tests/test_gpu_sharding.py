"""The multi-GPU path with the real HIP engine on the one-GPU test box: a single rank with a real
RCCL communicator (world 1: every collective of csrc/comm.hip executes on the stream), and two
ranks sharing device 0 running the same cases as the CPU socket test (RCCL refuses two ranks on
one device, so their device blocks are staged through the socket transport)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _make_engine_factory(ctx):
    from active_matrix_factorization_amd import sharding as sh
    return lambda r, n, m, d: sh.HipEngine(ctx, r, n, m, d)


def test_hip_engine_single_rank_rccl():
    from active_matrix_factorization_amd import _lib, sharding as sh
    import sharding_cases
    ctx = _lib.Context(0)
    comm = sh.Comm(0, 1, ctx=ctx)
    assert comm.has_rccl
    # the collectives themselves, on device pointers
    a = ctx.array(np.arange(12, dtype=np.float64).reshape(3, 4))
    g = comm.all_gather_blocks(a)
    assert g.shape == (1, 3, 4) and np.array_equal(g.to_host()[0], a.to_host())
    r = comm.all_reduce_sum(ctx.array(np.array([1.5, -2., 7.])))
    assert np.array_equal(r.to_host(), [1.5, -2., 7.])
    # the grouped exchange of a fit_lls trial (apmf_comm_allgather_multi: one RCCL group, one launch)
    b3 = [ctx.array(np.arange(6.).reshape(2, 3)), ctx.array(np.array([7., 8., 9., 10.])), ctx.array(np.zeros((0, 3)))]
    g3 = comm.all_gather_multi(b3)
    assert [x.shape for x in g3] == [(1, 2, 3), (1, 4), (1, 0, 3)]
    assert np.array_equal(g3[0].to_host()[0], np.arange(6.).reshape(2, 3)) and np.array_equal(g3[1].to_host()[0], [7., 8., 9., 10.])
    comm.barrier()
    sharding_cases.check_all(_make_engine_factory(ctx), comm)
    comm.close()
    assert 'torch' not in sys.modules


def test_comm_failure_is_loud():
    """A communicator that cannot be created is an error, never a silent single-rank fallback."""
    from active_matrix_factorization_amd import _lib, sharding as sh
    with pytest.raises(ValueError):
        sh.Comm(0, 1, rccl=True)                 # RCCL needs a Context
    with pytest.raises((TimeoutError, OSError)):
        sh.Comm(1, 2, '127.0.0.1', _free_port(), timeout=1.5)     # nobody listens


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        from active_matrix_factorization_amd import _lib, sharding as sh
        import sharding_cases
        ctx = _lib.Context(0)
        comm = sh.Comm(rank, world, '127.0.0.1', port, ctx=ctx, rccl=False)
        sharding_cases.check_all(_make_engine_factory(ctx), comm)
        comm.barrier()
        comm.close()
        assert 'torch' not in sys.modules
        q.put((rank, 'ok'))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def test_hip_engine_two_ranks_one_gpu():
    import multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context('spawn')      # HIP contexts are created in the children only
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(msg == 'ok' for _, msg in res), '\n'.join(str(x) for x in res)


def _lookahead_worker(rank, world, port, q):
    try:
        import random
        from active_matrix_factorization_amd import _lib, sharding as sh, mn_active_pmf as mn
        from conftest import load_golden
        g = load_golden('lookahead_c1_mn')
        ctx = _lib.default_context()
        comm = sh.Comm(rank, world, '127.0.0.1', port, ctx=ctx, rccl=False) if world > 1 else None
        a = mn.MNActivePMF(g['ratings'].copy(), latent_d=2, rating_values=(1, 2, 3, 4, 5), discrete_expectations=True)
        a.users, a.items = g['users'].copy(), g['items'].copy()
        a.mean, a.cov_useritems, a.cov_latents = g['mean'].copy(), g['cov_useritems'].copy(), g['cov_latents'].copy()
        a.lookahead_comm = comm
        pool = sorted(a.unrated)[:5]
        vals = a._get_key_vals(pool, mn.KEY_FUNCS['total-variance'])
        if comm is not None:
            comm.barrier()
            comm.close()
        assert 'torch' not in sys.modules
        q.put((rank, vals))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def test_lookahead_hypotheses_sharded_over_ranks():
    """BASELINE.json configs[3] shards the lookahead's candidates over the GPUs: the hypotheses are
    dealt to the ranks in blocks and the values all-gathered; two ranks (sharing the one GPU of the
    test box, exchange staged through the host) return on every rank what one rank computes."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_lookahead_worker, args=(0, 1, 0, q))
    p.start()
    _, single = q.get(timeout=300)
    p.join(30)
    assert isinstance(single, list), single
    port = _free_port()
    procs = [ctx.Process(target=_lookahead_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(30)
    for r in range(2):
        assert isinstance(res[r], list), res[r]
        assert res[r] == single
