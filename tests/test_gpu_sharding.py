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
