"""world_size-2 / -3 tests of the multi-GPU path on CPU: the sharding / exchange logic of
active_matrix_factorization_amd.sharding over its TCP-socket transport (the product exchanges
device blocks with RCCL instead; there is no PyTorch in either), with the compute supplied by the
test-only OracleEngine.  (The same cases run with the HIP engine under -m gpu.)"""
import os
import socket
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        from active_matrix_factorization_amd import sharding as sh
        from engines import OracleEngine
        import sharding_cases
        comm = sh.Comm(rank, world, '127.0.0.1', port, tag='t')
        assert comm.world == world and comm.rank == rank and not comm.has_rccl
        # the host collectives themselves
        got = comm.all_gather_scalars([rank, 2. * rank])
        assert np.array_equal(got, [[r, 2. * r] for r in range(world)])
        assert np.array_equal(comm.all_reduce_sum(np.array([1., rank])), [world, world * (world - 1) / 2])
        assert comm.bcast_bytes(b'id%d' % rank) == b'id0'
        rows = comm.all_gather_rows(np.full((sh.block_range(7, rank, world)[1] - sh.block_range(7, rank, world)[0], 2),
                                            float(rank)), 7)
        assert rows.shape == (7, 2) and rows[0, 0] == 0. and rows[-1, 0] == (7 - 1) // sh.block_size(7, world)
        sharding_cases.check_all(lambda r, n, m, d: OracleEngine(r, n, m, d), comm)
        comm.barrier()
        comm.close()
        assert 'torch' not in sys.modules       # the multi-process path needs no PyTorch
        q.put((rank, 'ok'))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_path_sockets(world):
    """world-2 / world-3 run of every sharded algorithm on CPU: TCP-socket transport, compute from
    the test-only oracle engine.  (The GPU runs replace the transport by RCCL and the engine by the
    HIP one; the logic in sharding.py is the same.)"""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(msg == 'ok' for _, msg in res), '\n'.join(str(x) for x in res)


def test_port_search_skips_busy_port():
    """Rank 0 listens on the first FREE port from the given one on (torchrun's store occupies
    MASTER_PORT itself); the other ranks find it through the handshake."""
    import multiprocessing as mp
    busy = socket.socket()
    busy.bind(('127.0.0.1', 0))
    busy.listen(1)
    port = busy.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(30)
    busy.close()
    assert all(msg == 'ok' for _, msg in res), '\n'.join(str(x) for x in res)


def test_single_process_comm():
    from active_matrix_factorization_amd import sharding as sh
    from engines import OracleEngine
    import sharding_cases
    comm = sh.Comm()
    assert comm.world == 1
    sharding_cases.check_all(lambda r, n, m, d: OracleEngine(r, n, m, d), comm)


def test_block_ranges():
    from active_matrix_factorization_amd import sharding as sh
    for n in (1, 7, 8, 943, 1682):
        for w in (1, 2, 3, 8):
            rows = [sh.block_range(n, r, w) for r in range(w)]
            assert sum(hi - lo for lo, hi in rows) == n
            assert all(hi - lo <= sh.block_size(n, w) for lo, hi in rows)
