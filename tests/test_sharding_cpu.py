"""world_size-2 gloo tests of the multi-GPU path on CPU: the sharding / exchange logic of
active_matrix_factorization_amd.sharding with the compute supplied by the test-only
OracleEngine.  (The same cases run with the HIP engine under -m gpu.)"""
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
        import torch.distributed as dist
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                                world_size=world)
        from active_matrix_factorization_amd import sharding as sh
        from engines import OracleEngine
        import sharding_cases
        comm = sh.Comm()
        assert comm.world == world and comm.rank == rank
        sharding_cases.check_all(lambda r, n, m, d: OracleEngine(r, n, m, d), comm)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_path_gloo(world):
    import multiprocessing as mp   # NOT torch.multiprocessing: the parent must not import torch
                                   # after the HIP libraries are loaded (see sharding.Comm)
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
