"""The multi-GPU path with the real HIP engine: single rank, and two ranks (gloo rendezvous,
both on device 0 of the one-GPU test box) running the same cases as the CPU gloo test."""
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _make_engine_factory():
    from active_matrix_factorization_amd import _lib, sharding as sh
    ctx = _lib.Context(0)
    return lambda r, n, m, d: sh.HipEngine(ctx, r, n, m, d)


def test_hip_engine_single_rank():
    from active_matrix_factorization_amd import sharding as sh
    import sharding_cases
    sharding_cases.check_all(_make_engine_factory(), sh.Comm())


def _worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                                world_size=world)
        from active_matrix_factorization_amd import sharding as sh
        import sharding_cases
        sharding_cases.check_all(_make_engine_factory(), sh.Comm())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def test_hip_engine_two_ranks_one_gpu():
    import multiprocessing as mp   # NOT torch.multiprocessing: the parent must not import torch
                                   # after the HIP libraries are loaded (see sharding.Comm)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')      # HIP contexts are created in the children only
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(msg == 'ok' for _, msg in res), '\n'.join(str(x) for x in res)
