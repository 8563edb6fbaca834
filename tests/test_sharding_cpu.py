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


def _worker(rank, world, port, q, dims=None):
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
        sharding_cases.check_all(lambda r, n, m, d: OracleEngine(r, n, m, d), comm, dims=dims)
        # the grouped exchange of a fit_lls trial: several blocks, one message
        a, b = np.full((2, 3), float(rank)), np.arange(4.) + 10 * rank
        ga, gb = comm.all_gather_multi([a, b])
        assert ga.shape == (world, 2, 3) and gb.shape == (world, 4)
        assert all((ga[r] == r).all() and gb[r][0] == 10 * r for r in range(world))
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


def _run_world(world, dims=None, timeout=400):
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, dims)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(msg == 'ok' for _, msg in res), '\n'.join(str(x) for x in res)


def test_sharded_path_world8():
    """The driver's scaling run is 8 ranks: every sharded algorithm at world 8 over the socket transport
    (23 users / 31 items: blocks of 3 and 4 rows, the last ones short)."""
    _run_world(8)


def test_sharded_path_fewer_rows_than_ranks():
    """n < world and m < world: most ranks own an EMPTY row block of one side or both (zero-sized blocks in
    every exchange, empty sweeps, partial sums of zero)."""
    _run_world(8, dims=dict(n=5, m=6, d=3, N=40))


def test_rendezvous_rejects_strangers():
    """Rank 0's accept loop: a connection that does not speak the handshake, announces a rank outside
    1..world-1 or a rank that is already connected is dropped, and the run still comes up."""
    import multiprocessing as mp
    import struct
    import threading
    import time
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(0, 2, port, q))]
    procs[0].start()

    def send(payload):
        for _ in range(200):
            try:
                s = socket.create_connection(('127.0.0.1', port), timeout=1.)
                s.sendall(payload)
                time.sleep(.1)
                s.close()
                return
            except OSError:
                time.sleep(.05)
    hello = b'APMFCOMM1t'
    send(b'\xff' * 8)                                                   # a garbage length
    send(struct.pack('<Q', 5) + b'hello')                               # not the handshake
    for bad in (b':0', b':2', b':-1', b':x'):
        msg = hello + bad
        send(struct.pack('<Q', len(msg)) + msg)                         # rank out of range / not a number
    procs.append(ctx.Process(target=_worker, args=(1, 2, port, q)))
    procs[1].start()
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


def _fallback_worker(rank, world, port, q):
    try:
        from active_matrix_factorization_amd import sharding as sh

        class NoDevice(object):      # stands where a Context would: no GPU in this process
            _h = None
            device = None

            def sync(self):
                pass
        comm = sh.Comm(rank, world, '127.0.0.1', port, ctx=NoDevice(), rccl=True, tag='f', fallback=True)
        assert not comm.has_rccl and comm.rccl_error and 'failed to start RCCL' in comm.rccl_error, comm.rccl_error
        assert str(list(range(world))) in comm.rccl_error          # every rank reported its failure to every other
        got = comm.all_gather_scalars([rank])                      # and the host transport carries the exchanges
        assert np.array_equal(got.reshape(-1), np.arange(world))
        comm.barrier()
        comm.close()
        q.put((rank, 'ok'))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def test_rccl_fallback_is_agreed_and_labelled():
    """bench.py's safety net for a first multi-GPU run (Comm(..., fallback=True)): RCCL cannot come up here (no GPU:
    the unique id / communicator calls fail), every rank learns over the sockets that not all of them did, all drop
    RCCL TOGETHER, `rccl_error` names the ranks, and the host-staged exchanges work.  (Without fallback=True the same
    failure raises: the product has no silent host path.)"""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(msg == 'ok' for _, msg in res), '\n'.join(str(x) for x in res)
    from active_matrix_factorization_amd import sharding as sh

    class NoDevice(object):
        _h = None
    with pytest.raises(Exception):
        sh.Comm(0, 1, ctx=NoDevice(), rccl=True)                   # strict by default


def _root_fails_worker(rank, world, port, q):
    try:
        from active_matrix_factorization_amd import _lib, sharding as sh
        created = []

        class FakeComm(object):          # stands where libapmf_comm.so would
            def apmf_comm_unique_id(self, ptr):
                return -2                # rank 0 cannot draw the id

            def apmf_comm_create(self, *a):
                created.append(a)        # must never be reached: with a zeroed id ncclCommInitRank would wait
                return 0

            def apmf_comm_destroy(self, h):
                return 0
        _lib.load_comm = lambda: FakeComm()
        _lib._check = lambda rc: (_ for _ in ()).throw(_lib.ApmfError(rc, 'fake failure')) if rc else None

        class NoDevice(object):
            _h = None
            device = None

            def sync(self):
                pass
        comm = sh.Comm(rank, world, '127.0.0.1', port, ctx=NoDevice(), rccl=True, tag='g', fallback=True)
        assert not created, 'apmf_comm_create was called although rank 0 had no id'
        assert not comm.has_rccl and str(list(range(world))) in comm.rccl_error, comm.rccl_error
        if rank:
            assert 'rank 0 has no unique id' in comm.rccl_error, comm.rccl_error
        comm.barrier()
        comm.close()
        q.put((rank, 'ok'))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def test_rccl_fallback_when_only_rank0_fails_to_draw_the_id():
    """Rank 0's verdict on the unique-id step travels with the id: when it failed, NO rank calls
    apmf_comm_create (ncclCommInitRank with an all-zero id would wait, not fail) and all fall back together."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_root_fails_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(msg == 'ok' for _, msg in res), '\n'.join(str(x) for x in res)
