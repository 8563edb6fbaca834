"""Multi-GPU sharding of the hot path: one process per GPU, row blocks of
users / items / candidate-grid rows per rank.  No PyTorch anywhere: the exchange steps are RCCL
collectives on device pointers behind the C-ABI (libapmf_comm.so, csrc/comm.hip), and the
rendezvous - shipping rank 0's 128-byte ncclUniqueId to the other ranks - is a plain TCP socket,
which also carries the host-side collectives of the CPU tests (world 2 / 3 without any GPU).

What shards, and what is exchanged:
  * candidate grid (K2): rank g sweeps the user rows of its block against all items.  No
    collective during the sweep; at the end an all-gather of one (value, flat index) pair
    and one partial sum per rank, merged with the single-device tie rule (lowest flat index).
  * lookahead hypotheses: independent; rank g takes every world-th chunk and the per-hypothesis
    results are all-gathered (bench.py's weak-scaling step).
  * PMF fit (K1): rank g owns a block of user rows and a block of item rows.  It computes
    grad_u for its users and grad_v for its items (each from the full ratings container,
    no reduction needed) and its blocks of the next trial point ON THE DEVICE.  ONE grouped
    all-gather per trial carries, device to device, this rank's three partial sums of the trial's
    log-likelihood together with its blocks of BOTH possible next trial points (the one that follows
    an accepted trial and the one that follows a rejected one: the decision lags, as in the
    single-GPU k_fit_iter); every rank adds the partial sums in rank order - the same bits, hence
    the same accept / reject decision everywhere, no all-reduce - and keeps the branch that applies.
  * Gibbs (K3): a half-sweep is split by row block, then the freshly sampled side is
    all-gathered before the other side's half-sweep (which needs the NEW values).
The functions here are engine-agnostic: `engine` is any object with the block methods of
HipEngine (the product, over libapmf_hip.so; arrays are DeviceArrays that stay resident between
steps) - the CPU tests plug in an engine over the oracle whose arrays are NumPy.

The reference's analogue is the forked multiprocessing.Pool that pickles whole model objects to
its workers (active_pmf.py:739-770, bayes_pmf.py:385-422).
"""
import os
import socket
import struct
import sys
import time

import numpy as np

try:
    from . import _lib
except ImportError:
    import _lib


def block_size(n, world):
    return (n + world - 1) // world


def block_range(n, rank, world):
    """Rows [lo, hi) owned by `rank`: uniform blocks of ceil(n / world) rows (the last ones
    may be short or empty) - equal sizes keep the all-gather a plain one."""
    bs = block_size(n, world)
    lo = min(n, rank * bs)
    return lo, min(n, lo + bs)


# ---------------------------------------------------------------------------- host transport
_MAGIC = b'APMFCOMM1'


def _send_msg(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_exact(sock, count):
    buf = bytearray()
    while len(buf) < count:
        chunk = sock.recv(count - len(buf))
        if not chunk:
            raise ConnectionError('peer closed the connection')
        buf += chunk
    return bytes(buf)


_MAX_MSG = 1 << 34       # 16 GiB: far above any block this path exchanges, far below a garbage length


def _recv_msg(sock, limit=_MAX_MSG):
    (count,) = struct.unpack('<Q', _recv_exact(sock, 8))
    if count > limit:
        raise ConnectionError('message length %d exceeds the limit %d' % (count, limit))
    return _recv_exact(sock, count)


class Comm(object):
    """Rank `rank` of `world` processes on one node.

    Host side: a star of TCP connections to rank 0 (world == 1 opens nothing), used for the
    rendezvous and for collectives on NumPy arrays.  Device side (when a Context is given): an
    RCCL communicator created from the id rank 0 drew; collectives on DeviceArrays then run on the
    context's stream over xGMI and nothing is staged through the host.  Without it DeviceArrays
    are staged through the host socket (two ranks sharing one GPU in a rehearsal: RCCL refuses
    two ranks on one device).

    `port`: rank 0 listens on the first free port of [port, port + span); the others try them in
    order until the handshake (magic + run tag) succeeds, so a busy port - e.g. torchrun's own
    store on MASTER_PORT - is skipped."""

    def __init__(self, rank=0, world=1, addr='127.0.0.1', port=None, ctx=None, rccl=None, tag='',
                 timeout=300., span=32, fallback=False):
        self.rank, self.world = int(rank), int(world)
        self.ctx = ctx
        self._peers = None        # rank 0: sockets of ranks 1..world-1
        self._root = None         # others: socket to rank 0
        self._server = None
        self._nccl = None
        self._libc = None
        self.rccl_error = None    # fallback=True: why the job runs on the socket transport instead of RCCL
        if self.world > 1:
            if port is None:
                raise ValueError('a port is needed for world > 1')
            self._connect(addr, int(port), (tag or '').encode(), timeout, span)
        want_rccl = (ctx is not None) if rccl is None else bool(rccl)
        if want_rccl:
            if ctx is None:
                raise ValueError('the RCCL communicator needs a Context')
            self._start_rccl(ctx, fallback)

    def _start_rccl(self, ctx, fallback):
        """The RCCL communicator over the socket rendezvous.  A failure is fatal for the caller (the product has no
        host path for a broken fabric) - unless `fallback`: then the ranks agree, over the sockets, whether EVERY one
        of them came up; if not, all drop RCCL together, the exchanges run host-staged over the sockets and
        `rccl_error` says why (bench.py does this so that a first multi-GPU run still reports, loudly labelled).
        Rank 0's own verdict on the unique-id step travels WITH the id: when it failed nobody calls
        ncclCommInitRank at all (an all-zero id would not fail there, it would wait).  What the agreement cannot
        cover is a PARTIAL failure inside ncclCommInitRank itself - some ranks return an error while the others wait
        in it for them: those never reach the flag exchange, so that case ends in the socket timeout of the ranks
        that did fail (a labelled ConnectionError), not in a fallback."""
        import ctypes as C
        err = None
        ident = np.zeros(129, np.uint8)       # [rank 0's ok flag | ncclUniqueId]
        try:
            self._libc = _lib.load_comm()
            if self.rank == 0:
                _lib._check(self._libc.apmf_comm_unique_id(ident[1:].ctypes.data))
                ident[0] = 1
        except Exception as e:      # noqa: BLE001 - reported, and fatal unless fallback
            if not fallback:
                raise
            err = 'unique id: %s' % e
        ident = np.frombuffer(self.bcast_bytes(ident.tobytes()), np.uint8).copy()
        root_ok = bool(ident[0])
        if not root_ok and not fallback:
            raise _lib.ApmfError(_lib.EHIP, 'rank 0 could not draw the RCCL unique id')
        if not root_ok and err is None:
            err = 'rank 0 has no unique id'
        uid = np.ascontiguousarray(ident[1:])
        h = C.c_void_p()
        if err is None:
            try:
                _lib._check(self._libc.apmf_comm_create(getattr(ctx, '_h', None), self.rank, self.world,
                                                        uid.ctypes.data, C.byref(h)))
            except Exception as e:  # noqa: BLE001
                if not fallback:
                    raise
                err = 'communicator: %s' % e
        if not fallback:
            self._nccl = h.value
            return
        flags = self.all_gather_blocks(np.array([0. if err is None else 1.]))      # sockets: _nccl is still None
        if flags.sum() == 0:
            self._nccl = h.value
            return
        if err is None and h.value:
            self._libc.apmf_comm_destroy(h.value)
        bad = [r for r in range(self.world) if flags.reshape(-1)[r] != 0]
        self.rccl_error = 'ranks %s failed to start RCCL%s' % (bad, '' if err is None else ' (this rank: %s)' % err)

    # -- set-up / tear-down -------------------------------------------------------------------
    def _connect(self, addr, port, tag, timeout, span):
        hello = _MAGIC + tag
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            for p in range(port, port + span):
                try:
                    srv.bind((addr, p))
                    break
                except OSError:
                    continue
            else:
                raise OSError('no free port in [%d, %d)' % (port, port + span))
            srv.listen(self.world)
            srv.settimeout(timeout)
            self._server = srv
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                try:
                    msg = _recv_msg(conn, limit=4096)         # anybody can connect to this port
                    if not msg.startswith(hello + b':'):
                        conn.close()
                        continue
                    r = int(msg[len(hello) + 1:])
                    if not (1 <= r < self.world) or r in peers:
                        conn.close()                          # not a rank of this run, or a duplicate
                        continue
                    _send_msg(conn, hello)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    peers[r] = conn
                except (OSError, ValueError, ConnectionError, struct.error):
                    conn.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            while True:
                for p in range(port, port + span):
                    s = None
                    try:
                        s = socket.create_connection((addr, p), timeout=2.)
                        s.settimeout(5.)
                        _send_msg(s, hello + b':%d' % self.rank)
                        if _recv_msg(s, limit=4096) == hello:
                            s.settimeout(timeout)
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self._root = s
                            return
                        s.close()
                    except (OSError, ConnectionError, struct.error):
                        if s is not None:
                            s.close()
                if time.time() > deadline:
                    raise TimeoutError('rank %d: no rank 0 on %s:[%d, %d)' % (self.rank, addr, port, port + span))
                time.sleep(.05)

    def close(self):
        if self._nccl is not None and self._libc is not None:
            self._libc.apmf_comm_destroy(self._nccl)
            self._nccl = None
        for s in (self._peers or []) + [self._root, self._server]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers = self._root = self._server = None

    @property
    def has_rccl(self):
        return self._nccl is not None

    # -- host collectives ----------------------------------------------------------------------
    def allgather_bytes(self, payload):
        """Every rank's payload, in rank order, on every rank."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv_msg(s) for s in self._peers]
            blob = b''.join(struct.pack('<Q', len(p)) + p for p in parts)
            for s in self._peers:
                _send_msg(s, blob)
            return parts
        _send_msg(self._root, payload)
        blob = _recv_msg(self._root)
        parts, off = [], 0
        for _ in range(self.world):
            (count,) = struct.unpack_from('<Q', blob, off)
            parts.append(blob[off + 8:off + 8 + count])
            off += 8 + count
        return parts

    def bcast_bytes(self, payload):
        """Rank 0's payload on every rank."""
        return self.allgather_bytes(payload if self.rank == 0 else b'')[0]

    def barrier(self):
        """All ranks have arrived; with a Context, their streams have drained first."""
        if self.ctx is not None:
            if self._nccl is not None:
                _lib._check(self._libc.apmf_comm_barrier(self.ctx._h, self._nccl))
            else:
                self.ctx.sync()
        if self.world > 1:
            self.allgather_bytes(b'')

    # -- collectives on arrays: NumPy -> NumPy (socket), DeviceArray -> DeviceArray (RCCL) ------
    def all_gather_blocks(self, block):
        """Equal-sized blocks -> one array of world x block on every rank (rank order)."""
        if isinstance(block, _lib.DeviceArray):
            out = self.ctx.empty((self.world,) + block.shape, block.dtype)
            if self._nccl is not None:
                _lib._check(self._libc.apmf_comm_allgather(self.ctx._h, self._nccl, block.ptr, out.ptr,
                                                           block.nbytes))
            elif self.world == 1:
                self.ctx.copy(out, block, block.nbytes)
            else:
                out.upload(self.all_gather_blocks(block.to_host()))
            return out
        block = np.ascontiguousarray(block)
        parts = self.allgather_bytes(block.tobytes())
        return np.frombuffer(b''.join(parts), block.dtype).reshape((self.world,) + block.shape).copy()

    def all_gather_multi(self, blocks):
        """Several equal-sized-per-rank blocks exchanged as ONE step: a list of (world,) + block.shape arrays.
        DeviceArrays go through one RCCL group (a single launch on the context's stream); NumPy arrays - and
        DeviceArrays in a rehearsal without RCCL - through one message of the socket transport."""
        blocks = list(blocks)
        if not blocks:
            return []
        if isinstance(blocks[0], _lib.DeviceArray):
            outs = [self.ctx.empty((self.world,) + b.shape, b.dtype) for b in blocks]
            if self._nccl is not None:
                import ctypes as C
                k = len(blocks)
                send = (C.c_void_p * k)(*[b.ptr for b in blocks])
                recv = (C.c_void_p * k)(*[o.ptr for o in outs])
                size = (C.c_size_t * k)(*[b.nbytes for b in blocks])
                _lib._check(self._libc.apmf_comm_allgather_multi(self.ctx._h, self._nccl, k, send, recv, size))
            elif self.world == 1:
                for o, b in zip(outs, blocks):
                    self.ctx.copy(o, b, b.nbytes)
            else:
                for o, h in zip(outs, self.all_gather_multi([b.to_host() for b in blocks])):
                    o.upload(h)
            return outs
        blocks = [np.ascontiguousarray(b) for b in blocks]
        if self.world == 1:
            return [b[None].copy() for b in blocks]
        parts = self.allgather_bytes(b''.join(b.tobytes() for b in blocks))
        outs, off = [], 0
        for b in blocks:
            outs.append(np.stack([np.frombuffer(p, b.dtype, b.size, off).reshape(b.shape) for p in parts]))
            off += b.nbytes
        return outs

    def all_gather_rows(self, block, n):
        """Row blocks (block_range layout, each padded to block_size rows) of an (n, ...) array ->
        the array on every rank: NumPy (n, ...) / DeviceArray (world * block_size, ...)."""
        bs = block_size(n, self.world)
        if isinstance(block, _lib.DeviceArray):
            if block.shape[0] != bs:
                raise ValueError('device blocks must be padded to block_size rows')
            full = self.all_gather_blocks(block)
            full.shape = (self.world * bs,) + block.shape[1:]
            return full
        block = np.asarray(block)
        padded = np.zeros((bs,) + block.shape[1:], block.dtype)
        padded[:block.shape[0]] = block
        return self.all_gather_blocks(padded).reshape((self.world * bs,) + block.shape[1:])[:n]

    def all_reduce_sum(self, a):
        """Element-wise float64 sum over the ranks.  Host arrays are summed in rank order (the same
        bits on every rank); a DeviceArray is reduced in place by RCCL."""
        if isinstance(a, _lib.DeviceArray):
            if self._nccl is not None:
                _lib._check(self._libc.apmf_comm_allreduce_f64(self.ctx._h, self._nccl, a.ptr, a.ptr, a.size))
            elif self.world > 1:
                a.upload(self.all_reduce_sum(a.to_host()))
            return a
        a = np.ascontiguousarray(a, np.float64)
        if self.world == 1:
            return a
        total = np.zeros_like(a)
        for part in self.all_gather_blocks(a):
            total += part
        return total

    def all_gather_scalars(self, row):
        """One small float64 row per rank -> (world, len(row)) on every rank."""
        row = np.asarray(row, np.float64).reshape(-1)
        return self.all_gather_blocks(row)


def comm_from_env(ctx=None, port_offset=1, fallback=False):
    """Comm of a process started by `python -m torch.distributed.run` / torchrun (used purely as
    a process launcher: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) or by hand.  The listening port
    is searched from MASTER_PORT + port_offset on (MASTER_PORT itself belongs to the launcher)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '29500')) + port_offset
    tag = os.environ.get('TORCHELASTIC_RUN_ID', '') + ':' + os.environ.get('MASTER_PORT', '')
    t0 = time.time()
    try:
        comm = Comm(rank, world, addr, port, ctx=ctx, tag=tag, fallback=fallback)
    except Exception as e:
        sys.stderr.write('[apmf comm] rank %d/%d device %s: communicator set-up FAILED after %.1f s (%s:%d+): %s\n'
                         % (rank, world, getattr(ctx, 'device', None), time.time() - t0, addr, port, e))
        sys.stderr.flush()
        raise
    # one line per rank, so that a first multi-GPU run that goes wrong can be read from the launcher's tail
    version = None
    if comm.has_rccl:
        import ctypes as C
        v = C.c_int(0)
        if comm._libc.apmf_comm_version(C.byref(v)) == 0:
            version = v.value
    sys.stderr.write('[apmf comm] rank %d/%d device %s: %s up in %.2f s (rendezvous %s:%d+, RCCL version %s)\n'
                     % (rank, world, getattr(ctx, 'device', None), 'RCCL communicator' if comm.has_rccl else
                        'socket transport' + (' (FALLBACK: %s)' % comm.rccl_error if comm.rccl_error else ''),
                        time.time() - t0, addr, port, version))
    sys.stderr.flush()
    return comm


# ---------------------------------------------------------------------------- the product's engine
class HipEngine(object):
    """Block operations on this rank's GPU through libapmf_hip.so (no fallback).  Arrays are
    DeviceArrays: factors stay resident on the device between the steps of a fit."""

    def __init__(self, ctx, ratings, n, m, d, sigma_sq=1., sigma_u_sq=10., sigma_v_sq=10.,
                 subtract_mean=False, mean_rating=0.):
        self.ctx, self.n, self.m, self.d = ctx, n, m, d
        self.hyper = (float(sigma_sq), float(sigma_u_sq), float(sigma_v_sq), bool(subtract_mean),
                      float(mean_rating))
        self.rt = ctx.ratings(ratings, n, m) if ratings is not None else None
        self._grad = {}

    # -- residency
    def put(self, host, rows=None):
        """(n, d) host array -> device array with `rows` >= n rows (zero padded)."""
        host = np.ascontiguousarray(host, np.float64)
        rows = host.shape[0] if rows is None else rows
        arr = self.ctx.zeros((rows,) + host.shape[1:])
        self.ctx.write_bytes(arr, 0, host)
        return arr

    def get(self, arr, n):
        return arr.to_host()[:n]

    # -- K1
    def grad_rows(self, side, lo, hi, U, V, slot=0):
        """Rows [lo, hi) of grad_u (side 0) / grad_v (side 1) in a full-size device array; two slots,
        so that the gradient at the current point and the speculative one at the trial point coexist."""
        ctx, d = self.ctx, self.d
        sig, su, sv, sub, mr = self.hyper
        rows = self.n if side == 0 else self.m
        g = self._grad.get((side, slot))
        if g is None:
            g = self._grad[(side, slot)] = ctx.empty((rows, d))
        ctx.pmf_gradient_rows(self.rt, side, lo, hi, U, V, d, sig, su, sv, sub, mr, g)
        return g

    def trial_block(self, X, G, lo, hi, lr, bs):
        """X[lo:hi] + lr * G[lo:hi] as a block of bs rows (zero padded), on the device."""
        d = self.d
        out = self.ctx.zeros((bs, d))
        if hi > lo:
            self.ctx.axpy(out, X, lr, G, (hi - lo) * d, 0, lo * d, lo * d)
        return out

    def ll_terms(self, U, V, u_lo, u_hi, v_lo, v_hi):
        _, _, _, sub, mr = self.hyper
        return self.ctx.pmf_ll_terms(self.rt, U, V, self.d, sub, mr, u_lo, u_hi, v_lo, v_hi)

    def ll_partial(self, U, V, u_lo, u_hi, v_lo, v_hi):
        """This rank's three partial sums of the log-likelihood as a 4-double DEVICE record (no host
        synchronisation): it travels inside the trial's collective."""
        _, _, _, sub, mr = self.hyper
        rec = self.ctx.zeros(4)
        self.ctx.pmf_ll_terms_dev(self.rt, U, V, self.d, sub, mr, u_lo, u_hi, v_lo, v_hi, rec)
        return rec

    def small_to_host(self, arr):
        return arr.to_host()

    def as_rows(self, gathered):
        """(world, bs, d) gathered blocks -> the (world * bs, d) array they are, without a copy."""
        gathered.shape = (gathered.shape[0] * gathered.shape[1],) + tuple(gathered.shape[2:])
        return gathered

    # -- K3
    def gibbs_rows(self, side, lo, hi, oth, alpha, mu, beta, mean_sub, z, bs):
        """The freshly sampled rows [lo, hi) of `side` as a block of bs rows."""
        ctx, d = self.ctx, self.d
        rows = self.n if side == 0 else self.m
        out = ctx.zeros((max(rows, lo + bs), d))
        ctx.gibbs_half_sweep_rows(self.rt, side, lo, hi, oth, d, ctx.array(alpha), ctx.array(mu),
                                  float(beta), float(mean_sub), z, out)
        blk = ctx.empty((bs, d))
        ctx.copy(blk, out, bs * d * 8, src_off=lo * d * 8)
        return blk

    # -- K2: a block of user rows against all items, float64
    def score_rows(self, mean_users, mean_items, cross_rows, diag_users, diag_items, omega, mask_rows,
                   want):
        ctx, d = self.ctx, self.d
        nl, m = cross_rows.shape
        if nl == 0:
            return 0., 0., -1
        cross = ctx.upload_grid(cross_rows, np.float64)
        ld = cross.shape[1]
        dmask = None if mask_rows is None else ctx.upload_grid(mask_rows.astype(np.uint8), np.uint8, ld=ld)
        dsum, bv, bi = ctx.empty(1), ctx.empty(1), ctx.empty(1, np.int64)
        ctx.mn_score_grid(np.float64, 1, nl, m, d, ld, ctx.array(np.vstack([mean_users, mean_items])),
                          cross, ctx.array(np.concatenate([diag_users, diag_items])), ctx.array(omega),
                          mask=dmask, dsum=dsum, want=want, best_val=bv, best_idx=bi)
        return float(dsum.to_host()[0]), float(bv.to_host()[0]), int(bi.to_host()[0])


    # -- K2 with this rank's rows of the state resident on the device (ShardedGrid)
    def hold_grid(self, rows_2d, dtype=np.float64):
        """(rows, m) host block -> resident zero-padded device grid (rows, ld)"""
        rows_2d = np.asarray(rows_2d)
        if rows_2d.shape[0] == 0:
            return None
        return self.ctx.upload_grid(rows_2d.astype(dtype, copy=False), dtype)

    def hold(self, host):
        return self.ctx.array(np.ascontiguousarray(host, np.float64))

    def score_resident(self, cross, mask, mean_u_rows, mean_items, diag_u_rows, diag_items, omega, m, want):
        """The sweep of this rank's resident rows: everything stays on the device; returns the device record
        [sum f64, best value f64, best local flat index i64 (-1: none), 0]."""
        ctx, d = self.ctx, self.d
        rec = ctx.zeros(4)
        if cross is None:
            ctx.write_bytes(rec, 16, np.array([-1], np.int64))
            return rec
        nl, ld = cross.shape
        mean = ctx.empty((nl + m, d))
        diag = ctx.empty(nl + m)
        ctx.copy(mean, mean_u_rows, nl * d * 8)
        ctx.copy(mean, mean_items, m * d * 8, dst_off=nl * d * 8)
        ctx.copy(diag, diag_u_rows, nl * 8)
        ctx.copy(diag, diag_items, m * 8, dst_off=nl * 8)
        ctx.mn_score_grid(np.float64, 1, nl, m, d, ld, mean, cross, diag, omega, mask=mask, dsum=rec.ptr, want=want,
                          best_val=rec.ptr + 8, best_idx=rec.ptr + 16)
        return rec


# ---------------------------------------------------------------------------- K2
class ShardedGrid(object):
    """One posterior state's candidate grid, user rows sharded, RESIDENT where it is consumed (SURVEY.md
    8(e) rows 1 and 4): rank g keeps its rows of the Sigma cross block (and of the candidate mask) in the
    engine's memory for the life of the object - the (n+m)^2 covariance never moves - and a selection step
    exchanges only what the other ranks cannot know: the item factors and the item entries of diag(Sigma),
    which are sharded by ITEM block (each rank owns and updates its block), in one grouped all-gather;
    cov_latents (d x d) is replicated.  After the sweep one (sum, value, flat index) record per rank is
    all-gathered and merged with the single-device tie rule."""

    def __init__(self, engine, comm, n, m, cross_rows, mask_rows=None):
        self.engine, self.comm, self.n, self.m = engine, comm, n, m
        self.lo, self.hi = block_range(n, comm.rank, comm.world)
        self.v_lo, self.v_hi = block_range(m, comm.rank, comm.world)
        nl = self.hi - self.lo
        if cross_rows.shape[0] != nl:
            raise ValueError('cross_rows must be this rank\'s %d rows' % nl)
        self.cross = engine.hold_grid(cross_rows)
        self.mask = None if mask_rows is None else engine.hold_grid(np.asarray(mask_rows).astype(np.uint8), np.uint8)

    def select(self, mean_u_rows, mean_v_block, diag_u_rows, diag_v_block, omega, want=1):
        """mean_u_rows (nl, d), diag_u_rows (nl,): this rank's user rows; mean_v_block / diag_v_block: its ITEM
        block (block_range(m) rows, host arrays).  -> (total, best value, (i, j)), identical on every rank."""
        eng, comm, n, m = self.engine, self.comm, self.n, self.m
        bs_v = block_size(m, comm.world)
        d = np.asarray(omega).shape[0]
        vb = np.zeros((bs_v, d))
        vb[:self.v_hi - self.v_lo] = mean_v_block
        db = np.zeros(bs_v)
        db[:self.v_hi - self.v_lo] = diag_v_block
        gv, gd = comm.all_gather_multi([eng.hold(vb), eng.hold(db)])          # the step's one exchange
        rec = eng.score_resident(self.cross, self.mask, eng.hold(mean_u_rows), eng.as_rows(gv), eng.hold(diag_u_rows),
                                 eng.as_rows(gd), eng.hold(omega), m, want)
        (recs,) = comm.all_gather_multi([rec])
        return _merge_select(eng.small_to_host(recs), self.lo_of_rank(), m, want)

    def lo_of_rank(self):
        return [block_range(self.n, r, self.comm.world)[0] for r in range(self.comm.world)]


def _merge_select(rows, los, m, want):
    rows = np.ascontiguousarray(rows, np.float64).reshape(len(los), -1)
    idx = np.ascontiguousarray(rows[:, 2]).view(np.int64)      # the index travels as its 8 bytes
    total = float(sum(float(r[0]) for r in rows))
    best_val, best_idx = 0., -1
    for v, i, lo in zip(rows[:, 1], idx, los):      # rank order = ascending row blocks; ties -> lowest flat index
        i = int(i)
        if i < 0:
            continue
        i = (i // m + lo) * m + i % m
        if best_idx < 0 or (v > best_val if want > 0 else v < best_val) or (v == best_val and i < best_idx):
            best_val, best_idx = float(v), i
    return total, best_val, (best_idx // m, best_idx % m) if best_idx >= 0 else None


def sharded_select(engine, comm, n, m, mean, cross, diag, omega, mask=None, want=1):
    """Arg-extremum and total of Var[U_i.V_j] over the (masked) n x m grid, user rows sharded.
    cross / mask may be the full (n, m) arrays or just this rank's row block.
    Returns (total, best_val, (i, j)) - identical on every rank."""
    lo, hi = block_range(n, comm.rank, comm.world)
    cr = cross if cross.shape[0] == hi - lo and comm.world > 1 else cross[lo:hi]
    mk = None if mask is None else (mask if mask.shape[0] == hi - lo and comm.world > 1 else mask[lo:hi])
    tot, bv, bi = engine.score_rows(mean[lo:hi], mean[n:], cr, diag[lo:hi], diag[n:], omega, mk, want)
    gidx = -1 if bi < 0 else (bi // m + lo) * m + bi % m
    rows = comm.all_gather_scalars([tot, bv, float(gidx)])
    total = float(rows[:, 0].sum())
    best_val, best_idx = 0., -1
    for _, v, i in rows:           # rank order = ascending row blocks; ties -> lowest flat index
        i = int(i)
        if i < 0:
            continue
        if best_idx < 0 or (v > best_val if want > 0 else v < best_val) or (v == best_val and i < best_idx):
            best_val, best_idx = float(v), i
    return total, best_val, (best_idx // m, best_idx % m) if best_idx >= 0 else None


# ---------------------------------------------------------------------------- K1
def _ll_from_partials(parts, sigma_sq, sigma_u_sq, sigma_v_sq):
    """log-likelihood from the (world, >= 3) per-rank partial sums, added in RANK ORDER: every rank computes
    the same bits, so every rank takes the same accept / reject decision without an all-reduce."""
    t = np.zeros(3)
    for row in np.asarray(parts, np.float64).reshape(len(parts), -1):
        t += row[:3]
    return -t[0] / (2. * sigma_sq) - t[1] / (2. * sigma_u_sq) - t[2] / (2. * sigma_v_sq)


def _ll(engine, comm, U, V, n, m, sigma_sq, sigma_u_sq, sigma_v_sq):
    u_lo, u_hi = block_range(n, comm.rank, comm.world)
    v_lo, v_hi = block_range(m, comm.rank, comm.world)
    (parts,) = comm.all_gather_multi([engine.ll_partial(U, V, u_lo, u_hi, v_lo, v_hi)])
    return _ll_from_partials(engine.small_to_host(parts), sigma_sq, sigma_u_sq, sigma_v_sq)


def sharded_log_likelihood(engine, comm, U, V, n, m, sigma_sq, sigma_u_sq, sigma_v_sq):
    bu, bv = block_size(n, comm.world) * comm.world, block_size(m, comm.world) * comm.world
    return _ll(engine, comm, engine.put(U, bu), engine.put(V, bv), n, m, sigma_sq, sigma_u_sq, sigma_v_sq)


def sharded_fit_lls(engine, comm, U, V, n, m, sigma_sq=1., sigma_u_sq=10., sigma_v_sq=10.,
                    learning_rate=1e-4, min_learning_rate=1e-10, stop_thresh=1e-2, max_steps=-1):
    """fit_lls (pmf_cy.pyx:257-291) with users and items row-sharded over the ranks, ONE collective per
    trial.  A trial point X is evaluated by every rank on its own rows: the three partial sums of LL(X) and -
    speculatively, as the single-GPU k_fit_iter does - the gradient at X.  Whether X is accepted is only known
    once all partial sums are in, so each rank forms its blocks of BOTH possible next trial points,
        A = X + 1.25 lr g(X)        (X accepted)            R = cur + 0.5 lr g(cur)       (X rejected),
    and a single grouped all-gather carries [partials | A_u | A_v | R_u | R_v]; every rank then adds the partials
    in rank order, decides, and continues with the gathered A or R.  The factors stay in the engine's arrays
    (device-resident for HipEngine) for the whole fit.  Returns (U, V, ll_trace)."""
    world = comm.world
    u_lo, u_hi = block_range(n, comm.rank, world)
    v_lo, v_hi = block_range(m, comm.rank, world)
    bs_u, bs_v = block_size(n, world), block_size(m, world)
    sig = (sigma_sq, sigma_u_sq, sigma_v_sq)

    def grads(A, B, slot):
        return (engine.grad_rows(0, u_lo, u_hi, A, B, slot), engine.grad_rows(1, v_lo, v_hi, A, B, slot))

    def blocks(A, B, g, rate):
        return [engine.trial_block(A, g[0], u_lo, u_hi, rate, bs_u), engine.trial_block(B, g[1], v_lo, v_hi, rate, bs_v)]

    cU, cV = engine.put(U, bs_u * world), engine.put(V, bs_v * world)
    lr, trace = learning_rate, []
    g_cur = grads(cU, cV, 0)
    # the starting LL and the first trial point (no decision pending yet): one exchange
    got = comm.all_gather_multi([engine.ll_partial(cU, cV, u_lo, u_hi, v_lo, v_hi)] + blocks(cU, cV, g_cur, lr))
    old_ll = _ll_from_partials(engine.small_to_host(got[0]), *sig)
    xU, xV = engine.as_rows(got[1]), engine.as_rows(got[2])
    slot = 1
    while max_steps < 0 or len(trace) < max_steps:
        part = engine.ll_partial(xU, xV, u_lo, u_hi, v_lo, v_hi)
        g_x = grads(xU, xV, slot)
        got = comm.all_gather_multi([part] + blocks(xU, xV, g_x, lr * 1.25) + blocks(cU, cV, g_cur, lr * .5))
        new_ll = _ll_from_partials(engine.small_to_host(got[0]), *sig)
        if new_ll > old_ll:
            cU, cV, g_cur, slot = xU, xV, g_x, slot ^ 1
            lr *= 1.25
            trace.append(new_ll)
            done = new_ll - old_ll < stop_thresh
            old_ll = new_ll
            if done:
                break
            xU, xV = engine.as_rows(got[1]), engine.as_rows(got[2])
        else:
            lr *= .5
            if lr < min_learning_rate:
                break
            xU, xV = engine.as_rows(got[3]), engine.as_rows(got[4])
    return engine.get(cU, n), engine.get(cV, m), np.array(trace)


# ---------------------------------------------------------------------------- K3
def sharded_gibbs_sweep(engine, comm, users, items, n, m, alpha_u, mu_u, alpha_v, mu_v, beta,
                        mean_sub, z_users, z_items):
    """One Gibbs sweep (bayes_pmf.py:280-296): users from the OLD items, all-gather, then
    items from the NEW users, all-gather.  z_* are the full (rows, d) host draws, identical on
    every rank (same seeded NumPy stream)."""
    world = comm.world
    u_lo, u_hi = block_range(n, comm.rank, world)
    v_lo, v_hi = block_range(m, comm.rank, world)
    bs_u, bs_v = block_size(n, world), block_size(m, world)
    h_items = engine.put(items, bs_v * world)
    hz_u, hz_v = engine.put(z_users), engine.put(z_items)
    new_users = comm.all_gather_rows(
        engine.gibbs_rows(0, u_lo, u_hi, h_items, alpha_u, mu_u, beta, mean_sub, hz_u, bs_u), n)
    new_items = comm.all_gather_rows(
        engine.gibbs_rows(1, v_lo, v_hi, new_users, alpha_v, mu_v, beta, mean_sub, hz_v, bs_v), m)
    return engine.get(new_users, n), engine.get(new_items, m)
