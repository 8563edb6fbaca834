"""Multi-GPU sharding of the hot path: one process per GPU, row blocks of
users / items / candidate-grid rows per rank, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPUs, "gloo" in the CPU tests) for the exchange steps.

What shards, and what is exchanged:
  * candidate grid (K2): rank g sweeps the user rows of its block against all items.  No
    collective during the sweep; at the end an all-gather of one (value, flat index) pair
    and one partial sum per rank, merged with the single-device tie rule (lowest flat index).
  * PMF fit (K1): rank g owns a block of user rows and a block of item rows.  It computes
    grad_u for its users and grad_v for its items (each from the full ratings container,
    no reduction needed), updates its blocks of the trial point, and the trial U, V are
    all-gathered; the three sums of the log-likelihood are all-reduced, so every rank takes
    the same accept / reject decision of fit_lls.
  * Gibbs (K3): a half-sweep is split by row block, then the freshly sampled side is
    all-gathered before the other side's half-sweep (which needs the NEW values).
The functions here are engine-agnostic: `engine` is any object with the four row-block
methods of HipEngine (the product, over libapmf_hip.so).  PyTorch is plumbing only.
"""
import numpy as np


def block_size(n, world):
    return (n + world - 1) // world


def block_range(n, rank, world):
    """Rows [lo, hi) owned by `rank`: uniform blocks of ceil(n / world) rows (the last ones
    may be short or empty) - equal sizes keep the all-gather a plain one."""
    bs = block_size(n, world)
    lo = min(n, rank * bs)
    return lo, min(n, lo + bs)


class Comm(object):
    """torch.distributed wrapper for host NumPy blocks; world == 1 needs no process group."""

    def __init__(self, device=None):
        self.dist = None
        self.rank, self.world = 0, 1
        self.device = device
        # torch is only touched when the caller has already imported it (a process group can
        # only exist then).  Import order matters on ROCm: torch bundles its own rocBLAS /
        # rocSOLVER, so in a multi-GPU run `import torch` must come BEFORE the first Context
        # (libapmf_dense.so then binds to the copies torch loaded).
        import sys
        if 'torch' in sys.modules:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist
                self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def _tensor(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.device) if self.device is not None else t

    def all_gather_rows(self, block, n):
        """Row blocks (block_range layout) of an (n, ...) array -> the full array on every rank."""
        if self.world == 1:
            return np.ascontiguousarray(block)
        import torch
        bs = block_size(n, self.world)
        tail = block.shape[1:]
        padded = np.zeros((bs,) + tail, block.dtype)
        padded[:block.shape[0]] = block
        out = torch.empty((self.world * bs,) + tail, dtype=self._tensor(padded).dtype,
                          device=self.device if self.device is not None else 'cpu')
        self.dist.all_gather_into_tensor(out, self._tensor(padded))
        return out.cpu().numpy()[:n]

    def all_reduce_sum(self, a):
        a = np.asarray(a, np.float64)
        if self.world == 1:
            return a
        t = self._tensor(a.copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def all_gather_scalars(self, row):
        """One small float64 row per rank -> (world, len(row)) on every rank."""
        row = np.asarray(row, np.float64).reshape(1, -1)
        if self.world == 1:
            return row
        import torch
        out = torch.empty((self.world, row.shape[1]), dtype=torch.float64,
                          device=self.device if self.device is not None else 'cpu')
        self.dist.all_gather_into_tensor(out, self._tensor(row))
        return out.cpu().numpy()


class HipEngine(object):
    """Row-block operations on this rank's GPU through libapmf_hip.so (no fallback)."""

    def __init__(self, ctx, ratings, n, m, d, sigma_sq=1., sigma_u_sq=10., sigma_v_sq=10.,
                 subtract_mean=False, mean_rating=0.):
        self.ctx, self.n, self.m, self.d = ctx, n, m, d
        self.hyper = (float(sigma_sq), float(sigma_u_sq), float(sigma_v_sq), bool(subtract_mean),
                      float(mean_rating))
        self.rt = ctx.ratings(ratings, n, m) if ratings is not None else None

    # K1
    def grad_rows(self, side, lo, hi, U, V):
        ctx, d = self.ctx, self.d
        sig, su, sv, sub, mr = self.hyper
        rows = self.n if side == 0 else self.m
        g = ctx.empty((rows, d))
        ctx.pmf_gradient_rows(self.rt, side, lo, hi, ctx.array(U), ctx.array(V), d, sig, su, sv, sub,
                              mr, g)
        return g.to_host()[lo:hi]

    def ll_terms(self, U, V, u_lo, u_hi, v_lo, v_hi):
        _, _, _, sub, mr = self.hyper
        return self.ctx.pmf_ll_terms(self.rt, self.ctx.array(U), self.ctx.array(V), self.d, sub, mr,
                                     u_lo, u_hi, v_lo, v_hi)

    # K3
    def gibbs_rows(self, side, lo, hi, oth, alpha, mu, beta, mean_sub, z):
        ctx, d = self.ctx, self.d
        rows = self.n if side == 0 else self.m
        out = ctx.empty((rows, d))
        ctx.gibbs_half_sweep_rows(self.rt, side, lo, hi, ctx.array(oth), d, ctx.array(alpha),
                                  ctx.array(mu), float(beta), float(mean_sub), ctx.array(z), out)
        return out.to_host()[lo:hi]

    # K2: a block of user rows against all items, float64
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


# ---------------------------------------------------------------------------- K2
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
def sharded_log_likelihood(engine, comm, U, V, n, m, sigma_sq, sigma_u_sq, sigma_v_sq):
    u_lo, u_hi = block_range(n, comm.rank, comm.world)
    v_lo, v_hi = block_range(m, comm.rank, comm.world)
    t = comm.all_reduce_sum(engine.ll_terms(U, V, u_lo, u_hi, v_lo, v_hi))
    return -t[0] / (2. * sigma_sq) - t[1] / (2. * sigma_u_sq) - t[2] / (2. * sigma_v_sq)


def sharded_fit_lls(engine, comm, U, V, n, m, sigma_sq=1., sigma_u_sq=10., sigma_v_sq=10.,
                    learning_rate=1e-4, min_learning_rate=1e-10, stop_thresh=1e-2, max_steps=-1):
    """fit_lls (pmf_cy.pyx:257-291) with users and items row-sharded over the ranks.  Every
    rank runs the same control flow on the same all-reduced LL.  Returns (U, V, ll_trace)."""
    u_lo, u_hi = block_range(n, comm.rank, comm.world)
    v_lo, v_hi = block_range(m, comm.rank, comm.world)
    ll = lambda A, B: sharded_log_likelihood(engine, comm, A, B, n, m, sigma_sq, sigma_u_sq, sigma_v_sq)
    U, V = np.array(U, np.float64), np.array(V, np.float64)
    lr, old_ll, trace, converged = learning_rate, ll(U, V), [], False
    while not converged and (max_steps < 0 or len(trace) < max_steps):
        gU_blk = engine.grad_rows(0, u_lo, u_hi, U, V)
        gV_blk = engine.grad_rows(1, v_lo, v_hi, U, V)
        while True:
            # each rank updates its own blocks of the trial point; the exchange step
            nU = comm.all_gather_rows(U[u_lo:u_hi] + lr * gU_blk, n)
            nV = comm.all_gather_rows(V[v_lo:v_hi] + lr * gV_blk, m)
            new_ll = ll(nU, nV)
            if new_ll > old_ll:
                U, V = nU, nV
                lr *= 1.25
                if new_ll - old_ll < stop_thresh:
                    converged = True
                trace.append(new_ll)
                old_ll = new_ll
                break
            lr *= .5
            if lr < min_learning_rate:
                converged = True
                break
    return U, V, np.array(trace)


# ---------------------------------------------------------------------------- K3
def sharded_gibbs_sweep(engine, comm, users, items, n, m, alpha_u, mu_u, alpha_v, mu_v, beta,
                        mean_sub, z_users, z_items):
    """One Gibbs sweep (bayes_pmf.py:280-296): users from the OLD items, all-gather, then
    items from the NEW users, all-gather.  z_* are the full (rows, d) host draws, identical on
    every rank (same seeded NumPy stream)."""
    u_lo, u_hi = block_range(n, comm.rank, comm.world)
    v_lo, v_hi = block_range(m, comm.rank, comm.world)
    new_users = comm.all_gather_rows(
        engine.gibbs_rows(0, u_lo, u_hi, items, alpha_u, mu_u, beta, mean_sub, z_users), n)
    new_items = comm.all_gather_rows(
        engine.gibbs_rows(1, v_lo, v_hi, new_users, alpha_v, mu_v, beta, mean_sub, z_items), m)
    return new_users, new_items
