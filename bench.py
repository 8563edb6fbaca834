#!/usr/bin/env python3
"""bench.py - headline benchmark of the active-PMF hot path on MI355X.

Metric (BASELINE.json): candidate (i,j) variance scores/sec + PMF grad-steps/sec at
943 x 1682, rank 10.  One JSON line is printed by rank 0.  No PyTorch: one process per GPU
(torch.distributed.run is only the process launcher the driver uses; RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR / MASTER_PORT are read from the environment), rendezvous over a TCP
socket, collectives = RCCL behind the C-ABI (libapmf_comm.so) on the context's stream.

Workload (config.workload = "943x1682-d10-B256", BASELINE.json configs[1] shape):
  * scoring step : one sweep of the full 943 x 1682 candidate grid for B = 256 posterior
    states (what the one-step lookahead of mn_active_pmf.py:627-696 generates: one state per
    hypothesised rating), f32, scores written + fused per-state sum and arg-max, then - with
    more than one rank - the step's exchange: an RCCL all-gather of every rank's per-state
    results (sum f64, arg-max index i64, value f32: 20 B per state).  All inputs are resident
    in HBM before the timed region.  `value` = candidates scored per second, whole job.
  * grad step    : `grad_steps_per_call` accepted iterations of fit_lls (pmf_cy.pyx:257-291:
    gradient + trial log-likelihood + update) on the synthetic N = 100 000 ratings, f64.
--scaling weak (default): every rank owns B states and its own PMF refit (the lookahead's
hypotheses and the per-key refits are independent units: SURVEY.md 8(e)); the sweep has no
data-path collective.  --scaling strong: the B states are split over the ranks (B / N each) and
the PMF fit is ONE fit row-sharded over the ranks (sharding.sharded_fit_lls: per trial ONE grouped
device-to-device all-gather carrying the log-likelihood partial sums and the U / V row blocks of both
speculative next trial points).
The product is strict about its fabric (a communicator that cannot be set up raises).  bench.py alone sets it up with
fallback=True: if RCCL does not come up on EVERY rank the ranks agree on that over the rendezvous sockets, all drop to
the host-staged exchange together, and the line says so at the top level ("exchange_fallback": "<which ranks, why>")
as well as in config.exchange - a number measured over the fallback is labelled, never silent.  A rendezvous that
fails, or ranks stuck inside ncclCommInitRank while others failed, end in a time-out and a non-zero exit.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL's peer buffers): exported by the driver's shell already,
# set here too in case a launcher scrubbed the environment - before the HSA runtime comes up
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

N_USERS, N_ITEMS, LATENT_D, N_RATINGS = 943, 1682, 10, 100000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def build_id():
    """First 16 hex digits of the sha256 over the three product libraries: says which build a number (or a committed
    PMC pass, tools/pmc_*.py stamp the same id) belongs to."""
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, 'active_matrix_factorization_amd')
    for name in ('libapmf_hip.so', 'libapmf_dense.so', 'libapmf_comm.so'):
        try:
            with open(os.path.join(pkg, name), 'rb') as f:
                h.update(f.read())
        except OSError:
            h.update(b'missing:' + name.encode())
    return h.hexdigest()[:16]


def synth_ratings(seed):
    """U*, V* ~ N(0, 2^2), R = U* V*' + N(0, .25^2), N observed entries
    uniformly with >= 1 per row/column."""
    rs = np.random.RandomState(seed)
    n, m, d, N = N_USERS, N_ITEMS, LATENT_D, N_RATINGS
    Ut, Vt = rs.normal(0, 2, (n, d)), rs.normal(0, 2, (m, d))
    mask = np.zeros((n, m), bool)
    mask[np.arange(n), rs.randint(0, m, n)] = True
    mask[rs.randint(0, n, m), np.arange(m)] = True
    free = np.flatnonzero(~mask.ravel())
    mask.ravel()[rs.choice(free, N - int(mask.sum()), replace=False)] = True
    ii, jj = mask.nonzero()
    perm = rs.permutation(ii.size)
    ii, jj = ii[perm], jj[perm]
    vals = (Ut[ii] * Vt[jj]).sum(1) + rs.normal(0, .25, ii.size)
    return np.column_stack([ii, jj, vals]).astype(np.float64), mask


def synth_states(rs, B, ld, rank_r=32):
    """B posterior states MN(M, Sigma, Omega): Sigma = A A' + .1 I (A standard normal / sqrt(r),
    only its (n, m) cross block and diagonal are formed), Omega = Bm Bm' / d + .1 I."""
    n, m, d = N_USERS, N_ITEMS, LATENT_D
    mean = rs.normal(0, 1, (B, n + m, d)).astype(np.float32)
    cross = np.zeros((B, n, ld), np.float32)
    diag = np.empty((B, n + m), np.float32)
    omega = np.empty((B, d, d), np.float32)
    for b in range(B):
        A = (rs.standard_normal((n + m, rank_r)) / np.sqrt(rank_r)).astype(np.float32)
        cross[b, :, :m] = A[:n] @ A[n:].T
        diag[b] = (A * A).sum(1) + .1
        Bm = rs.standard_normal((d, d))
        omega[b] = Bm @ Bm.T / d + .1 * np.eye(d)
    return mean, cross, diag, omega


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--states', type=int, default=256, help='posterior states per sweep (B)')
    ap.add_argument('--grad-steps-per-call', type=int, default=25)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--workload', choices=('c2', 'c5'), default='c2',
                    help="c2: the metric's 943x1682-d10-B256 workload (default); c5: BASELINE.json configs[4] - ONE fixed "
                         "100000x20000 rank-64 problem, 5e7 ratings, row-sharded over the ranks (strong scaling)")
    ap.add_argument('--c5-rows', type=int, default=100000, help='user rows of the c5 workload (reduce for rehearsals)')
    ap.add_argument('--c5-ratings', type=int, default=50000000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-class-level', action='store_true')
    ap.add_argument('--no-grad-batch', action='store_true', help='skip the batched PMF re-fit leg')
    ap.add_argument('--no-f64', action='store_true', help='skip the float64 sweep legs (roofline_f64)')
    ap.add_argument('--no-gibbs', action='store_true', help='skip the configs[2] Gibbs / sample-statistics leg')
    ap.add_argument('--no-lookahead', action='store_true', help='skip the configs[3] batched-lookahead leg')
    ap.add_argument('--lookahead-candidates', type=int, default=100, help='pool size of the lookahead leg (x 5 rating values)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run '
                     '--nproc-per-node %d' % (args.gpus, args.gpus))
        args.gpus = world

    from active_matrix_factorization_amd import _lib, sharding
    rehearsal = bool(os.environ.get('APMF_BENCH_SHARE_GPU'))   # N > 1 on a one-GPU box only
    dev = local_rank % max(_lib.device_count(), 1) if rehearsal else local_rank
    ctx = _lib.Context(dev)
    # the communicator: a failure here ends the run (no silent single-rank numbers).  RCCL prints a
    # version banner when it initialises: keep stdout for the ONE JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal and os.environ.get('APMF_BENCH_SHARE_GPU') != 'rccl':
            comm = sharding.Comm(rank, world, os.environ['MASTER_ADDR'],
                                 int(os.environ.get('MASTER_PORT', '29500')) + 1, ctx=ctx, rccl=False)
        else:
            # (fallback: if RCCL does not come up on every rank, all ranks drop to the host-staged exchange together
            # and the line says so - a labelled number instead of no number from a first multi-GPU run)
            comm = sharding.comm_from_env(ctx, fallback=True)
    else:
        comm = sharding.Comm(0, 1, ctx=ctx, rccl=bool(os.environ.get('APMF_BENCH_FORCE_EXCHANGE')))
    comm.barrier()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    strong = args.scaling == 'strong'
    if args.workload == 'c5':
        out = bench_c5(args, ctx, comm, rank, world)
        if rank == 0:
            out['build_id'] = build_id()
            if getattr(comm, 'rccl_error', None):
                out['exchange_fallback'] = comm.rccl_error
            print(json.dumps(out))
        comm.barrier()
        comm.close()
        ctx.close()
        return

    def max_over_ranks(x):
        return float(comm.all_gather_scalars([x]).max())

    n, m, d = N_USERS, N_ITEMS, LATENT_D
    B_total = args.states
    B = max(1, B_total // world) if strong else B_total
    ld = _lib.grid_ld(m)
    K, W = args.steps, args.warmup

    # ------------------------------------------------------------------ inputs -> HBM
    rs = np.random.RandomState(1000 + rank)
    mean, cross, diag, omega = synth_states(rs, B, ld)
    ratings, rated_mask = synth_ratings(0)
    d_mean, d_cross, d_diag, d_omega = (ctx.array(x, np.float32) for x in (mean, cross, diag, omega))
    cand = np.zeros((n, ld), np.uint8)
    cand[:, :m] = ~rated_mask
    d_mask = ctx.array(cand, np.uint8)
    d_var = ctx.empty((B, n, ld), np.float32)
    # per-state results in ONE record [sum f64 (B) | arg-max index i64 (B) | value f32 (B)]: the
    # step's exchange is a single all-gather of it
    res = ctx.empty(20 * B, np.uint8)
    p_sum, p_idx, p_val = res.ptr, res.ptr + 8 * B, res.ptr + 16 * B
    res_all = ctx.empty((world, 20 * B), np.uint8)
    del cross
    exchange_note = 'single rank: no exchange'
    libc = _lib.load_comm() if comm.has_rccl else None
    if world > 1 or comm.has_rccl:
        exchange_note = ('%s all-gather of the per-state results (%d B per rank) after every sweep, on the '
                         'sweep\'s stream' % ('RCCL' if comm.has_rccl else ('FALLBACK host-staged (%s)' % comm.rccl_error) if comm.rccl_error else 'REHEARSAL host-staged', 20 * B))

    def step():
        ctx.mn_score_grid(np.float32, B, n, m, d, ld, d_mean, d_cross, d_diag, d_omega, mask=d_mask,
                          var=d_var, dsum=p_sum, want=1, best_val=p_val, best_idx=p_idx)
        if libc is not None:
            _lib._check(libc.apmf_comm_allgather(ctx._h, comm._nccl, res.ptr, res_all.ptr, res.nbytes))
        elif world > 1:
            res_all.upload(comm.all_gather_blocks(res.to_host()))

    # ------------------------------------------------------------------ region A: scoring
    for _ in range(W):
        step()
    comm.barrier()                      # drains this rank's stream, then waits for the others
    ctx.profile_begin(min(K, 1024))
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    comm.barrier()
    t_score = max_over_ranks(time.perf_counter() - t0)
    kern_ms = ctx.profile_end()
    cand_per_step = B * n * m
    scores_per_s = world * cand_per_step * K / t_score
    # sanity of the exchange: every rank's record arrived (sums are positive and finite)
    if world > 1 or comm.has_rccl:
        sums = np.frombuffer(res_all.to_host().tobytes(), np.uint8).reshape(world, 20 * B)[:, :8 * B]
        sums = sums.copy().view(np.float64)
        if not (np.isfinite(sums).all() and (sums > 0).all()):
            sys.exit('bench.py: the exchanged per-state results are not valid on rank %d' % rank)

    # ------------------------------------------------------------------ region B: PMF grad-steps
    G = args.grad_steps_per_call
    rs2 = np.random.RandomState(7)
    U0, V0 = rs2.random_sample((n, d)), rs2.random_sample((m, d))
    if strong and world > 1:
        eng = sharding.HipEngine(ctx, ratings, n, m, d)
        Kg = max(1, min(K, 40))
        sharding.sharded_fit_lls(eng, comm, U0, V0, n, m, stop_thresh=0., max_steps=2)
        comm.barrier()
        t0 = time.perf_counter()
        _, _, tr = sharding.sharded_fit_lls(eng, comm, U0, V0, n, m, stop_thresh=0., max_steps=Kg)
        comm.barrier()
        t_grad = max_over_ranks(time.perf_counter() - t0)
        done = int(tr.size)
        grad_steps_per_s = done / t_grad
        grad_note = ('ONE fit_lls row-sharded over %d ranks (one grouped device-to-device all-gather per trial: '
                     'log-likelihood partial sums + the U, V row blocks of both speculative next trial points), '
                     'N=100000 ratings' % world)
        per_gpu_steps = done
    else:
        rt = ctx.ratings(ratings, n, m)
        dU, dV = ctx.array(U0), ctx.array(V0)

        def fit_call(lr):
            _, steps, lr, _ = ctx.pmf_fit_lls(rt, dU, dV, d, 1., 10., 10., False, 0., lr, 1e-10, 0.0,
                                              max_steps=G, max_trace=0)
            return steps, lr

        lr = 1e-4
        for _ in range(min(W, 10)):
            _, lr = fit_call(lr)
        comm.barrier()
        t0 = time.perf_counter()
        done = 0
        for _ in range(min(K, 200)):
            s_, lr = fit_call(lr)
            done += s_
        comm.barrier()
        t_grad = max_over_ranks(time.perf_counter() - t0)
        grad_steps_per_s = world * done / t_grad
        grad_note = 'fit_lls accepted steps, N=100000 ratings, one independent refit per rank'
        per_gpu_steps = done

    # ------------------------------------------------------------------ results
    out = None
    if rank == 0:
        kern_ms = kern_ms[kern_ms > 0]
        avg_ms = float(kern_ms.mean()) if kern_ms.size else None
        algo_bytes = 8.0 * cand_per_step  # 4 B Sigma_ij read + 4 B score written
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms else None
        # HBM bytes per launch from the PMC counters: separate rocprofv3 passes of this same command (FETCH_SIZE and
        # WRITE_SIZE each in a pass of its own, the guide's gfx950 correction applied by tools/pmc_summary.py), so
        # they come from a committed file - quoted only for the workload they were collected on, with their
        # provenance beside them
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_k2_latest.json')
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if abs(float(rec.get('algorithmic_bytes_per_launch', 0)) - algo_bytes) < 1.:
                    traffic = rec.get('hbm_bytes_per_launch')
                    traffic_source = {'file': 'profiles/pmc_k2_latest.json', 'collected': rec.get('collected', 'round 3 (unstamped)'),
                                      'same_build': rec.get('build_id') == build_id() if rec.get('build_id') else None}
            except Exception:
                traffic, traffic_source = None, None
        # grad-step algorithmic bytes, SURVEY 8(d) f32/int32 COO basis
        gs_bytes = N_RATINGS * (12 + 16 * d) + N_RATINGS * (12 + 8 * d) + 16 * (n + m) * d
        out = {
            'metric': 'candidate (i,j) scores/sec + PMF grad-steps/sec, 943x1682 rank-10',
            'value': scores_per_s,
            'unit': 'candidate scores/s',
            'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': 1e3 * t_score / K,
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '943x1682-d10-B%d' % B_total, 'states_per_gpu': B,
                       'candidates_per_step_per_gpu': cand_per_step,
                       'outputs': 'scores written + per-state sum + arg-max over unrated',
                       'parallelism': ('%d states split over %d rank(s)' % (B_total, world)) if strong else
                                      ('%d states per rank x %d rank(s)' % (B, world)),
                       'exchange': exchange_note, 'rehearsal_shared_gpu': rehearsal},
            'roofline': {'bound': 'hbm', 'kernel': 'k_mn_grid<float,10,4>',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS if achieved else None,
                         'traffic': traffic, 'traffic_source': traffic_source,
                         'algorithmic_bytes_per_launch': algo_bytes,
                         'avg_kernel_ms': avg_ms, 'launches_timed': int(kern_ms.size)},
            'build_id': build_id(),
            'grad_steps': {'value': grad_steps_per_s, 'unit': 'PMF grad-steps/s', 'dtype': 'f64',
                           'ms_per_grad_step': 1e3 * t_grad / max(per_gpu_steps, 1),
                           'grad_steps_timed_per_gpu': per_gpu_steps,
                           'algorithmic_GBs': gs_bytes * grad_steps_per_s / (1 if strong else world) / 1e9,
                           'workload': grad_note},
        }
    if world == 1 and not args.no_grad_batch:
        print('bench.py: batched PMF re-fits ...', file=sys.stderr, flush=True)
        out['grad_steps']['batched'] = grad_steps_batched(ctx, ratings, rated_mask, U0, V0, min(K, 40))
    if world == 1 and not args.no_f64:
        # the product's precision on the same workload (every class computes in float64)
        print('bench.py: float64 sweep legs ...', file=sys.stderr, flush=True)
        out['roofline_f64'] = f64_sweeps(ctx, B, ld, d_mean, d_cross, d_diag, d_omega, d_mask, min(K, 200), min(W, 5))
    d_cross = d_var = None
    if world == 1 and not args.no_class_level:
        print('bench.py: kernel regions done; class-level selection ...', file=sys.stderr, flush=True)
        cl = class_level(ctx, ratings, rs)
        if out is not None:
            out['class_level'] = cl
    if world == 1 and not args.no_gibbs:
        print('bench.py: Gibbs sweep and sample statistics at configs[2] ...', file=sys.stderr, flush=True)
        out['gibbs'] = gibbs_leg(ctx, ratings)
    if world == 1 and not args.no_lookahead:
        # the product's real hot loop: the batched one-step lookahead (K6) at configs[3]'s shape
        print('bench.py: batched lookahead at configs[3] (about 15 s) ...', file=sys.stderr, flush=True)
        out['lookahead'] = lookahead_leg(args.lookahead_candidates)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            print('bench.py: CPU baseline (about 30 s) ...', file=sys.stderr, flush=True)
            out['cpu_baseline'] = cpu_baseline(mean[0], diag[0], omega[0], rs, ratings, U0, V0)
        if getattr(comm, 'rccl_error', None):
            out['exchange_fallback'] = comm.rccl_error      # top level: this number was measured host-staged
        print(json.dumps(out))
    comm.barrier()
    comm.close()
    ctx.close()


def bench_c5(args, ctx, comm, rank, world):
    """BASELINE.json configs[4] as ONE fixed problem, strong-scaled: 100 000 x 20 000, rank 64,
    5e7 ratings.  Rank g keeps its block of user rows of the Sigma cross block and of the candidate mask RESIDENT
    (f32, what the batched API streams) and the whole ratings container; a scoring step = every rank sweeps its rows
    (scores written + sum + arg-max over unrated) and ONE all-gather of the (sum, value, index) record per rank; the
    PMF half = ONE fit_lls row-sharded over the ranks (sharding.sharded_fit_lls: one grouped all-gather per trial).
    value = candidates of the WHOLE grid per second.  At N = 1 this is configs[4] on one GPU."""
    from active_matrix_factorization_amd import _lib, sharding
    n, m, d, N = args.c5_rows, 20000, 64, args.c5_ratings
    K, W = min(args.steps, 50), min(args.warmup, 5)
    lo, hi = sharding.block_range(n, rank, world)
    nl = hi - lo
    ld = _lib.grid_ld(m)
    rs = np.random.RandomState(5)                      # the same problem on every rank
    flat = np.unique(rs.randint(0, n * m, size=int(N * 1.04), dtype=np.int64))
    flat = flat[rs.permutation(flat.size)[:N]]
    N = flat.size
    ratings = np.empty((N, 3))
    ratings[:, 0], ratings[:, 1] = flat // m, flat % m
    ratings[:, 2] = rs.randint(1, 6, size=N)
    U, V = .35 * rs.standard_normal((n, d)), .35 * rs.standard_normal((m, d))
    A = rs.standard_normal((d, d)) / np.sqrt(d)
    omega = (A @ A.T).astype(np.float32)
    diag = (1. + rs.random_sample(n + m)).astype(np.float32)
    period = 1024
    blk = (.3 * rs.standard_normal((period, m))).astype(np.float32)
    # this rank's rows, resident: cross block, mask, [U rows ; V], [diag rows ; diag items]
    d_cross = ctx.zeros((max(nl, 1), ld), np.float32)
    for r0 in range(lo, hi, period):
        k = min(period - r0 % period, hi - r0)
        _lib._check(ctx._lib.apmf_h2d_2d(ctx._h, d_cross.ptr + (r0 - lo) * ld * 4, ld * 4,
                                         blk[r0 % period:].ctypes.data, m * 4, m * 4, k))
    mine = (flat // m >= lo) & (flat // m < hi)
    cand = np.ones((max(nl, 1), m), np.uint8)
    cand[flat[mine] // m - lo, flat[mine] % m] = 0
    d_mask = ctx.upload_grid(cand, np.uint8, ld=ld)
    del cand, flat
    d_mean = ctx.array(np.vstack([U[lo:hi], V]).astype(np.float32))
    d_diag = ctx.array(np.concatenate([diag[lo:hi], diag[n:]]))
    d_omega = ctx.array(omega)
    d_var = ctx.empty((max(nl, 1), ld), np.float32)
    rec = ctx.zeros(4)                                   # [sum f64 | value f32 .. | index i64]
    libc = _lib.load_comm() if comm.has_rccl else None

    def step():
        if nl:
            ctx.mn_score_grid(np.float32, 1, nl, m, d, ld, d_mean, d_cross, d_diag, d_omega, mask=d_mask, var=d_var,
                              dsum=rec.ptr, want=1, best_val=rec.ptr + 8, best_idx=rec.ptr + 16)
        return comm.all_gather_multi([rec])[0]

    for _ in range(W):
        step()
    comm.barrier()
    ctx.profile_begin(min(K, 1024))
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        got = step()
    comm.barrier()
    t_score = float(comm.all_gather_scalars([time.perf_counter() - t0]).max())
    kern_ms = ctx.profile_end()
    kern_ms = kern_ms[kern_ms > 0]
    sums = got.to_host()[:, 0]
    if not (np.isfinite(sums).all() and (sums > 0).sum() >= min(world, n)):
        sys.exit('bench.py: the exchanged per-rank results are not valid on rank %d' % rank)
    # ---- the PMF half: ONE fit, row-sharded
    eng = sharding.HipEngine(ctx, ratings, n, m, d)
    Kg = 6
    sharding.sharded_fit_lls(eng, comm, U, V, n, m, learning_rate=1e-6, min_learning_rate=1e-12, stop_thresh=0., max_steps=1)
    comm.barrier()
    t0 = time.perf_counter()
    _, _, tr = sharding.sharded_fit_lls(eng, comm, U, V, n, m, learning_rate=1e-6, min_learning_rate=1e-12, stop_thresh=0.,
                                        max_steps=Kg)
    comm.barrier()
    t_grad = float(comm.all_gather_scalars([time.perf_counter() - t0]).max())
    if rank != 0:
        return None
    avg_ms = float(kern_ms.mean()) if kern_ms.size else None
    algo = 8. * nl * m
    ach = algo / (avg_ms * 1e-3) / 1e9 if avg_ms else None
    gs_bytes = N * (12 + 16 * d) + N * (12 + 8 * d) + 16 * (n + m) * d
    return {'metric': 'candidate (i,j) scores/sec + PMF grad-steps/sec, configs[4] 100000x20000 rank-64 (strong scaling)',
            'value': n * m * K / t_score, 'unit': 'candidate scores/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': 1e3 * t_score / K, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%dx%d-d%d-N%d' % (n, m, d, N), 'rows_per_rank': nl,
                       'parallelism': 'user rows split over %d rank(s), cross block and mask resident per rank' % world,
                       'exchange': '%s all-gather of one 32-byte (sum, value, index) record per rank after every sweep'
                                   % ('RCCL' if comm.has_rccl else ('FALLBACK host-staged (%s)' % comm.rccl_error) if getattr(comm, 'rccl_error', None)
                                      else 'host-staged' if world > 1 else 'single rank: local copy')},
            'roofline': {'bound': 'hbm', 'kernel': 'k_mn_grid_mfma<64,...> (rank 0\'s rows)', 'achieved': ach, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS if ach else None, 'traffic': None,
                         'algorithmic_bytes_per_launch': algo, 'avg_kernel_ms': avg_ms, 'launches_timed': int(kern_ms.size)},
            'grad_steps': {'value': tr.size / t_grad, 'unit': 'PMF grad-steps/s', 'dtype': 'f64',
                           'ms_per_grad_step': 1e3 * t_grad / max(tr.size, 1), 'grad_steps_timed': int(tr.size),
                           'algorithmic_GBs': gs_bytes * tr.size / t_grad / 1e9,
                           'workload': 'ONE fit_lls row-sharded over %d rank(s): one grouped all-gather per trial' % world}}


def grad_steps_batched(ctx, ratings, mask, U0, V0, steps, H=64):
    """The PMF half of the metric as the package runs it when MANY re-fits are due at once (the MAP re-fits of a
    lookahead with refit_lookahead, mn_active_pmf.py:660-665; exp_variance(fit_first), bayes_pmf.py:457-468; the per-key
    models of compare()): H independent fit_lls searches - the 943 x 1682 rank-10 model on the N = 100 000 ratings, each
    with ONE extra rating - advanced together by apmf_pmf_batch_fit_lls (one chain of launches, per-hypothesis
    accept / reject on the device, bit-identical to H separate fits).  A single search is a dependent chain of 13 us
    launches (grad_steps.value: latency-bound, 0.14 of the HBM roof); H of them per launch stream at the roof."""
    n, m, d = N_USERS, N_ITEMS, LATENT_D
    rt = ctx.ratings(ratings, n, m)
    free_i, free_j = np.nonzero(~mask)
    pick = np.linspace(0, free_i.size - 1, H).astype(np.int64)
    ei, ej = free_i[pick].astype(np.int32), free_j[pick].astype(np.int32)
    ev = (1. + np.arange(H) % 5).astype(np.float64)

    def run(nsteps):
        dU = ctx.array(np.broadcast_to(U0, (H,) + U0.shape).copy())
        dV = ctx.array(np.broadcast_to(V0, (H,) + V0.shape).copy())
        ctx.sync()
        t0 = time.perf_counter()
        _, st, _ = ctx.pmf_batch_fit_lls(rt, dU, dV, d, H, ei, ej, ev, 1., 10., 10., False, None, 1e-4, 1e-10, 0.,
                                         max_steps=nsteps)
        ctx.sync()
        return time.perf_counter() - t0, int(st.sum())
    run(3)
    dt, done = run(max(steps, 5))
    gs_bytes = N_RATINGS * (12 + 16 * d) + N_RATINGS * (12 + 8 * d) + 16 * (n + m) * d
    rate = done / dt
    return {'value': rate, 'unit': 'PMF grad-steps/s', 'dtype': 'f64', 'concurrent_fits': H,
            'accepted_steps_timed': done, 'seconds': dt, 'us_per_grad_step': 1e6 * dt / max(done, 1),
            'algorithmic_GBs': gs_bytes * rate / 1e9, 'frac_of_hbm_peak': gs_bytes * rate / 1e9 / HBM_PEAK_GBS,
            'workload': '%d concurrent fit_lls searches (N=100000 ratings + one hypothesised rating each), '
                        'apmf_pmf_batch_fit_lls' % H}


F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X float64 matrix peak (vendor figure; the guide lists none for f64)


def f64_sweeps(ctx, B, ld, d_mean, d_cross, d_diag, d_omega, d_mask, K, W):
    """The SAME 943x1682-d10-B256 workload through apmf_mn_score_grid_f64 - the sweep every class of the
    package actually calls: (a) scores written + per-state sum + arg-max, 8 B read + 8 B written = 16 B per
    candidate (what get_key_evals / pick_query_point run), (b) reduction only (per-state sums, no grid written:
    8 B per candidate) - how the batched lookahead reduces its H hypothesis states (_total_variance) - and (c) the
    same with the masked arg-max fused.
    HIP events around the kernel, in-library, on the launch stream."""
    n, m, d = N_USERS, N_ITEMS, LATENT_D
    m64, c64, g64, o64 = (ctx.cast(x, np.float64) for x in (d_mean, d_cross, d_diag, d_omega))
    var = ctx.empty((B, n, ld), np.float64)
    dsum, bv, bi = ctx.empty(B), ctx.empty(B), ctx.empty(B, np.int64)
    out = {'workload': '943x1682-d10-B%d' % B, 'dtype': 'f64', 'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s'}
    # (the reduction leg is called exactly as the lookahead calls it - apmf_mn_batch_total_variance: per-state sums
    # only, no mask, no arg-extremum; 'reduce_argmax' adds the masked arg-max a selection over hypotheses would)
    for name, v, bpc, msk, want in (('written', var, 16., d_mask, 1), ('reduce_only', None, 8., None, 0),
                                    ('reduce_argmax', None, 8., d_mask, 1)):
        def step():
            ctx.mn_score_grid(np.float64, B, n, m, d, ld, m64, c64, g64, o64, mask=msk, var=v, dsum=dsum, want=want,
                              best_val=bv if want else None, best_idx=bi if want else None)
        for _ in range(W):
            step()
        ctx.sync()
        ctx.profile_begin(min(K, 1024))
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        ctx.sync()
        dt = time.perf_counter() - t0
        ms = ctx.profile_end()
        ms = ms[ms > 0]
        avg = float(ms.mean()) if ms.size else None
        algo = bpc * B * n * m
        ach = algo / (avg * 1e-3) / 1e9 if avg else None
        out[name] = {'achieved': ach, 'frac': ach / HBM_PEAK_GBS if ach else None, 'avg_kernel_ms': avg,
                     'algorithmic_bytes_per_launch': algo, 'bytes_per_candidate': bpc, 'launches_timed': int(ms.size),
                     'candidates_per_s': B * n * m * K / dt, 'ms_per_step': 1e3 * dt / K}
    sums = dsum.to_host()
    if not (np.isfinite(sums).all() and (sums > 0).all()):
        sys.exit('bench.py: the float64 sweep produced invalid sums')
    return out


def gibbs_leg(ctx, ratings, d=15, S=128):
    """BASELINE.json configs[2] (the Bayesian class at 943 x 1682, rank 15): K3 - one Gibbs sweep = the user
    half-sweep and the item half-sweep (bayes_pmf.py:266-300: per row alpha + beta F'F, its Cholesky factor, the
    draw) - and K4 - the statistics of the predicted matrix over S = 128 samples (bayes_pmf.py:361-440), the
    f64-MFMA kernel.  HIP events around each launch, in-library."""
    from active_matrix_factorization_amd import _lib
    n, m, N = N_USERS, N_ITEMS, ratings.shape[0]
    rs = np.random.RandomState(7)
    rt = ctx.ratings(ratings, n, m)
    V = ctx.array(rs.normal(size=(m, d)))
    al, mu = ctx.array(np.eye(d) * 2.), ctx.array(np.zeros(d))
    zu, zv = ctx.array(rs.normal(size=(n, d))), ctx.array(rs.normal(size=(m, d)))
    ou, ov = ctx.empty((n, d)), ctx.empty((m, d))

    def sweep():
        ctx.gibbs_half_sweep(rt, 0, V, d, al, mu, 2., 0., zu, ou)
        ctx.gibbs_half_sweep(rt, 1, ou, d, al, mu, 2., 0., zv, ov)
    for _ in range(3):
        sweep()
    ctx.sync()
    R = 50
    ctx.profile_begin(2 * R)
    t0 = time.perf_counter()
    for _ in range(R):
        sweep()
    ctx.sync()
    dt = (time.perf_counter() - t0) / R
    ms = ctx.profile_end()
    ms = ms[ms > 0]
    pair = float(ms.sum() / R) if ms.size else None
    # per row: its ratings (4 B index + 8 B value) and the other side's factor row per rating, z in, the row out
    algo = 2 * N * (12 + 8 * d) + 2 * (n + m) * d * 8
    # per rating a rank-1 update of the d x d precision (d^2 multiply-adds on the lower+upper square as the
    # reference's F'F forms it) and of the d-vector; per row the factorisation and two triangular solves
    flops = 2 * N * (2 * d * d + 2 * d) + (n + m) * (d ** 3 / 3 + 4 * d * d)
    out = {'workload': 'configs[2]: 943x1682 rank %d, N=%d' % (d, N), 'dtype': 'f64',
           'k3_sweep': {'kernel_ms_per_sweep': pair, 'ms_per_sweep_wall': dt * 1e3,
                        'row_updates_per_s': (n + m) / (pair * 1e-3) if pair else None,
                        'algorithmic_GBs': algo / (pair * 1e-3) / 1e9 if pair else None,
                        'gflops': flops / (pair * 1e-3) / 1e9 if pair else None,
                        'note': 'two launches of 943 / 1682 one-wave workgroups: latency-bound at this shape'}}
    Us, Vs = ctx.array(rs.normal(size=(S, n, d))), ctx.array(rs.normal(size=(S, m, d)))
    ld = _lib.grid_ld(m)
    gv, gm = ctx.empty((n, ld)), ctx.empty((n, ld))
    for _ in range(2):
        ctx.sample_stats(S, n, m, d, ld, Us, Vs, 0., 0., mean=gm, var=gv)
    ctx.sync()
    R = 20
    ctx.profile_begin(R)
    for _ in range(R):
        ctx.sample_stats(S, n, m, d, ld, Us, Vs, 0., 0., mean=gm, var=gv)
    ctx.sync()
    ms = ctx.profile_end()
    ms = ms[ms > 0]
    avg = float(ms.mean()) if ms.size else None
    dp = (d + 3) // 4 * 4
    mf = 2. * dp * S * n * m
    out['k4_sample_stats'] = {'samples': S, 'avg_kernel_ms': avg, 'bound': 'mfma', 'unit': 'TFLOP/s',
                              'peak': F64_MFMA_PEAK_TFLOPS,
                              'achieved': mf / (avg * 1e-3) / 1e12 if avg else None,
                              'frac': mf / (avg * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS if avg else None,
                              'flops_per_launch': mf,
                              'note': 'MFMA flops of the S x n x m inner products at d padded to %d; the mean / variance '
                                      'accumulation rides on the VALU beside them' % dp}
    if not np.isfinite(gv.to_host()[:, :m]).all():
        sys.exit('bench.py: the sample-statistics kernel produced non-finite values')
    rt.free()
    return out


def lookahead_leg(n_cand):
    """BASELINE.json configs[3]'s shape (94 x 425 rank 5, 500 known ratings, five rating values): the
    'total-variance' lookahead criterion (mn_active_pmf.py:627-696) over a pool of `n_cand` unrated entries =
    5 n_cand hypotheses, each a variational re-fit to convergence - one K6 batch (csrc/k6_batch.hip) - reduced by one
    float64 sweep with B = H.  Reports hypotheses/s, kernel launches per hypothesis and the matrix-core rate:
    float64 MFMA flops per second against the 78.6 TFLOP/s peak, from the instruction count of a committed
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 pass of this same leg when one is on file (profiles/pmc_lookahead_latest.json),
    and from the operation count of the factorisations otherwise."""
    import io
    import random
    from contextlib import redirect_stdout
    from active_matrix_factorization_amd import mn_active_pmf as mn, _lib
    state = np.random.get_state(), random.getstate()
    np.random.seed(0)
    random.seed(0)
    n, m, d, known = 94, 425, 5, 500
    u, v = np.random.normal(0, 1, (n, d)), np.random.normal(0, 1, (m, d))
    x = u @ v.T
    real = np.clip(np.round(3 + x / x.std() * 1.2), 1, 5)
    flat = np.random.choice(n * m, known, replace=False)
    ratings = np.c_[flat // m, flat % m, real[flat // m, flat % m]].astype(float)
    a = mn.MNActivePMF(ratings, latent_d=d, rating_values=range(1, 6), discrete_expectations=True)
    key = mn.KEY_FUNCS['total-variance']
    log = io.StringIO()
    with redirect_stdout(log):
        a.do_fit()
        a.initialize_approx()
        t0 = time.perf_counter()
        a.fit_normal()
        t_fit = time.perf_counter() - t0
        pool = sorted(a.unrated)[:n_cand]
        a._get_key_vals(pool[:1], key)                   # warm: code objects, workspaces
        l0 = _lib.batch_launch_count()
        t0 = time.perf_counter()
        vals = a._get_key_vals(pool, key)
        dt = time.perf_counter() - t0
        launches = _lib.batch_launch_count() - l0
    np.random.set_state(state[0])
    random.setstate(state[1])
    H = 5 * len(pool)
    N = n + m
    stats = dict(mn.LAST_BATCH_STATS) if getattr(mn, 'LAST_BATCH_STATS', None) else {}
    out = {'workload': 'configs[3] shape 94x425-d5, %d known, pool %d x 5 rating values' % (known, len(pool)),
           'hypotheses': H, 'seconds': dt, 'hypotheses_per_s': H / dt, 'launches': launches,
           'launches_per_hypothesis': launches / float(H), 'fit_stats': stats,
           'stream_groups': mn.lookahead_streams() if hasattr(mn, 'lookahead_streams') else 1,
           'base_fit_normal_s': t_fit, 'argmin': [int(q) for q in pool[int(np.argmin(vals))]],
           'full_grid_estimate_s': len(a.unrated) * 5 / (H / dt), 'dtype': 'f64',
           'matrix_dim': N, 'bytes_per_hypothesis_state': 5 * N * N * 8}
    # matrix-core work: two Cholesky factorisations per trial point (the second skipped when the eigenvalue bound
    # certifies the trial), N^3/3 flops each, and one inverse-from-factor per gradient (triangular inverse + W'W:
    # 2 N^3/3 flops).  A lower bound: the tile kernels also run the zero padding of 519 -> 576.
    if stats:
        fact = 2 * stats.get('trials', 0) - stats.get('cert_by_bound', 0)
        flops = (fact + 2 * stats.get('gradients', 0)) * (N ** 3 / 3.)
        out['mfma_model'] = {'flops': flops, 'tflops': flops / dt / 1e12, 'frac_of_peak': flops / dt / 1e12 / F64_MFMA_PEAK_TFLOPS,
                             'peak_tflops': F64_MFMA_PEAK_TFLOPS, 'note': 'operation count of the factorisations / inverses'}
    # Counters (not rates) of a separate rocprofv3 --pmc pass of this same leg, when one is on file for this
    # hypothesis count: the flops and bytes a run executes do not depend on the clock, so the RATES are formed with
    # THIS run's seconds; the file, when it was collected and whether it was this build are stated beside them
    pmc = os.path.join(ROOT, 'profiles', 'pmc_lookahead_latest.json')
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            if rec.get('hypotheses') == H:
                flops, hbm = float(rec['mfma_f64_flops']), float(rec['hbm_bytes'])
                out['pmc_counters'] = {
                    'source': 'profiles/pmc_lookahead_latest.json', 'collected': rec.get('collected', 'round 3 (unstamped)'),
                    'same_build': rec.get('build_id') == build_id() if rec.get('build_id') else None,
                    'mfma_f64_flops': flops, 'hbm_bytes': hbm, 'dispatches': rec.get('dispatches'),
                    'mfma_f64_tflops_at_this_runs_seconds': flops / dt / 1e12,
                    'mfma_f64_frac_of_peak': flops / dt / 1e12 / F64_MFMA_PEAK_TFLOPS, 'peak_tflops': F64_MFMA_PEAK_TFLOPS,
                    'hbm_GBs_at_this_runs_seconds': hbm / dt / 1e9, 'hbm_frac_of_8TBs': hbm / dt / 8e12,
                    'note': 'counter totals of a separate profiled pass (possibly an earlier build: same_build); '
                            'rates = those totals / the seconds of the run printed here'}
        except Exception:
            pass
    return out


def class_level(ctx, ratings, rs):
    """What the drop-in class delivers on the same grid (the kernel number is not the
    path): MNActivePMF.pick_query_point() over the full default pool - 1 486 126 unrated entries of
    943 x 1682, d = 10 - from the device-resident approximation: masked float64 sweep + fused
    arg-max + exact tie rule (mn_active_pmf.py:701-729, 886); and the float64 B = 1 sweep alone."""
    from active_matrix_factorization_amd import mn_active_pmf as mn, _lib
    n, m, d = N_USERS, N_ITEMS, LATENT_D
    a = mn.MNActivePMF(ratings, latent_d=d)
    a.users, a.items = rs.standard_normal((n, d)), rs.standard_normal((m, d))
    A = rs.standard_normal((n + m, 64)) / 8.
    a.mean = np.vstack((a.users, a.items))
    a.cov_useritems = A @ A.T + .1 * np.eye(n + m)
    Bm = rs.standard_normal((d, d))
    a.cov_latents = Bm @ Bm.T / d + .1 * np.eye(d)
    pool = len(a.unrated)
    first = a.pick_query_point()              # builds the device resident and the candidate mask
    ts = []
    for _ in range(7):
        t = time.perf_counter()
        pick = a.pick_query_point()
        ts.append(time.perf_counter() - t)
    assert pick == first
    t_pick = float(np.median(ts))
    # the float64 single-state sweep by itself (kernel events)
    res = a._resident()
    c = res.batch.ctx
    ld = _lib.grid_ld(m)
    var = c.empty((n, ld))
    res.batch.score_grid(0, 1, var=var, ld=ld)
    c.profile_begin(16)
    for _ in range(8):
        res.batch.score_grid(0, 1, var=var, ld=ld)
    ms = c.profile_end()
    ms = ms[ms > 0]
    a.get_key_evals()                         # first call loads the kernel variant
    te = []
    for _ in range(3):
        t = time.perf_counter()
        ev = a.get_key_evals()
        te.append(time.perf_counter() - t)
    t_evals = float(np.median(te))
    return {'pool': pool, 'pick_query_point_s': t_pick, 'cand_per_s': pool / t_pick,
            'pick': [int(pick[0]), int(pick[1])],
            'path': 'MNActivePMF.pick_query_point(): masked f64 sweep + fused arg-max from the device resident',
            'f64_B1_sweep_kernel_ms': float(ms.mean()) if ms.size else None,
            'f64_B1_sweep_GBs': (16. * n * m / (float(ms.mean()) * 1e-3) / 1e9) if ms.size else None,
            'get_key_evals_s': t_evals, 'get_key_evals_cand_per_s': pool / t_evals,
            'evals_finite': int(np.isfinite(ev).sum())}


def usable_cores():
    """Cores this process can actually run on: the affinity mask, capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(mean, diag, omega, rs, ratings, U0, V0):
    """The oracle (CPU restatement of the reference's per-candidate path: pred_variance ->
    approx_pred_mean_var -> exp_dotprod_sq) timed on a bounded sample of the same workload:
    whole rows of one 943 x 1682 grid, single thread.  kind = "port"."""
    from oracle import oracle as orc
    n, m, d = N_USERS, N_ITEMS, LATENT_D
    A = rs.standard_normal((n + m, 32)) / np.sqrt(32)
    R = A @ A.T + .1 * np.eye(n + m)
    mean64, om64 = mean.astype(np.float64), omega.astype(np.float64)
    t0 = time.perf_counter()
    orc.mn_score_rows(n, m, mean64, R, om64, 0, 16)
    per_row = (time.perf_counter() - t0) / 16
    # ~12 s of single-thread CPU work: whole grids of the same state, then a partial one
    total_rows = int(max(16, 12.0 / max(per_row, 1e-9)))
    grids, rem = divmod(total_rows, n)
    t0 = time.perf_counter()
    for _ in range(grids):
        orc.mn_score_rows(n, m, mean64, R, om64, 0, n)
    if rem:
        orc.mn_score_rows(n, m, mean64, R, om64, 0, rem)
    dt = time.perf_counter() - t0
    rows = total_rows
    # grad-step port: one gradient + one log-likelihood over the N ratings
    t1 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        orc.pmf_gradient(ratings, U0, V0)
        orc.pmf_log_likelihood(ratings, U0, V0)
    dt_g = (time.perf_counter() - t1) / reps
    # all-cores leg: the same rows dealt to every core this process may use
    # (affinity mask, capped by the cgroup CPU quota of the box share), ~8 s of wall clock
    cores = usable_cores()
    orc.mn_score_rows_threads(n, m, mean64, R, om64, 0, min(n, 4 * cores), cores)      # spin the team up
    reps_all = 0
    t2 = time.perf_counter()
    while True:
        orc.mn_score_rows_threads(n, m, mean64, R, om64, 0, n, cores)
        reps_all += 1
        dt_all = time.perf_counter() - t2
        if dt_all > 8.0:
            break
    ref = None
    try:
        ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_timings.json')))
    except Exception:
        pass
    return {'value': rows * m / dt, 'unit': 'candidate scores/s', 'cores': 1, 'kind': 'port',
            'all_cores': {'value': reps_all * n * m / dt_all, 'unit': 'candidate scores/s', 'cores': cores,
                          'sample': '%d grids of 943x1682 (%.1f s), OpenMP over rows' % (reps_all, dt_all)},
            'host_cores': os.cpu_count(),
            'reference_in_build_container': ref,
            'sample': '%d rows x %d items (%.2f grids of 943x1682, d=10; %.1f s), '
                      'oracle/apmf_oracle.c orc_mn_score_rows, f64' % (rows, m, rows / float(n), dt),
            'grad_steps_per_s': 1.0 / dt_g,
            'grad_sample': '%d x (gradient + log_likelihood) over N=100000, oracle, f64' % reps}


if __name__ == '__main__':
    main()
