#!/usr/bin/env python3
"""bench.py - headline benchmark of the active-PMF hot path on MI355X.

Metric (BASELINE.json): candidate (i,j) variance scores/sec + PMF grad-steps/sec at
943 x 1682, rank 10.  One JSON line is printed by rank 0.

Workload (config.workload = "943x1682-d10-B256", BASELINE.json configs[1] shape):
  * scoring step : one sweep of the full 943 x 1682 candidate grid for B = 256 posterior
    states (what the one-step lookahead of mn_active_pmf.py:627-696 generates: one state per
    hypothesised rating), f32, scores written + fused per-state sum and arg-max.  All inputs
    are resident in HBM before the timed region.  `value` = candidates scored per second.
  * grad step    : `grad_steps_per_call` accepted iterations of fit_lls (pmf_cy.pyx:257-291:
    gradient + trial log-likelihood + update) on the synthetic N = 100 000 ratings, f64.
Multi-GPU (--gpus N, one process per GPU): posterior states and independent PMF refits are
sharded across ranks (each rank owns B states / its own refit); the sweep has no data-path
collective, so scaling is weak and `value` is the whole-job aggregate.  The path's one real
exchange step - after an outer iteration every rank publishes its row block of the refitted
factors U, V and of diag(Sigma) - is an RCCL all-gather (torch.distributed "nccl" group) issued
once per step on its own stream, overlapped with the next sweep.
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

N_USERS, N_ITEMS, LATENT_D, N_RATINGS = 943, 1682, 10, 100000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


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
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--states', type=int, default=256, help='posterior states per sweep (B)')
    ap.add_argument('--grad-steps-per-call', type=int, default=25)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run '
                     '--nproc-per-node %d' % (args.gpus, args.gpus))
        args.gpus = world

    dist = None
    if world > 1 or os.environ.get('APMF_BENCH_FORCE_EXCHANGE'):   # the latter: one-rank rehearsal
        # torch.distributed is plumbing only: rendezvous, barrier and the max over ranks
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    # exchange step over RCCL (xGMI): (n + m) x (d + 1) float64 factor/diagonal blocks
    exch = {'group': None, 'note': 'single rank: no exchange'}
    if dist is not None:
        # torch (with its bundled rocBLAS) is already imported: now the HIP libraries may load
        from active_matrix_factorization_amd import _lib as _early
        _early.load()
        try:
            import torch
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) %
                                  max(torch.cuda.device_count(), 1))
            grp = dist.new_group(backend='nccl')
            rows = -(-(N_USERS + N_ITEMS) // world)
            blk = torch.randn(rows, LATENT_D + 1, dtype=torch.float64, device='cuda')
            full = torch.empty(world * rows, LATENT_D + 1, dtype=torch.float64, device='cuda')
            dist.all_gather_into_tensor(full, blk, group=grp)
            torch.cuda.synchronize()
            exch = {'group': grp, 'blk': blk, 'full': full, 'torch': torch,
                    'note': 'RCCL all_gather_into_tensor of %d x %d f64 per rank per step, overlapped'
                            % (rows, LATENT_D + 1)}
        except Exception as e:     # never let the plumbing take the measurement down
            exch = {'group': None, 'note': 'exchange skipped: %s' % str(e).splitlines()[0][:160]}

    def exchange():
        if exch['group'] is not None:
            return dist.all_gather_into_tensor(exch['full'], exch['blk'], group=exch['group'],
                                               async_op=True)
        return None

    def device_sync():
        ctx.sync()
        if exch['group'] is not None:
            exch['torch'].cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    from active_matrix_factorization_amd import _lib
    dev = local_rank
    if os.environ.get('APMF_BENCH_SHARE_GPU'):   # rehearsal of N > 1 on a one-GPU box only
        dev = local_rank % max(_lib.device_count(), 1)
    ctx = _lib.Context(dev)
    n, m, d, B = N_USERS, N_ITEMS, LATENT_D, args.states
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
    d_sum, d_bv, d_bi = ctx.empty(B, np.float64), ctx.empty(B, np.float32), ctx.empty(B, np.int64)
    del cross

    def sweep():
        ctx.mn_score_grid(np.float32, B, n, m, d, ld, d_mean, d_cross, d_diag, d_omega, mask=d_mask,
                          var=d_var, dsum=d_sum, want=1, best_val=d_bv, best_idx=d_bi)

    # ------------------------------------------------------------------ region A: scoring
    for _ in range(W):
        sweep()
        exchange()
    device_sync()
    barrier()
    ctx.profile_begin(min(K, 1024))
    device_sync()
    t0 = time.perf_counter()
    pending = None
    for _ in range(K):
        sweep()                 # enqueued on the ctx stream
        if pending is not None:
            pending.wait()
        pending = exchange()    # RCCL stream: runs beside the sweep just enqueued
    if pending is not None:
        pending.wait()
    device_sync()
    barrier()
    t_score = max_over_ranks(time.perf_counter() - t0)
    kern_ms = ctx.profile_end()
    cand_per_step = B * n * m
    scores_per_s = world * cand_per_step * K / t_score

    # ------------------------------------------------------------------ region B: PMF grad-steps
    G = args.grad_steps_per_call
    rt = ctx.ratings(ratings, n, m)
    rs2 = np.random.RandomState(7)
    U0, V0 = rs2.random_sample((n, d)), rs2.random_sample((m, d))
    dU, dV = ctx.array(U0), ctx.array(V0)

    def fit_call(lr):
        _, steps, lr, _ = ctx.pmf_fit_lls(rt, dU, dV, d, 1., 10., 10., False, 0., lr, 1e-10, 0.0,
                                          max_steps=G, max_trace=0)
        return steps, lr

    lr = 1e-4
    for _ in range(W):
        _, lr = fit_call(lr)
    ctx.sync()
    barrier()
    t0 = time.perf_counter()
    done = 0
    for _ in range(K):
        s, lr = fit_call(lr)
        done += s
    ctx.sync()
    barrier()
    t_grad = max_over_ranks(time.perf_counter() - t0)
    grad_steps_per_s = world * done / t_grad

    # ------------------------------------------------------------------ results
    if rank == 0:
        kern_ms = kern_ms[kern_ms > 0]
        avg_ms = float(kern_ms.mean()) if kern_ms.size else None
        algo_bytes = 8.0 * cand_per_step  # 4 B Sigma_ij read + 4 B score written
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms else None
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_k2_latest.json')
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                # PMC passes are separate rocprofv3 runs of this same command; only quote them
                # for the workload they were measured on
                if rec.get('workload') == '943x1682-d10-B%d' % B:
                    traffic = rec.get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        # grad-step algorithmic bytes, SURVEY 8(d) f32/int32 COO basis
        gs_bytes = N_RATINGS * (12 + 16 * d) + N_RATINGS * (12 + 8 * d) + 16 * (n + m) * d
        out = {
            'metric': 'candidate (i,j) scores/sec + PMF grad-steps/sec, 943x1682 rank-10',
            'value': scores_per_s,
            'unit': 'candidate scores/s',
            'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': 1e3 * t_score / K,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '943x1682-d10-B%d' % B, 'states_per_gpu': B,
                       'candidates_per_step_per_gpu': cand_per_step,
                       'outputs': 'scores written + per-state sum + arg-max over unrated',
                       'parallelism': 'states sharded over %d rank(s), no collective inside the sweep' % world,
                       'exchange': exch['note']},
            'roofline': {'bound': 'hbm', 'kernel': 'k_mn_grid<float,10,4>',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS if achieved else None,
                         'traffic': traffic, 'algorithmic_bytes_per_launch': algo_bytes,
                         'avg_kernel_ms': avg_ms, 'launches_timed': int(kern_ms.size)},
            'grad_steps': {'value': grad_steps_per_s, 'unit': 'PMF grad-steps/s', 'dtype': 'f64',
                           'ms_per_grad_step': 1e3 * t_grad / max(done, 1) * world,
                           'grad_steps_timed_per_gpu': done,
                           'algorithmic_GBs': gs_bytes * grad_steps_per_s / world / 1e9,
                           'workload': 'fit_lls accepted steps, N=100000 ratings, one independent '
                                       'refit per rank'},
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(mean[0], diag[0], omega[0], rs, ratings, U0, V0)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


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
    return {'value': rows * m / dt, 'unit': 'candidate scores/s', 'cores': 1, 'kind': 'port',
            'sample': '%d rows x %d items (%.2f grids of 943x1682, d=10; %.1f s), '
                      'oracle/apmf_oracle.c orc_mn_score_rows, f64' % (rows, m, rows / float(n), dt),
            'grad_steps_per_s': 1.0 / dt_g,
            'grad_sample': '%d x (gradient + log_likelihood) over N=100000, oracle, f64' % reps}


if __name__ == '__main__':
    main()
