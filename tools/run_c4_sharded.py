#!/usr/bin/env python3
"""BASELINE.json configs[3] across GPUs: the lookahead criterion E[total variance] over a pool of
candidates of the 94 x 425 rank-5 problem, hypotheses sharded over the ranks (one process per GPU,
RCCL all-gather of the values).  Launch:
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/run_c4_sharded.py [candidates]
(torchrun is only the process launcher; APMF_BENCH_SHARE_GPU=1 rehearses N ranks on one GPU with the
exchange staged through the host).  Rank 0 prints JSON."""
import io
import json
import os
import random
import sys
import time
from contextlib import redirect_stdout

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from active_matrix_factorization_amd import _lib, sharding, mn_active_pmf as mn  # noqa: E402


def main():
    cands = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    rehearsal = bool(os.environ.get('APMF_BENCH_SHARE_GPU'))
    os.environ['APMF_DEVICE'] = str(local % max(_lib.device_count(), 1) if rehearsal else local)
    ctx = _lib.default_context()
    if world > 1:
        comm = sharding.Comm(rank, world, os.environ.get('MASTER_ADDR', '127.0.0.1'),
                             int(os.environ.get('MASTER_PORT', '29500')) + 1, ctx=ctx,
                             rccl=False if rehearsal else None)
    else:
        comm = sharding.Comm(0, 1, ctx=ctx, rccl=False)
    np.random.seed(0)
    random.seed(0)
    n, m, d, known = 94, 425, 5, 500
    u, v = np.random.normal(0, 1, (n, d)), np.random.normal(0, 1, (m, d))
    x = u @ v.T
    real = np.clip(np.round(3 + x / x.std() * 1.2), 1, 5)
    flat = np.random.choice(n * m, known, replace=False)
    ii, jj = flat // m, flat % m
    ratings = np.c_[ii, jj, real[ii, jj]].astype(float)
    a = mn.MNActivePMF(ratings, latent_d=d, rating_values=range(1, 6), discrete_expectations=True)
    a.do_fit()
    a.initialize_approx()
    a.fit_normal()                      # every rank computes the same model (deterministic)
    a.lookahead_comm = comm
    pool = sorted(a.unrated)[:cands]
    log = io.StringIO()
    comm.barrier()
    t = time.perf_counter()
    with redirect_stdout(log):
        vals = a._get_key_vals(pool, mn.KEY_FUNCS['total-variance'])
    comm.barrier()
    dt = time.perf_counter() - t
    if rank == 0:
        print(json.dumps({'world': world, 'candidates': cands, 'hypotheses': 5 * cands, 'seconds': dt,
                          'hypotheses_per_s': 5 * cands / dt, 'rccl': comm.has_rccl,
                          'argmin': list(map(int, pool[int(np.argmin(vals))])), 'checksum': float(np.sum(vals))}))
    comm.close()


if __name__ == '__main__':
    main()
