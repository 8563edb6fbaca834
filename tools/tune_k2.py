#!/usr/bin/env python3
"""Kernel-time probe for the K2 sweep (dev tool): times apmf_mn_score_grid_f32 at the bench
shape with different output sets, using the in-library HIP-event timing."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from active_matrix_factorization_amd import _lib

n, m, d = 943, 1682, 10
B = int(os.environ.get('B', 256))
ld = int(os.environ.get("LD", _lib.grid_ld(m)))
ctx = _lib.Context(0)
rs = np.random.RandomState(0)
mean = ctx.array(rs.normal(size=(B, n + m, d)), np.float32)
cross = ctx.array((rs.normal(size=(1, n, ld)) * .05).astype(np.float32).repeat(1, 0), np.float32) if B == 1 else None
if cross is None:
    cross = ctx.empty((B, n, ld), np.float32)
    one = (rs.normal(size=(n, ld)) * .05).astype(np.float32)
    for b in range(B):
        ctx._lib.apmf_h2d(ctx._h, cross.ptr + b * one.nbytes, (one + b * 1e-3).astype(np.float32).ctypes.data, one.nbytes)
diag = ctx.array(rs.rand(B, n + m) + .5, np.float32)
om = ctx.array(np.array([np.eye(d) * .3 + .01 for _ in range(B)]), np.float32)
mask = ctx.array((rs.rand(n, ld) < .94).astype(np.uint8), np.uint8)
var = ctx.empty((B, n, ld), np.float32)
pm = ctx.empty((B, n, ld), np.float32)
dsum, bv, bi = ctx.empty(B), ctx.empty(B, np.float32), ctx.empty(B, np.int64)

def run(name, reps=10, **kw):
    for _ in range(2):
        ctx.mn_score_grid(np.float32, B, n, m, d, ld, mean, cross, diag, om, **kw)
    ctx.sync()
    ctx.profile_begin(reps)
    for _ in range(reps):
        ctx.mn_score_grid(np.float32, B, n, m, d, ld, mean, cross, diag, om, **kw)
    ms = ctx.profile_end()
    import time as _t
    ctx.sync(); _t0 = _t.perf_counter()
    for _ in range(reps):
        ctx.mn_score_grid(np.float32, B, n, m, d, ld, mean, cross, diag, om, **kw)
    ctx.sync(); wall = (_t.perf_counter() - _t0) / reps * 1e3
    _t0 = _t.perf_counter()
    for _ in range(reps):
        ctx.mn_score_grid(np.float32, B, n, m, d, ld, mean, cross, diag, om, **kw)
    enq = (_t.perf_counter() - _t0) / reps * 1e3
    ctx.sync()
    name = '%s [wall %.3f ms/call, enqueue %.3f ms/call]' % (name, wall, enq)
    cand = B * n * m
    byt = 4 * cand * (1 + ('var' in kw) + ('pmean' in kw))
    print('%-80s min %.3f ms  med %.3f ms   %.0f GB/s (algorithmic, median)  %.3e cand/s' % (
        name, ms.min(), np.median(ms), byt / np.median(ms) / 1e6, cand / np.median(ms) * 1e3))

run('var', var=var)
run('var+sum', var=var, dsum=dsum)
run('var+sum+argmax', var=var, dsum=dsum, want=1, best_val=bv, best_idx=bi)
run('var+sum+argmax+mask (bench)', var=var, dsum=dsum, want=1, best_val=bv, best_idx=bi, mask=mask)
run('sum+argmax+mask (no write)', dsum=dsum, want=1, best_val=bv, best_idx=bi, mask=mask)
run('var+mean', var=var, pmean=pm)

# calibration: device-to-device copy of the same 1.6 GB in / 1.6 GB out (hipMemcpyAsync)
import time
for _ in range(2):
    ctx._lib.apmf_d2d(ctx._h, var.ptr, cross.ptr, cross.nbytes)
ctx.sync()
t0 = time.perf_counter()
R = 10
for _ in range(R):
    ctx._lib.apmf_d2d(ctx._h, var.ptr, cross.ptr, cross.nbytes)
ctx.sync()
dt = (time.perf_counter() - t0) / R
print('d2d copy of the same bytes: %.3f ms  %.0f GB/s (read+write)' % (dt * 1e3, 2 * cross.nbytes / dt / 1e9))
