#!/usr/bin/env python3
"""Kernel-time probe for the K2 sweep at larger shapes / ranks (env N, M, D, B), dev tool:
d <= 16 runs the streaming VALU kernel, d > 16 the matrix-core kernel."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from active_matrix_factorization_amd import _lib

n, m, d = int(os.environ.get('N', 32768)), int(os.environ.get('M', 8192)), int(os.environ.get('D', 64))
B = int(os.environ.get('B', 1))
ld = _lib.grid_ld(m)
ctx = _lib.Context(0)
rs = np.random.RandomState(0)
mean = ctx.array(rs.normal(size=(B, n + m, d)) * .3, np.float32)
cross = ctx.empty((B, n, ld), np.float32)
one = (rs.normal(size=(4096, ld)) * .05).astype(np.float32)
for off in range(0, B * n, 4096):
    k = min(4096, B * n - off)
    ctx._lib.apmf_h2d(ctx._h, cross.ptr + off * ld * 4, one.ctypes.data, k * ld * 4)
diag = ctx.array(rs.rand(B, n + m) + .5, np.float32)
om = ctx.array(np.array([np.eye(d) * .3 + .01 for _ in range(B)]), np.float32)
mask = ctx.array((rs.rand(n, ld) < .94).astype(np.uint8), np.uint8)
var = ctx.empty((B, n, ld), np.float32)
pm = ctx.empty((B, n, ld), np.float32)
dsum, bv, bi = ctx.empty(B), ctx.empty(B, np.float32), ctx.empty(B, np.int64)


def run(name, reps=5, **kw):
    for _ in range(2):
        ctx.mn_score_grid(np.float32, B, n, m, d, ld, mean, cross, diag, om, **kw)
    ctx.sync()
    ctx.profile_begin(reps)
    for _ in range(reps):
        ctx.mn_score_grid(np.float32, B, n, m, d, ld, mean, cross, diag, om, **kw)
    ms = ctx.profile_end()
    cand = B * n * m
    print('%-28s %8.3f ms  %7.1f Gcand/s  %6.2f TB/s(8B)  %6.1f TFLOP/s' %
          (name, ms.mean(), cand / ms.mean() / 1e6, 8 * cand / ms.mean() / 1e9, 2 * d * cand / ms.mean() / 1e9))


print('n=%d m=%d d=%d B=%d' % (n, m, d, B))
run('var', var=var)
run('var+sum+argmax', var=var, mask=mask, dsum=dsum, want=1, best_val=bv, best_idx=bi)
run('sum+argmax only', mask=mask, dsum=dsum, want=1, best_val=bv, best_idx=bi)
run('var+mean', var=var, pmean=pm)
