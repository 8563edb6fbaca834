#!/usr/bin/env python3
"""Latency of rocSOLVER's symmetric eigensolvers at the variational fit's (n+m) = 519: syevd vs the
Jacobi syevj, batch 1 / 8 / 64 (the batched projection of csrc/k6_batch.hip's slow path)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from active_matrix_factorization_amd import _lib  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 519
    ctx = _lib.Context(0)
    dense = _lib.load_dense()
    rs = C.CDLL('librocsolver.so.0', mode=C.RTLD_GLOBAL)
    handle = C.c_void_p()
    dense.apmf_dense_handle.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    assert dense.apmf_dense_handle(ctx._h, C.byref(handle)) == 0
    rng = np.random.RandomState(0)
    a = rng.normal(size=(N, N)) / np.sqrt(N)
    S = a @ a.T + 1e-3 * np.eye(N)
    for B in (1, 8, 64):
        mats = ctx.array(np.stack([S] * B))
        ptrs = ctx.array(np.array([mats.ptr + k * N * N * 8 for k in range(B)], np.uint64))
        D, E = ctx.empty((B, N)), ctx.empty((B, N))
        info = ctx.empty(B, np.int32)
        resid, nsw = ctx.empty(B), ctx.empty(B, np.int32)
        evect, lower, asc = 211, 122, 252     # rocblas_evect_original, rocblas_fill_lower, rocblas_esort_ascending
        for name in ('syevd', 'syevj'):
            ts = []
            for rep in range(4):
                mats.upload(np.stack([S] * B))
                ctx.sync()
                t = time.perf_counter()
                if name == 'syevd':
                    rc = rs.rocsolver_dsyevd_batched(handle, evect, lower, N, C.c_void_p(ptrs.ptr), N, C.c_void_p(D.ptr),
                                                     C.c_int64(N), C.c_void_p(E.ptr), C.c_int64(N),
                                                     C.c_void_p(info.ptr), B)
                else:
                    rc = rs.rocsolver_dsyevj_batched(handle, asc, evect, lower, N, C.c_void_p(ptrs.ptr), N,
                                                     C.c_double(0.), C.c_void_p(resid.ptr), 100,
                                                     C.c_void_p(nsw.ptr), C.c_void_p(D.ptr), C.c_int64(N),
                                                     C.c_void_p(info.ptr), B)
                ctx.sync()
                ts.append(time.perf_counter() - t)
            w = D.to_host()[0]
            err = np.abs(w - np.linalg.eigvalsh(S)).max()
            print('N=%d batch=%d %s: rc=%d %.2f ms (best of 4; first %.1f ms)  max |dw| %.2e  info %s' % (
                N, B, name, rc, min(ts) * 1e3, ts[0] * 1e3, err, info.to_host()[:3]))


if __name__ == '__main__':
    main()
