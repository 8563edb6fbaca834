"""ctypes binding of libapmf_hip.so (include/apmf.h) - the only way the Python classes of
this package reach the GPU.  No PyTorch, no CPU fallback: if the shared library or a
gfx950 device is missing, every call fails loudly with ApmfError.

The binding is deliberately thin: `Context` owns one device + stream, `DeviceArray` owns one
hipMalloc'd buffer, and each method maps 1:1 onto a C entry point.
"""
import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libapmf_hip.so')

GRID_ALIGN = 64
MAX_D = 64

OK, EINVAL, EHIP, ENOMEM, ENOTPD, ENODEV = 0, -1, -2, -3, -4, -5


class ApmfError(RuntimeError):
    """A libapmf_hip call failed (status < 0)."""

    def __init__(self, status, message):
        super().__init__('libapmf_hip status %d: %s' % (status, message))
        self.status = status


_vp = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_dbl = C.c_double
_sz = C.c_size_t

_SIGS = {
    'apmf_version': (_i, []),
    'apmf_last_error': (C.c_char_p, []),
    'apmf_device_count': (_i, [C.POINTER(_i)]),
    'apmf_ctx_create': (_i, [_i, C.POINTER(_vp)]),
    'apmf_ctx_destroy': (_i, [_vp]),
    'apmf_ctx_set_stream': (_i, [_vp, _vp]),
    'apmf_ctx_sync': (_i, [_vp]),
    'apmf_ctx_trim': (_i, [_vp]),
    'apmf_grid_ld': (_i, [_i]),
    'apmf_profile_begin': (_i, [_vp, _i]),
    'apmf_profile_end': (_i, [_vp, _vp, _i, C.POINTER(_i)]),
    'apmf_malloc': (_i, [_vp, _sz, C.POINTER(_vp)]),
    'apmf_free': (_i, [_vp, _vp]),
    'apmf_memset': (_i, [_vp, _vp, _i, _sz]),
    'apmf_h2d': (_i, [_vp, _vp, _vp, _sz]),
    'apmf_d2h': (_i, [_vp, _vp, _vp, _sz]),
    'apmf_d2d': (_i, [_vp, _vp, _vp, _sz]),
    'apmf_h2d_2d': (_i, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    'apmf_d2h_2d': (_i, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    'apmf_d2d_2d': (_i, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    'apmf_axpy_f64': (_i, [_vp, _vp, _vp, _dbl, _vp, _sz]),
    'apmf_cast_f64_f32': (_i, [_vp, _vp, _vp, _sz]),
    'apmf_cast_f32_f64': (_i, [_vp, _vp, _vp, _sz]),
    'apmf_ratings_create': (_i, [_vp, _vp, _i64, _i, _i, C.POINTER(_vp)]),
    'apmf_ratings_create_dense': (_i, [_vp, _vp, _i, _vp, _i, _i, _i, C.POINTER(_vp)]),
    'apmf_ratings_destroy': (_i, [_vp, _vp]),
    'apmf_ratings_count': (_i, [_vp, C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i)]),
    'apmf_pmf_gradient': (_i, [_vp, _vp, _vp, _vp, _i, _dbl, _dbl, _dbl, _i, _dbl, _vp, _vp]),
    'apmf_pmf_gradient_rows': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _dbl, _dbl, _dbl, _i, _dbl, _vp]),
    'apmf_pmf_ll_terms': (_i, [_vp, _vp, _vp, _vp, _i, _i, _dbl, _i, _i, _i, _i, C.POINTER(_dbl)]),
    'apmf_pmf_ll_terms_dev': (_i, [_vp, _vp, _vp, _vp, _i, _i, _dbl, _i, _i, _i, _i, _vp]),
    'apmf_pmf_log_likelihood': (_i, [_vp, _vp, _vp, _vp, _i, _dbl, _dbl, _dbl, _i, _dbl,
                                     C.POINTER(_dbl), C.POINTER(_dbl)]),
    'apmf_pmf_fit_lls': (_i, [_vp, _vp, _vp, _vp, _i, _dbl, _dbl, _dbl, _i, _dbl, C.POINTER(_dbl),
                              C.POINTER(_dbl), _dbl, _dbl, _i64, _vp, _i64, C.POINTER(_i64),
                              C.POINTER(_i)]),
    'apmf_pmf_batch_fit_lls': (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _i, _vp, _dbl, _dbl,
                                    _dbl, _i64, _vp, _vp, _vp]),
    'apmf_pmf_minibatch_epoch': (_i, [_vp, _vp, _i64, _i, _i, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                      _i, _dbl, _dbl, _dbl, _i, _dbl]),
    'apmf_pmf_predicted_matrix': (_i, [_vp, _vp, _vp, _i, _i, _i, _dbl, _vp, _i]),
    'apmf_mn_score_grid_f32': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _i, _vp, _vp]),
    'apmf_mn_score_grid_f64': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _i, _vp, _vp]),
    'apmf_mn_score_pool_f64': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'apmf_grid_collect_near_f32': (_i, [_vp, _i, _i, _i, _vp, _vp, _i, C.c_float, C.c_float, _vp, _vp,
                                        _i, C.POINTER(_i)]),
    'apmf_grid_collect_eq_f64': (_i, [_vp, _i, _i, _i, _vp, _vp, _dbl, _vp, _vp, _i, C.POINTER(_i)]),
    'apmf_full_score_grid_f64': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'apmf_gibbs_half_sweep': (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _dbl, _dbl, _vp, _vp]),
    'apmf_gibbs_half_sweep_rows': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _dbl, _dbl, _vp, _vp]),
    'apmf_gibbs_half_sweep_batched': (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _dbl, _vp, _vp, _vp, _vp, _vp, _vp]),
    'apmf_sample_stats_f64': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _dbl, _dbl, _vp, _vp, _vp]),
    'apmf_gibbs_sample_hyperparam': (_i, [_vp, _vp, _i, _i, _vp, _dbl, _dbl, _vp, _vp, C.c_uint64, C.c_uint64, _vp,
                                          _vp, _vp]),
    'apmf_gibbs_sample_hyperparam_batched': (_i, [_vp, _i, _vp, _i, _i, _vp, _dbl, _dbl, _vp, _vp, _i, _vp, _vp, _vp]),
    'apmf_fill_normal_f64': (_i, [_vp, _vp, _i64, C.c_uint64, C.c_uint64]),
    'apmf_grid_gather_f64': (_i, [_vp, _vp, _i, _vp, _vp, _i64, _vp]),
    'apmf_pool_argext_f64': (_i, [_vp, _vp, _i64, _i, C.POINTER(_i64), C.POINTER(_dbl)]),
}

_DENSE_SIGS = {
    'apmf_dense_project_psd': (_i, [_vp, _vp, _i, _dbl, C.POINTER(_i)]),
    'apmf_dense_inv_logdet': (_i, [_vp, _vp, _i, _vp, C.POINTER(_dbl)]),
    'apmf_mn_kl_value': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _dbl, _dbl, _dbl, C.POINTER(_dbl)]),
    'apmf_mn_kl_gradient': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _dbl, _dbl, _dbl, _vp, _vp, _vp]),
    'apmf_full_kl_gradient': (_i, [_vp, _vp, _i, _vp, _vp, _dbl, _dbl, _dbl, _vp, _vp]),
    'apmf_full_kl_value': (_i, [_vp, _vp, _i, _vp, _vp, _dbl, _dbl, _dbl, C.POINTER(_dbl)]),
    'apmf_full_fit_normal': (_i, [_vp, _vp, _i, _vp, _vp, _dbl, _dbl, _dbl, _dbl, C.POINTER(_dbl),
                                  C.POINTER(_dbl), _i64, _vp, _i64, C.POINTER(_i64), C.POINTER(_i)]),
    'apmf_mn_fit_normal': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, C.POINTER(_dbl),
                                C.POINTER(_dbl), _i64, _vp, _i64, C.POINTER(_i64), C.POINTER(_i)]),
}
_DENSE_SIGS.update({
    'apmf_mn_batch_create': (_i, [_vp, _vp, _i, _i, _i64, C.POINTER(_vp)]),
    'apmf_mn_batch_destroy': (_i, [_vp, _vp]),
    'apmf_mn_batch_size': (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'apmf_mn_batch_fit_stats': (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64),
                                     C.POINTER(_i64), C.POINTER(_i64)]),
    'apmf_mn_batch_launch_count': (_i, [C.POINTER(_i64)]),
    'apmf_mn_batch_set_state': (_i, [_vp, _vp, _vp, _vp, _vp, _i]),
    'apmf_mn_batch_set_extra': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'apmf_mn_batch_get_state': (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    'apmf_mn_batch_fit': (_i, [_vp, _vp, _dbl, _dbl, _dbl, _dbl, _vp, _vp, _i64, _vp, _vp, _vp]),
    'apmf_mn_batch_kl': (_i, [_vp, _vp, _dbl, _dbl, _dbl, _vp]),
    'apmf_mn_batch_gradient': (_i, [_vp, _vp, _i, _dbl, _dbl, _dbl, _vp, _vp, _vp]),
    'apmf_mn_batch_total_variance': (_i, [_vp, _vp, _vp]),
    'apmf_mn_batch_entropy': (_i, [_vp, _vp, _vp]),
    'apmf_mn_batch_score_pool': (_i, [_vp, _vp, _i, _vp, _vp, _i64, _vp, _vp]),
    'apmf_mn_batch_score_grid': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
})
_DENSE_SIGS.update({
    'apmf_full_batch_create': (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    'apmf_full_batch_destroy': (_i, [_vp, _vp]),
    'apmf_full_batch_set_state': (_i, [_vp, _vp, _vp, _vp, _i]),
    'apmf_full_batch_set_extra': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'apmf_full_batch_get_state': (_i, [_vp, _vp, _i, _vp, _vp]),
    'apmf_full_batch_fit': (_i, [_vp, _vp, _dbl, _dbl, _dbl, _dbl, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    'apmf_full_batch_fit_stats': (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    'apmf_full_batch_kl': (_i, [_vp, _vp, _dbl, _dbl, _dbl, _vp]),
    'apmf_full_batch_logdet': (_i, [_vp, _vp, _vp]),
    'apmf_full_batch_total_variance': (_i, [_vp, _vp, _vp]),
})
DENSE_LIB_PATH = os.path.join(_HERE, 'libapmf_dense.so')

_COMM_SIGS = {
    'apmf_comm_unique_id': (_i, [_vp]),
    'apmf_comm_create': (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    'apmf_comm_destroy': (_i, [_vp]),
    'apmf_comm_rank': (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    'apmf_comm_allgather': (_i, [_vp, _vp, _vp, _vp, _sz]),
    'apmf_comm_allgather_multi': (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    'apmf_comm_version': (_i, [C.POINTER(_i)]),
    'apmf_comm_allreduce_f64': (_i, [_vp, _vp, _vp, _vp, _sz]),
    'apmf_comm_barrier': (_i, [_vp, _vp]),
}
COMM_LIB_PATH = os.path.join(_HERE, 'libapmf_comm.so')

_lib = None
_dense = None
_comm = None
_lock = threading.Lock()


def exported_symbols():
    """Names include/apmf.h declares (used by the CPU-side ABI test)."""
    return sorted(_SIGS) + sorted(_DENSE_SIGS) + sorted(_COMM_SIGS)


def load_comm():
    """libapmf_comm.so (RCCL underneath): opened on first use by the multi-GPU path only."""
    global _comm
    load()
    with _lock:
        if _comm is None:
            if not os.path.exists(COMM_LIB_PATH):
                raise ApmfError(ENODEV, 'HIP extension %s is not built; there is no CPU fallback'
                                % COMM_LIB_PATH)
            lib = C.CDLL(COMM_LIB_PATH, mode=C.RTLD_GLOBAL)
            for name, (res, args) in _COMM_SIGS.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _comm = lib
    return _comm


def load_dense():
    """libapmf_dense.so (rocSOLVER / rocBLAS underneath); opened by load()."""
    load()
    if _dense is None:
        raise ApmfError(ENODEV, 'HIP extension %s is not built; there is no CPU fallback'
                        % DENSE_LIB_PATH)
    return _dense


def load():
    """dlopen libapmf_hip.so once.  Loading needs no GPU; creating a Context does."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ApmfError(ENODEV, 'HIP extension %s is not built (run `python -c "import '
                                '__graft_entry__ as g; g.build()"` or `make -C %s/csrc`); there '
                                'is no CPU fallback' % (LIB_PATH, _HERE))
            # The lookahead criteria drive one stream per worker thread (4 by default).  With the HIP
            # runtime's default of 4 hardware queues two of those streams end up sharing a queue and
            # 4 threads run no faster than 2 (3.6 fits/s); with 8 queues they overlap like separate
            # processes do (6.5 fits/s, tools/probe_scaling.py).  Read by the runtime when it
            # initialises, so it only takes effect if nothing has touched HIP yet; never overrides
            # a value the user set.
            os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
            lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)   # libapmf_dense.so resolves apmf_ws etc. here
            for name, (res, args) in _SIGS.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
            # libapmf_dense.so pulls in rocSOLVER / rocBLAS, whose (very large) device-code
            # bundles must be registered with the HIP runtime BEFORE the first kernel launch of
            # the process: registered afterwards they are unpacked eagerly, which takes minutes.
            # So it is opened here, together with the main library, never later.
            try:
                _open_dense()
            except OSError:
                pass     # dense_* calls will raise; the main library is unaffected
    return _lib


class _LateTorchGuard(object):
    """libapmf_dense.so links the system rocBLAS / rocSOLVER.  PyTorch bundles its own copies
    under the same SONAMEs; importing torch AFTER the system copies are loaded crashes inside
    torch's static initialisers (a segmentation fault, not an exception).  The product itself never
    imports torch; this import hook turns the one remaining way to hit the clash - user code
    importing torch later in the same process - into an ImportError that says what to do.
    (torch imported first is fine: the dense library then binds to the copies torch loaded.)"""

    _MESSAGE = (
        "import torch after active_matrix_factorization_amd has loaded libapmf_dense.so: PyTorch's "
        "bundled rocBLAS / rocSOLVER clash with the system copies already loaded and the process "
        "would crash.  Import torch BEFORE the first use of this package (or run the two in "
        "separate processes); APMF_ALLOW_LATE_TORCH=1 disables this check.")

    def find_spec(self, name, path=None, target=None):
        if name != 'torch' or 'torch' in sys.modules or os.environ.get('APMF_ALLOW_LATE_TORCH') == '1':
            return None
        # availability probes (importlib.util.find_spec('torch')) get the real answer - None, or a spec -
        # and only an actual import, which runs the spec's loader, raises
        spec = None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, 'find_spec'):
                continue
            spec = finder.find_spec(name, path, target)
            if spec is not None:
                break
        if spec is None:
            return None
        spec.loader = _RefuseLoader(self._MESSAGE)
        return spec


class _RefuseLoader(object):
    def __init__(self, message):
        self.message = message

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        raise ImportError(self.message)


def _open_dense():
    global _dense
    if _dense is None and os.path.exists(DENSE_LIB_PATH):
        if 'torch' not in sys.modules and not any(isinstance(f, _LateTorchGuard) for f in sys.meta_path):
            sys.meta_path.insert(0, _LateTorchGuard())
        lib = C.CDLL(DENSE_LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _DENSE_SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _dense = lib


def _check(status):
    if status != OK:
        msg = load().apmf_last_error().decode('utf-8', 'replace')
        if status == EINVAL:
            raise ValueError(msg)
        if status == ENOMEM:
            raise MemoryError(msg)
        if status == ENOTPD:
            raise np.linalg.LinAlgError(msg)
        raise ApmfError(status, msg)


def device_count():
    n = _i(0)
    _check(load().apmf_device_count(C.byref(n)))
    return n.value


def grid_ld(m):
    return (int(m) + GRID_ALIGN - 1) // GRID_ALIGN * GRID_ALIGN


def _ptr(x):
    """device pointer of a DeviceArray / int / None"""
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return x.ptr
    return int(x)


class DeviceArray(object):
    """One device allocation with a NumPy-like shape/dtype tag."""

    def __init__(self, ctx, shape, dtype, zero=False):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = _vp()
        _check(ctx._lib.apmf_malloc(ctx._h, max(self.nbytes, 1), C.byref(p)))
        self.ptr = p.value
        if zero:
            self.zero()

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def zero(self):
        _check(self.ctx._lib.apmf_memset(self.ctx._h, self.ptr, 0, self.nbytes))
        return self

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError('upload size mismatch: %r vs %r' % (host.shape, self.shape))
        _check(self.ctx._lib.apmf_h2d(self.ctx._h, self.ptr, host.ctypes.data, self.nbytes))
        return self

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        _check(self.ctx._lib.apmf_d2h(self.ctx._h, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib, h = self.ctx._lib, self.ctx._h
            if h:
                lib.apmf_free(h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Ratings(object):
    """Device-resident CSR-by-user + CSR-by-item of an (N, 3) ratings array (apmf_ratings)."""

    def __init__(self, ctx, ratings, n, m):
        ratings = np.ascontiguousarray(ratings, dtype=np.float64)
        if ratings.ndim != 2 or ratings.shape[1] != 3:
            raise TypeError('invalid rating tuple length')
        self.ctx = ctx
        self.N, self.n, self.m = ratings.shape[0], int(n), int(m)
        h = _vp()
        _check(ctx._lib.apmf_ratings_create(ctx._h, ratings.ctypes.data, self.N, self.n, self.m,
                                            C.byref(h)))
        self._h = h.value

    @classmethod
    def from_dense(cls, ctx, real, known):
        """Container built on the device from a dense (n, m) matrix and its boolean `known` mask,
        entries in row-major order (what choose_training.py's make_ratings() produces)."""
        real = np.ascontiguousarray(real, dtype=np.float64)
        known = np.ascontiguousarray(known, dtype=bool)
        if real.ndim != 2 or known.shape != real.shape:
            raise TypeError('real and known must be (n, m) arrays of the same shape')
        self = cls.__new__(cls)
        self.ctx = ctx
        self.n, self.m = real.shape
        d_real, d_known = ctx.array(real), ctx.array(known.view(np.uint8), np.uint8)
        h = _vp()
        _check(ctx._lib.apmf_ratings_create_dense(ctx._h, d_real.ptr, self.m, d_known.ptr, self.m, self.n,
                                                  self.m, C.byref(h)))
        self._h = h.value
        cnt = _i64(0)
        _check(ctx._lib.apmf_ratings_count(self._h, C.byref(cnt), None, None))
        self.N = cnt.value
        return self

    def free(self):
        if getattr(self, '_h', None):
            if self.ctx._h:
                self.ctx._lib.apmf_ratings_destroy(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MNBatch(object):
    """H matrix-normal approximations resident on the device (apmf_mn_batch): the batched
    one-step lookahead, and - with H = 1 - the device-resident state of one MNActivePMF."""

    def __init__(self, ctx, rt, d, H, trace_cap=0):
        self.ctx, self.rt = ctx, rt          # rt is kept alive: the batch borrows it
        self.H, self.d, self.n, self.m = int(H), int(d), rt.n, rt.m
        self.trace_cap = int(trace_cap)
        self._dense = load_dense()
        h = _vp()
        _check(self._dense.apmf_mn_batch_create(ctx._h, rt._h, self.d, self.H, self.trace_cap, C.byref(h)))
        self._h = h.value

    def free(self):
        if getattr(self, '_h', None):
            if self.ctx._h:
                self._dense.apmf_mn_batch_destroy(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def set_state(self, mean=None, cov_useritems=None, cov_latents=None, per_hypothesis=0):
        _check(self._dense.apmf_mn_batch_set_state(self.ctx._h, self._h, _ptr(mean), _ptr(cov_useritems),
                                                   _ptr(cov_latents), int(per_hypothesis)))

    def set_extra(self, hi=None, hj=None, hv=None):
        if hi is None:
            _check(self._dense.apmf_mn_batch_set_extra(self.ctx._h, self._h, None, None, None))
            return
        hi = np.ascontiguousarray(hi, np.int32)
        hj = np.ascontiguousarray(hj, np.int32)
        hv = np.ascontiguousarray(hv, np.float64)
        if not (hi.size == hj.size == hv.size == self.H):
            raise ValueError('one extra rating per hypothesis expected')
        _check(self._dense.apmf_mn_batch_set_extra(self.ctx._h, self._h, hi.ctypes.data, hj.ctypes.data,
                                                   hv.ctypes.data))

    def get_state(self, h, mean=None, cov_useritems=None, cov_latents=None):
        _check(self._dense.apmf_mn_batch_get_state(self.ctx._h, self._h, int(h), _ptr(mean),
                                                   _ptr(cov_useritems), _ptr(cov_latents)))

    def fit(self, sig, sig_u, sig_v, min_eig, lr, old_kl=None, max_steps=-1):
        """-> (lr (H), old_kl (H), steps (H), converged (H), kl_trace (H, trace_cap))"""
        lr_io = np.ascontiguousarray(np.broadcast_to(np.asarray(lr, np.float64), (self.H,))).copy()
        kl_io = np.full(self.H, np.nan) if old_kl is None else \
            np.ascontiguousarray(np.broadcast_to(np.asarray(old_kl, np.float64), (self.H,))).copy()
        steps = np.zeros(self.H, np.int64)
        conv = np.zeros(self.H, np.int32)
        trace = np.full((self.H, max(self.trace_cap, 1)), np.nan)
        _check(self._dense.apmf_mn_batch_fit(self.ctx._h, self._h, float(sig), float(sig_u), float(sig_v),
                                             float(min_eig), lr_io.ctypes.data, kl_io.ctypes.data,
                                             int(max_steps), trace.ctypes.data if self.trace_cap else None,
                                             steps.ctypes.data, conv.ctypes.data))
        return lr_io, kl_io, steps, conv.astype(bool), trace

    def fit_stats(self):
        """Counters of the last fit(): search iterations, trial points, projections finished through the lowest
        eigenpairs / through the full eigendecomposition, trial certificates settled by the eigenvalue bound."""
        v = [C.c_int64(0) for _ in range(6)]
        _check(self._dense.apmf_mn_batch_fit_stats(self._h, *[C.byref(x) for x in v]))
        return dict(zip(('iterations', 'trials', 'proj_low', 'proj_full', 'cert_by_bound', 'gradients'),
                        (int(x.value) for x in v)))

    def kl(self, sig, sig_u, sig_v):
        out = np.empty(self.H)
        _check(self._dense.apmf_mn_batch_kl(self.ctx._h, self._h, float(sig), float(sig_u), float(sig_v),
                                            out.ctypes.data))
        return out

    def gradient(self, h, sig, sig_u, sig_v, g_mean, g_cov_useritems, g_cov_latents):
        _check(self._dense.apmf_mn_batch_gradient(self.ctx._h, self._h, int(h), float(sig), float(sig_u),
                                                  float(sig_v), _ptr(g_mean), _ptr(g_cov_useritems),
                                                  _ptr(g_cov_latents)))

    def total_variance(self):
        out = np.empty(self.H)
        _check(self._dense.apmf_mn_batch_total_variance(self.ctx._h, self._h, out.ctypes.data))
        return out

    def entropy(self):
        out = np.empty(self.H)
        _check(self._dense.apmf_mn_batch_entropy(self.ctx._h, self._h, out.ctypes.data))
        return out

    def score_pool(self, h, pi, pj, P, pmean, var):
        _check(self._dense.apmf_mn_batch_score_pool(self.ctx._h, self._h, int(h), _ptr(pi), _ptr(pj), int(P),
                                                    _ptr(pmean), _ptr(var)))

    def score_grid(self, h0=0, B=None, f32=False, mask=None, var=None, pmean=None, dsum=None, want=0,
                   best_val=None, best_idx=None, ld=None):
        B = self.H - h0 if B is None else B
        ld = grid_ld(self.m) if ld is None else ld
        _check(self._dense.apmf_mn_batch_score_grid(self.ctx._h, self._h, int(h0), int(B), int(bool(f32)), ld,
                                                    _ptr(mask), _ptr(var), _ptr(pmean), _ptr(dsum), int(want),
                                                    _ptr(best_val), _ptr(best_idx)))


class FullBatch(object):
    """H full-covariance approximations N(mean, cov) resident on the device (apmf_full_batch): the batched
    one-step lookahead of ActivePMF and - with H = 1 - the engine behind its fit_normal."""

    def __init__(self, ctx, rt, d, H):
        self.ctx, self.rt = ctx, rt          # rt is kept alive: the batch borrows it
        self.H, self.d, self.n, self.m = int(H), int(d), rt.n, rt.m
        self.K = (rt.n + rt.m) * int(d)
        self._dense = load_dense()
        h = _vp()
        _check(self._dense.apmf_full_batch_create(ctx._h, rt._h, self.d, self.H, C.byref(h)))
        self._h = h.value

    def free(self):
        if getattr(self, '_h', None):
            if self.ctx._h:
                self._dense.apmf_full_batch_destroy(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def set_state(self, mean, cov, per_hypothesis=0):
        _check(self._dense.apmf_full_batch_set_state(self.ctx._h, self._h, _ptr(mean), _ptr(cov), int(per_hypothesis)))

    def set_extra(self, hi=None, hj=None, hv=None):
        if hi is None:
            _check(self._dense.apmf_full_batch_set_extra(self.ctx._h, self._h, None, None, None))
            return
        hi = np.ascontiguousarray(hi, np.int32)
        hj = np.ascontiguousarray(hj, np.int32)
        hv = np.ascontiguousarray(hv, np.float64)
        if not (hi.size == hj.size == hv.size == self.H):
            raise ValueError('one extra rating per hypothesis expected')
        _check(self._dense.apmf_full_batch_set_extra(self.ctx._h, self._h, hi.ctypes.data, hj.ctypes.data,
                                                     hv.ctypes.data))

    def get_state(self, h, mean=None, cov=None):
        _check(self._dense.apmf_full_batch_get_state(self.ctx._h, self._h, int(h), _ptr(mean), _ptr(cov)))

    def fit(self, sig, sig_u, sig_v, min_eig, lr, old_kl=None, max_steps=-1, trace_cap=0):
        """-> (lr (H), old_kl (H), steps (H), converged (H), kl_trace (H, trace_cap))"""
        lr_io = np.ascontiguousarray(np.broadcast_to(np.asarray(lr, np.float64), (self.H,))).copy()
        kl_io = np.full(self.H, np.nan) if old_kl is None else \
            np.ascontiguousarray(np.broadcast_to(np.asarray(old_kl, np.float64), (self.H,))).copy()
        steps = np.zeros(self.H, np.int64)
        conv = np.zeros(self.H, np.int32)
        trace = np.full((self.H, max(int(trace_cap), 1)), np.nan)
        _check(self._dense.apmf_full_batch_fit(self.ctx._h, self._h, float(sig), float(sig_u), float(sig_v),
                                               float(min_eig), lr_io.ctypes.data, kl_io.ctypes.data, int(max_steps),
                                               trace.ctypes.data if trace_cap else None, int(trace_cap),
                                               steps.ctypes.data, conv.ctypes.data))
        return lr_io, kl_io, steps, conv.astype(bool), trace

    def fit_stats(self):
        v = [C.c_int64(0) for _ in range(3)]
        _check(self._dense.apmf_full_batch_fit_stats(self._h, *[C.byref(x) for x in v]))
        return dict(zip(('iterations', 'trials', 'projected'), (int(x.value) for x in v)))

    def kl(self, sig, sig_u, sig_v):
        out = np.empty(self.H)
        _check(self._dense.apmf_full_batch_kl(self.ctx._h, self._h, float(sig), float(sig_u), float(sig_v),
                                              out.ctypes.data))
        return out

    def logdet(self):
        out = np.empty(self.H)
        _check(self._dense.apmf_full_batch_logdet(self.ctx._h, self._h, out.ctypes.data))
        return out

    def total_variance(self):
        out = np.empty(self.H)
        _check(self._dense.apmf_full_batch_total_variance(self.ctx._h, self._h, out.ctypes.data))
        return out


def batch_launch_count():
    """Kernel launches enqueued by the batch engine (k6_batch.hip) in this process so far."""
    v = _i64(0)
    _check(load_dense().apmf_mn_batch_launch_count(C.byref(v)))
    return int(v.value)


class Context(object):
    """One device + one stream (apmf_ctx).  Not thread-safe; never create before fork()."""

    def __init__(self, device=0):
        self._lib = load()
        h = _vp()
        _check(self._lib.apmf_ctx_create(int(device), C.byref(h)))
        self._h = h.value
        self.device = int(device)
        self._pid = os.getpid()

    def close(self):
        if self._h and os.getpid() == self._pid:
            self._lib.apmf_ctx_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ------------------------------------------------------------------
    def sync(self):
        _check(self._lib.apmf_ctx_sync(self._h))

    def trim(self):
        """Return the context's cache of freed device blocks to the driver (after a sync)."""
        _check(self._lib.apmf_ctx_trim(self._h))

    def set_stream(self, hip_stream):
        _check(self._lib.apmf_ctx_set_stream(self._h, hip_stream))

    def profile_begin(self, max_records=256):
        _check(self._lib.apmf_profile_begin(self._h, max_records))

    def profile_end(self, cap=4096):
        ms = np.zeros(cap, np.float32)
        cnt = _i(0)
        _check(self._lib.apmf_profile_end(self._h, ms.ctypes.data, cap, C.byref(cnt)))
        return ms[:min(cnt.value, cap)].astype(np.float64)

    def empty(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype, zero=True)

    def array(self, host, dtype=None):
        host = np.asarray(host)
        dtype = host.dtype if dtype is None else np.dtype(dtype)
        return DeviceArray(self, host.shape, dtype).upload(host)

    def upload_grid(self, host_2d, dtype=np.float64, ld=None):
        """(rows, m) host matrix (any row stride) -> zero-padded (rows, ld) device grid."""
        host_2d = np.asarray(host_2d, dtype=dtype)
        rows, m = host_2d.shape
        ld = grid_ld(m) if ld is None else ld
        g = self.zeros((rows, ld), dtype)
        if host_2d.strides[1] != host_2d.itemsize:
            host_2d = np.ascontiguousarray(host_2d)
        isz = host_2d.itemsize
        _check(self._lib.apmf_h2d_2d(self._h, g.ptr, ld * isz, host_2d.ctypes.data,
                                     host_2d.strides[0], m * isz, rows))
        return g

    def copy(self, dst, src, nbytes, dst_off=0, src_off=0):
        """device-to-device copy of nbytes (byte offsets into the two arrays), on the stream"""
        if dst_off < 0 or src_off < 0 or dst_off + nbytes > dst.nbytes or src_off + nbytes > src.nbytes:
            raise ValueError('copy outside the device arrays')
        _check(self._lib.apmf_d2d(self._h, dst.ptr + int(dst_off), src.ptr + int(src_off), int(nbytes)))

    def write_bytes(self, darr, offset, host):
        """host bytes -> darr at byte `offset` (small patches of a resident array)"""
        host = np.ascontiguousarray(host)
        if offset < 0 or offset + host.nbytes > darr.nbytes:
            raise ValueError('patch outside the device array')
        _check(self._lib.apmf_h2d(self._h, darr.ptr + int(offset), host.ctypes.data, host.nbytes))

    def download_grid(self, grid, m):
        """(rows, ld) device grid -> contiguous (rows, m) host array."""
        rows, ld = grid.shape[-2], grid.shape[-1]
        lead = int(np.prod(grid.shape[:-2], dtype=np.int64)) if len(grid.shape) > 2 else 1
        out = np.empty(grid.shape[:-2] + (rows, m), grid.dtype)
        isz = grid.dtype.itemsize
        _check(self._lib.apmf_d2h_2d(self._h, out.ctypes.data, m * isz, grid.ptr, ld * isz, m * isz,
                                     rows * lead))
        return out

    def axpy(self, out, x, alpha, y, count, out_off=0, x_off=0, y_off=0):
        """out[out_off:][:count] = x[x_off:][:count] + alpha * y[y_off:][:count] (float64 elements)"""
        _check(self._lib.apmf_axpy_f64(self._h, out.ptr + 8 * int(out_off), x.ptr + 8 * int(x_off),
                                       float(alpha), y.ptr + 8 * int(y_off), int(count)))

    def cast(self, src, dtype):
        dst = DeviceArray(self, src.shape, dtype)
        if src.dtype == np.float64 and dst.dtype == np.float32:
            _check(self._lib.apmf_cast_f64_f32(self._h, dst.ptr, src.ptr, src.size))
        elif src.dtype == np.float32 and dst.dtype == np.float64:
            _check(self._lib.apmf_cast_f32_f64(self._h, dst.ptr, src.ptr, src.size))
        else:
            raise ValueError('unsupported cast %s -> %s' % (src.dtype, dst.dtype))
        return dst

    def ratings(self, ratings, n, m):
        return Ratings(self, ratings, n, m)

    def mn_batch(self, rt, d, H, trace_cap=0):
        return MNBatch(self, rt, d, H, trace_cap)

    def full_batch(self, rt, d, H):
        return FullBatch(self, rt, d, H)

    # -- K1 ---------------------------------------------------------------------------
    def pmf_gradient(self, rt, dU, dV, d, sig, sig_u, sig_v, subtract_mean, mean_rating, dgU, dgV):
        _check(self._lib.apmf_pmf_gradient(self._h, rt._h, _ptr(dU), _ptr(dV), d, sig, sig_u, sig_v,
                                           int(bool(subtract_mean)), mean_rating, _ptr(dgU), _ptr(dgV)))

    def pmf_gradient_rows(self, rt, side, lo, hi, dU, dV, d, sig, sig_u, sig_v, subtract_mean,
                          mean_rating, dg):
        _check(self._lib.apmf_pmf_gradient_rows(self._h, rt._h, side, lo, hi, _ptr(dU), _ptr(dV), d, sig,
                                                sig_u, sig_v, int(bool(subtract_mean)), mean_rating,
                                                _ptr(dg)))

    def pmf_ll_terms(self, rt, dU, dV, d, subtract_mean, mean_rating, u_lo, u_hi, v_lo, v_hi):
        t = (_dbl * 3)()
        _check(self._lib.apmf_pmf_ll_terms(self._h, rt._h, _ptr(dU), _ptr(dV), d,
                                           int(bool(subtract_mean)), mean_rating, u_lo, u_hi, v_lo, v_hi,
                                           t))
        return np.array(t[:], np.float64)

    def pmf_ll_terms_dev(self, rt, dU, dV, d, subtract_mean, mean_rating, u_lo, u_hi, v_lo, v_hi, out3):
        """the three sums of pmf_ll_terms left on the device (out3: pointer / DeviceArray), no host sync"""
        _check(self._lib.apmf_pmf_ll_terms_dev(self._h, rt._h, _ptr(dU), _ptr(dV), d, int(bool(subtract_mean)),
                                               mean_rating, u_lo, u_hi, v_lo, v_hi, _ptr(out3)))

    def pmf_log_likelihood(self, rt, dU, dV, d, sig, sig_u, sig_v, subtract_mean, mean_rating):
        ll, sq = _dbl(), _dbl()
        _check(self._lib.apmf_pmf_log_likelihood(self._h, rt._h, _ptr(dU), _ptr(dV), d, sig, sig_u,
                                                 sig_v, int(bool(subtract_mean)), mean_rating,
                                                 C.byref(ll), C.byref(sq)))
        return ll.value, sq.value

    def pmf_fit_lls(self, rt, dU, dV, d, sig, sig_u, sig_v, subtract_mean, mean_rating, lr, min_lr,
                    stop_thresh, max_steps=-1, max_trace=1 << 16, old_ll=None, return_ll=False):
        trace = np.empty(max(max_trace, 1), np.float64)
        lr_io, steps, conv = _dbl(lr), _i64(0), _i(0)
        ll_io = _dbl(float('nan') if old_ll is None else old_ll)
        _check(self._lib.apmf_pmf_fit_lls(self._h, rt._h, _ptr(dU), _ptr(dV), d, sig, sig_u, sig_v,
                                          int(bool(subtract_mean)), mean_rating, C.byref(lr_io),
                                          C.byref(ll_io), min_lr, stop_thresh, max_steps,
                                          trace.ctypes.data, max_trace, C.byref(steps), C.byref(conv)))
        res = (trace[:min(steps.value, max_trace)].copy(), steps.value, lr_io.value, bool(conv.value))
        return res + (ll_io.value,) if return_ll else res

    def pmf_batch_fit_lls(self, rt, dU, dV, d, H, ei, ej, ev, sig, sig_u, sig_v, subtract_mean, mean_ratings, lr,
                          min_lr, stop_thresh, max_steps=-1):
        """H fit_lls searches over the shared container `rt`, hypothesis h with the extra rating (ei[h], ej[h],
        ev[h]); dU (H, n, d) / dV (H, m, d) updated in place.  -> (final LL (H), accepted steps (H), converged (H))"""
        ei = np.ascontiguousarray(ei, np.int32)
        ej = np.ascontiguousarray(ej, np.int32)
        ev = np.ascontiguousarray(ev, np.float64)
        if not (ei.size == ej.size == ev.size == H):
            raise ValueError('one extra rating per hypothesis expected')
        mr = None if mean_ratings is None else np.ascontiguousarray(mean_ratings, np.float64)
        ll, steps, conv = np.empty(H), np.zeros(H, np.int64), np.zeros(H, np.int32)
        _check(self._lib.apmf_pmf_batch_fit_lls(self._h, rt._h, d, H, ei.ctypes.data, ej.ctypes.data, ev.ctypes.data,
                                                _ptr(dU), _ptr(dV), sig, sig_u, sig_v, int(bool(subtract_mean)),
                                                None if mr is None else mr.ctypes.data, lr, min_lr, stop_thresh,
                                                max_steps, ll.ctypes.data, steps.ctypes.data, conv.ctypes.data))
        return ll, steps, conv.astype(bool)

    def pmf_minibatch_epoch(self, shuffled, n, m, batch_size, lr, momentum, dU, dV, duinc, dvinc, d,
                            sig, sig_u, sig_v, subtract_mean, mean_rating):
        shuffled = np.ascontiguousarray(shuffled, dtype=np.float64)
        _check(self._lib.apmf_pmf_minibatch_epoch(self._h, shuffled.ctypes.data, shuffled.shape[0], n,
                                                  m, batch_size, lr, momentum, _ptr(dU), _ptr(dV),
                                                  _ptr(duinc), _ptr(dvinc), d, sig, sig_u, sig_v,
                                                  int(bool(subtract_mean)), mean_rating))

    def pmf_predicted_matrix(self, dU, dV, n, m, d, add, dout, ld):
        _check(self._lib.apmf_pmf_predicted_matrix(self._h, _ptr(dU), _ptr(dV), n, m, d, add,
                                                   _ptr(dout), ld))

    # -- K2 ---------------------------------------------------------------------------
    def mn_score_grid(self, dtype, B, n, m, d, ld, mean, cross, diag, omega, mask=None, var=None,
                      pmean=None, dsum=None, want=0, best_val=None, best_idx=None):
        fn = self._lib.apmf_mn_score_grid_f32 if np.dtype(dtype) == np.float32 \
            else self._lib.apmf_mn_score_grid_f64
        _check(fn(self._h, B, n, m, d, ld, _ptr(mean), _ptr(cross), _ptr(diag), _ptr(omega),
                  _ptr(mask), _ptr(var), _ptr(pmean), _ptr(dsum), want, _ptr(best_val),
                  _ptr(best_idx)))

    def mn_score_pool(self, n, m, d, ld, mean, cross, diag, omega, pi, pj, P, pmean, var):
        _check(self._lib.apmf_mn_score_pool_f64(self._h, n, m, d, ld, _ptr(mean), _ptr(cross),
                                                _ptr(diag), _ptr(omega), _ptr(pi), _ptr(pj), P,
                                                _ptr(pmean), _ptr(var)))

    def grid_collect_near(self, n, m, ld, var, mask, want, best, rel_tol, pi, pj, cap):
        cnt = _i(0)
        _check(self._lib.apmf_grid_collect_near_f32(self._h, n, m, ld, _ptr(var), _ptr(mask), want,
                                                    best, rel_tol, _ptr(pi), _ptr(pj), cap,
                                                    C.byref(cnt)))
        return cnt.value

    def grid_collect_eq(self, n, m, ld, var, mask, best, pi, pj, cap):
        cnt = _i(0)
        _check(self._lib.apmf_grid_collect_eq_f64(self._h, n, m, ld, _ptr(var), _ptr(mask), float(best),
                                                  _ptr(pi), _ptr(pj), cap, C.byref(cnt)))
        return cnt.value

    def full_score_grid(self, n, m, d, ld, mean, cov, mask=None, var=None, pmean=None, dsum=None,
                        want=0, best_val=None, best_idx=None):
        _check(self._lib.apmf_full_score_grid_f64(self._h, n, m, d, ld, _ptr(mean), _ptr(cov),
                                                  _ptr(mask), _ptr(var), _ptr(pmean), _ptr(dsum), want,
                                                  _ptr(best_val), _ptr(best_idx)))

    # -- dense symmetric linear algebra of the variational fit ------------------------------
    def dense_project_psd(self, dmat, N, min_eig):
        flag = _i(0)
        _check(load_dense().apmf_dense_project_psd(self._h, _ptr(dmat), N, float(min_eig), C.byref(flag)))
        return bool(flag.value)

    def dense_inv_logdet(self, dmat, N, dinv=None, want_logdet=True):
        ld = _dbl(0.)
        _check(load_dense().apmf_dense_inv_logdet(self._h, _ptr(dmat), N, _ptr(dinv),
                                                  C.byref(ld) if want_logdet else None))
        return ld.value

    def mn_kl_value(self, rt, d, mean, cov_ui, cov_lat, sig, sig_u, sig_v):
        kl = _dbl(0.)
        _check(load_dense().apmf_mn_kl_value(self._h, rt._h, d, _ptr(mean), _ptr(cov_ui), _ptr(cov_lat),
                                             sig, sig_u, sig_v, C.byref(kl)))
        return kl.value

    def mn_kl_gradient(self, rt, d, mean, cov_ui, cov_lat, sig, sig_u, sig_v, g_mean, g_cov_ui, g_cov_lat):
        _check(load_dense().apmf_mn_kl_gradient(self._h, rt._h, d, _ptr(mean), _ptr(cov_ui),
                                                _ptr(cov_lat), sig, sig_u, sig_v, _ptr(g_mean),
                                                _ptr(g_cov_ui), _ptr(g_cov_lat)))

    def full_kl_gradient(self, rt, d, mean, cov, sig, sig_u, sig_v, g_mean, g_cov):
        _check(load_dense().apmf_full_kl_gradient(self._h, rt._h, d, _ptr(mean), _ptr(cov), sig, sig_u, sig_v,
                                                  _ptr(g_mean), _ptr(g_cov)))

    def full_kl_value(self, rt, d, mean, cov, sig, sig_u, sig_v):
        kl = _dbl(0.)
        _check(load_dense().apmf_full_kl_value(self._h, rt._h, d, _ptr(mean), _ptr(cov), sig, sig_u, sig_v,
                                               C.byref(kl)))
        return kl.value

    def full_fit_normal(self, rt, d, mean, cov, sig, sig_u, sig_v, min_eig, lr, old_kl=None, max_steps=-1,
                        max_trace=1 << 16):
        """Device-resident ActivePMF.fit_normal_kls steps; returns (kl_trace, steps, lr, converged, old_kl)."""
        trace = np.empty(max(max_trace, 1), np.float64)
        lr_io = _dbl(lr)
        kl_io = _dbl(float('nan') if old_kl is None else old_kl)
        steps, conv = _i64(0), _i(0)
        _check(load_dense().apmf_full_fit_normal(self._h, rt._h, d, _ptr(mean), _ptr(cov), sig, sig_u, sig_v,
                                                 float(min_eig), C.byref(lr_io), C.byref(kl_io), max_steps,
                                                 trace.ctypes.data, max_trace, C.byref(steps), C.byref(conv)))
        return (trace[:min(steps.value, max_trace)].copy(), steps.value, lr_io.value, bool(conv.value),
                kl_io.value)

    def mn_fit_normal(self, rt, d, mean, cov_ui, cov_lat, sig, sig_u, sig_v, min_eig, lr, old_kl=None,
                      max_steps=-1, max_trace=1 << 16):
        """Device-resident fit_normal_kls steps; returns (kl_trace, steps, lr, converged, old_kl)."""
        trace = np.empty(max(max_trace, 1), np.float64)
        lr_io = _dbl(lr)
        kl_io = _dbl(float('nan') if old_kl is None else old_kl)
        steps, conv = _i64(0), _i(0)
        _check(load_dense().apmf_mn_fit_normal(self._h, rt._h, d, _ptr(mean), _ptr(cov_ui), _ptr(cov_lat),
                                               sig, sig_u, sig_v, float(min_eig), C.byref(lr_io),
                                               C.byref(kl_io), max_steps, trace.ctypes.data, max_trace,
                                               C.byref(steps), C.byref(conv)))
        return (trace[:min(steps.value, max_trace)].copy(), steps.value, lr_io.value, bool(conv.value),
                kl_io.value)

    # -- K3 / K4 -----------------------------------------------------------------------
    def gibbs_half_sweep(self, rt, side, oth, d, alpha, mu, beta, mean_sub, z, out):
        _check(self._lib.apmf_gibbs_half_sweep(self._h, rt._h, side, _ptr(oth), d, _ptr(alpha),
                                               _ptr(mu), beta, mean_sub, _ptr(z), _ptr(out)))

    def gibbs_half_sweep_rows(self, rt, side, lo, hi, oth, d, alpha, mu, beta, mean_sub, z, out):
        _check(self._lib.apmf_gibbs_half_sweep_rows(self._h, rt._h, side, lo, hi, _ptr(oth), d,
                                                    _ptr(alpha), _ptr(mu), beta, mean_sub, _ptr(z),
                                                    _ptr(out)))

    def gibbs_sample_hyperparam(self, feats, N, d, wi, b0, df, mu0, mu, alpha, variates=None, seed=0, offset=0,
                                status=None):
        """(mu, alpha) ~ Gaussian-Wishart posterior on the device (bayes_pmf.py:157-186)."""
        _check(self._lib.apmf_gibbs_sample_hyperparam(self._h, _ptr(feats), int(N), int(d), _ptr(wi), float(b0),
                                                      float(df), _ptr(mu0), _ptr(variates),
                                                      int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1),
                                                      _ptr(mu), _ptr(alpha), _ptr(status)))

    def gibbs_sample_hyperparam_batched(self, H, feats, N, d, wi, b0, df, mu0, variates, direct_dof, mu, alpha, status):
        """(mu, alpha) of H chains at once from supplied variates (bayes_pmf.py:157-186 per chain)."""
        _check(self._lib.apmf_gibbs_sample_hyperparam_batched(self._h, int(H), _ptr(feats), int(N), int(d), _ptr(wi),
                                                              float(b0), float(df), _ptr(mu0), _ptr(variates),
                                                              int(direct_dof), _ptr(mu), _ptr(alpha), _ptr(status)))

    def gibbs_half_sweep_batched(self, rt, side, H, oth, d, alpha, mu, beta, mean_sub, z, out, ei=None, ej=None,
                                 ev=None):
        _check(self._lib.apmf_gibbs_half_sweep_batched(self._h, rt._h, side, int(H), _ptr(oth), d, _ptr(alpha),
                                                       _ptr(mu), float(beta), _ptr(mean_sub), _ptr(z), _ptr(out),
                                                       _ptr(ei), _ptr(ej), _ptr(ev)))

    def fill_normal(self, out, seed, offset=0):
        """out[...] ~ N(0, 1) from the device generator (Philox4x32-10 + Box-Muller); not NumPy's stream."""
        _check(self._lib.apmf_fill_normal_f64(self._h, _ptr(out), out.size, int(seed) & (2 ** 64 - 1),
                                              int(offset) & (2 ** 64 - 1)))

    def sample_stats(self, S, n, m, d, ld, Us, Vs, add, cutoff, mean=None, var=None, prob=None):
        _check(self._lib.apmf_sample_stats_f64(self._h, S, n, m, d, ld, _ptr(Us), _ptr(Vs), add,
                                               cutoff, _ptr(mean), _ptr(var), _ptr(prob)))

    def grid_gather(self, grid, ld, pi, pj, P, out):
        _check(self._lib.apmf_grid_gather_f64(self._h, _ptr(grid), ld, _ptr(pi), _ptr(pj), P, _ptr(out)))

    def pool_argext(self, vals, P, want):
        pos, val = _i64(0), _dbl(0)
        _check(self._lib.apmf_pool_argext_f64(self._h, _ptr(vals), P, want, C.byref(pos), C.byref(val)))
        return pos.value, val.value


_default_ctx = {}


def default_device():
    """Device the Python classes use when none is named: APMF_DEVICE, else LOCAL_RANK (one
    process per GPU under torch.distributed.run), else 0 - modulo the visible devices."""
    want = int(os.environ.get('APMF_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    return want % max(device_count(), 1)


def release_thread_contexts(keep_current=True):
    """Close the per-thread default Contexts of threads other than the caller's (worker threads of
    the lookahead map keep theirs alive between calls: stream, workspaces, rocBLAS handle and a
    block cache each).  Only call this when no other thread is inside a library call."""
    me = (os.getpid(), threading.get_ident())
    for key in list(_default_ctx):
        if keep_current and key[:2] == me:
            continue
        ctx = _default_ctx.pop(key, None)
        if ctx is not None and key[0] == os.getpid():
            ctx.close()


def default_context(device=None):
    """Per-process, per-THREAD, per-device Context, created lazily (never before a fork: a HIP
    context does not survive it).  A Context is one stream plus its
    workspaces and must not be driven from two threads at once; giving every thread its own
    is what lets the lookahead hypotheses be re-fitted concurrently (_active_base.py)."""
    if device is None:
        device = default_device()
    key = (os.getpid(), threading.get_ident(), int(device))
    ctx = _default_ctx.get(key)
    if ctx is None:
        ctx = Context(device)
        _default_ctx[key] = ctx
    return ctx
