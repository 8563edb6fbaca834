"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol that
include/apmf.h declares, and fails loudly (no fallback) when no GPU is present."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as ge
    ge.build()
    from active_matrix_factorization_amd import _lib
    return _lib


def test_header_and_binding_agree(lib):
    hdr = open(os.path.join(ROOT, 'include', 'apmf.h')).read()
    declared = set(re.findall(r'\b(apmf_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'apmf_status'}
    bound = set(lib.exported_symbols())
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    so, dense, comm = lib.load(), lib.load_dense(), lib.load_comm()
    for name in declared:
        assert getattr(dense if name in lib._DENSE_SIGS else comm if name in lib._COMM_SIGS else so, name) is not None


def test_no_gpu_fails_loudly(lib):
    if lib.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(lib.ApmfError) as e:
        lib.Context(0)
    assert e.value.status == lib.ENODEV
    assert 'no CPU fallback' in str(e.value)


def test_grid_ld(lib):
    so = lib.load()
    for m in (1, 63, 64, 65, 1682, 20000):
        assert so.apmf_grid_ld(m) == lib.grid_ld(m)
        assert lib.grid_ld(m) % lib.GRID_ALIGN == 0 and 0 <= lib.grid_ld(m) - m < lib.GRID_ALIGN
    assert lib.grid_ld(1682) == 1728


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'active_matrix_factorization_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, fn)).read()
                assert 'oracle' not in src.replace('oracle/', '').lower() or fn == '__init__.py' or \
                    'import oracle' not in src and 'from oracle' not in src
                assert 'from oracle' not in src and 'import oracle' not in src


def test_modules_import_like_the_reference():
    """The reference's scripts import their siblings by bare module name (python-pmf/ on
    sys.path); the drop-in modules support that, and classes then pickle under the same
    module.Class names as the reference's."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import pmf_cy, mn_active_pmf, active_pmf, bayes_pmf, "
            "matrix_normal_exps_cy, normal_exps_cy; "
            "assert mn_active_pmf.MNActivePMF.__module__ == 'mn_active_pmf'; "
            "assert issubclass(bayes_pmf.BayesianPMF, pmf_cy.ProbabilisticMatrixFactorization); "
            "assert mn_active_pmf.KEY_FUNCS['pred-variance'].chooser is max; print('ok')"
            % os.path.join(ROOT, 'active_matrix_factorization_amd'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr


def test_late_torch_import_is_a_clear_error():
    """importing torch after libapmf_dense.so (system rocBLAS / rocSOLVER) crashed the
    process inside torch's static initialisers.  The product no longer imports torch anywhere; a
    user who does so afterwards gets an ImportError that names the ordering rule (checked in a
    child process: the guard must fire BEFORE torch's libraries are touched)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from active_matrix_factorization_amd import _lib\n"
            "_lib.load()\n"
            "assert _lib._dense is not None, 'dense library not built'\n"
            "assert 'torch' not in sys.modules\n"
            "try:\n"
            "    import torch\n"
            "except ImportError as e:\n"
            "    assert 'BEFORE' in str(e), e\n"
            "    print('guarded')\n" % ROOT)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'guarded' in out.stdout, out.stdout + out.stderr
    # the other order stays allowed: torch first, then the package
    code2 = ("import sys; sys.path.insert(0, %r)\n"
             "import torch\n"
             "from active_matrix_factorization_amd import _lib\n"
             "_lib.load(); import torch.nn; print('fine')\n" % ROOT)
    out = subprocess.run([sys.executable, '-c', code2], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'fine' in out.stdout, out.stdout + out.stderr


def test_product_never_imports_torch():
    pkg = os.path.join(ROOT, 'active_matrix_factorization_amd')
    for fn in os.listdir(pkg) + ['../bench.py']:
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert 'import torch' not in src.replace("import torch AFTER", '').replace('"import torch', '') \
                .replace("import torch after", '').replace('Import torch BEFORE', ''), fn


def test_cli_refuses_arbitrary_pickles(tmp_path):
    """--load-data / --load-model go through a restricted unpickler: arrays and builtin containers
    load, anything that would run code on load is refused."""
    import pickle
    import numpy as np
    from active_matrix_factorization_amd import _cli
    good = tmp_path / 'data.pkl'
    with open(good, 'wb') as f:
        pickle.dump({'_real': np.eye(3), '_ratings': np.zeros((2, 3)), '_rating_vals': (1, 2)}, f)
    d = _cli._load_data(str(good))
    assert set(d) == {'_real', '_ratings', '_rating_vals'} and d['_real'].shape == (3, 3)

    class Evil(object):
        def __reduce__(self):
            import os
            return (os.system, ('echo pwned > %s' % (tmp_path / 'pwned'),))
    bad = tmp_path / 'bad.pkl'
    with open(bad, 'wb') as f:
        pickle.dump({'_real': Evil()}, f)
    with pytest.raises(pickle.UnpicklingError):
        _cli._load_data(str(bad))
    assert not (tmp_path / 'pwned').exists()
    npz = tmp_path / 'data.npz'
    np.savez(npz, _real=np.eye(2), _ratings=np.zeros((1, 3)))
    assert set(_cli._load_data(str(npz))) == {'_real', '_ratings'}
