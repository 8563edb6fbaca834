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
    # a protocol-4 pickle can name 'os.getcwd' THROUGH one of the package's own modules (dotted
    # names are resolved attribute by attribute), or a function of such a module instead of a class
    import io
    import pickletools
    for module, name in (('active_matrix_factorization_amd._active_base', 'os.getcwd'),
                         ('active_matrix_factorization_amd.pmf_cy', 'np.load'),
                         ('active_matrix_factorization_amd.bayes_pmf', 'os.system'),
                         ('active_matrix_factorization_amd.pmf_cy', 'fake_ratings'),
                         ('pmf_cy', 'deepcopy'), ('os', 'getcwd')):
        payload = (b'\x80\x04' + b'\x8c' + bytes([len(module)]) + module.encode() +
                   b'\x8c' + bytes([len(name)]) + name.encode() + b'\x93' + b')R.')      # STACK_GLOBAL, (), REDUCE
        pickletools.dis(payload, out=io.StringIO())       # well-formed
        with pytest.raises(pickle.UnpicklingError):
            _cli.restricted_load(io.BytesIO(payload))
    # the classes themselves load, under the package's name and under the reference's bare module name
    for module in ('active_matrix_factorization_amd.mn_active_pmf', 'mn_active_pmf'):
        payload = (b'\x80\x04' + b'\x8c' + bytes([len(module)]) + module.encode() + b'\x8c\x0bMNActivePMF\x93.')
        cls = _cli.restricted_load(io.BytesIO(payload))
        assert cls.__module__ == 'active_matrix_factorization_amd.mn_active_pmf' and cls.__name__ == 'MNActivePMF'
    npz = tmp_path / 'data.npz'
    np.savez(npz, _real=np.eye(2), _ratings=np.zeros((1, 3)))
    assert set(_cli._load_data(str(npz))) == {'_real', '_ratings'}


def test_lazy_pairs_state_and_deepcopy():
    """the storage behind the rated / unrated properties ('_rated_set', '_unrated_set',
    '_pairs_recipe') is not model state - a pickle carries 'rated' / 'unrated' only, like the reference's
    (pmf_cy.pyx:90-126) - and a deep copy of a model whose sets are still a recipe (from_dense) does not build
    the 10^6-tuple sets: it gets its own recipe and masks."""
    import copy
    import pickle
    import numpy as np
    from active_matrix_factorization_amd import pmf_cy, mn_active_pmf, bayes_pmf
    rs = np.random.RandomState(3)
    n, m = 7, 9
    known = rs.rand(n, m) < .4
    known[np.arange(n), np.arange(n)] = True
    known[n - 1, m - 1] = True
    real = rs.randint(1, 6, (n, m)).astype(float)
    ii, jj = known.nonzero()
    ratings = np.column_stack([ii, jj, real[ii, jj]]).astype(float)
    for cls, kw in ((pmf_cy.ProbabilisticMatrixFactorization, {}), (mn_active_pmf.MNActivePMF, {}),
                    (bayes_pmf.BayesianPMF, {})):
        eager = cls(ratings.copy(), latent_d=2, **kw)
        lazy = cls(ratings.copy(), latent_d=2, knowable=pmf_cy._PairsRecipe(ratings[:, :2].copy(), None, n, m), **kw)
        free = sorted(eager.unrated)[:2]
        lazy.add_rating(free[0][0], free[0][1], 3.)
        eager.add_rating(free[0][0], free[0][1], 3.)
        dup = copy.deepcopy(lazy)
        assert lazy.__dict__['_pairs_recipe'] is not None and dup.__dict__['_pairs_recipe'] is not None
        assert dup.__dict__['_pairs_recipe'] is not lazy.__dict__['_pairs_recipe']
        dup.add_rating(free[1][0], free[1][1], 4.)            # the copy's masks are its own
        assert lazy._n_rated() == eager._n_rated() == dup._n_rated() - 1
        assert lazy.__dict__['_pairs_recipe'] is not None
        with pytest.raises(ValueError):
            dup.add_rating(free[1][0], free[1][1], 4.)
        state = lazy.__getstate__()
        flat = dict(state)
        flat.update(flat.pop('__dict__', {}))
        assert not {'_rated_set', '_unrated_set', '_pairs_recipe', '_dev'} & set(flat)
        assert state['rated'] == eager.rated and state['unrated'] == eager.unrated
        assert list(state['unrated']) == list(eager.unrated)        # same construction sequence: same iteration order
        back = pickle.loads(pickle.dumps(lazy))
        assert back.rated == eager.rated and back.unrated == eager.unrated
        assert dup.rated == eager.rated | {free[1]}


def test_late_torch_guard_lets_probes_through():
    """once libapmf_dense.so is loaded, `import torch` must fail with an explanation (the SONAME clash
    would crash the process) - but importlib.util.find_spec('torch'), the usual availability probe, must not."""
    import subprocess
    import sys
    code = (
        "import importlib.util, sys\n"
        "from active_matrix_factorization_amd import _lib\n"
        "_lib.load_dense()\n"
        "assert 'torch' not in sys.modules\n"
        "spec = importlib.util.find_spec('torch')\n"
        "assert spec is not None and spec.name == 'torch'\n"
        "assert importlib.util.find_spec('no_such_module_xyz') is None\n"
        "try:\n"
        "    import torch\n"
        "except ImportError as e:\n"
        "    assert 'libapmf_dense' in str(e), e\n"
        "else:\n"
        "    raise SystemExit('late import torch was not refused')\n"
        "assert 'torch' not in sys.modules\n"
        "print('ok')\n")
    out = subprocess.run([sys.executable, '-c', code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == 'ok', out.stderr[-2000:]
