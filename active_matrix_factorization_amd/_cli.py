"""Command-line front ends of mn_active_pmf / active_pmf / bayes_pmf: the flags of the
reference's main() functions (mn_active_pmf.py:1011-1128, active_pmf.py:1095-1257,
bayes_pmf.py:828-937), so the results/*/Makefile targets keep working unchanged."""
import argparse
import os
import pickle
import sys

import numpy as np

try:
    from .pmf_cy import parse_fit_type
except ImportError:
    from pmf_cy import parse_fit_type


def _bool_opt(parser, name, default=False):
    parser.add_argument('--' + name, action='store_true', default=default)
    parser.add_argument('--no-' + name, action='store_false', dest=name.replace('-', '_'))


class _RestrictedUnpickler(pickle.Unpickler):
    """Data and result files only ever need NumPy arrays, builtin containers, argparse.Namespace and
    this package's own model classes; anything else in a pickle handed to the CLI is refused instead
    of executed (--load-data / --load-model took arbitrary pickles)."""
    _ALLOWED = {
        ('numpy', 'ndarray'), ('numpy', 'dtype'), ('numpy.core.multiarray', '_reconstruct'),
        ('numpy._core.multiarray', '_reconstruct'), ('numpy.core.multiarray', 'scalar'),
        ('numpy._core.multiarray', 'scalar'), ('numpy._core.numeric', '_frombuffer'),
        ('numpy.core.numeric', '_frombuffer'), ('argparse', 'Namespace'), ('builtins', 'set'),
        ('builtins', 'frozenset'), ('builtins', 'slice'), ('builtins', 'range'), ('builtins', 'complex'),
        ('collections', 'OrderedDict'), ('copyreg', '_reconstructor'), ('builtins', 'object'),
    }
    # model classes a results / model file may name, per module of this package.  A bare module name
    # (what the reference's own pickles say: 'mn_active_pmf.MNActivePMF') is resolved to THIS package's
    # module of that name, never to whatever 'mn_active_pmf' happens to be importable from sys.path.
    _OWN_CLASSES = {
        'pmf_cy': ('ProbabilisticMatrixFactorization', '_PairsRecipe'),
        'mn_active_pmf': ('MNActivePMF',),
        'active_pmf': ('ActivePMF',),
        'bayes_pmf': ('BayesianPMF',),
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        base = module.rpartition('.')[2]
        pkg = __package__ or ''
        # protocol-4 pickles resolve dotted names attribute by attribute ('os.getcwd' through a module
        # that imports os): only plain class names from the explicit list, and the object found must be
        # a class defined in that very module
        if '.' not in name and name in self._OWN_CLASSES.get(base, ()) and \
                module in (base, (pkg + '.' + base) if pkg else base):
            import importlib
            mod = importlib.import_module((pkg + '.' + base) if pkg else base)
            obj = getattr(mod, name, None)
            if isinstance(obj, type) and obj.__module__ == mod.__name__:
                return obj
        raise pickle.UnpicklingError("refusing to load %s.%s from a pickle: only NumPy arrays, builtin "
                                     "containers and this package's model classes are accepted (convert the "
                                     "file to .npz)" % (module, name))


def restricted_load(f):
    return _RestrictedUnpickler(f).load()


def _load_data(path):
    """.npz/.npy via numpy (nothing is executed), else a pickle of plain arrays / containers such
    as choose_training.py and generate.py write, through the restricted unpickler."""
    with open(path, 'rb') as f:
        try:
            data = np.load(f, allow_pickle=False)
        except Exception:
            f.seek(0)
            data = restricted_load(f)
        if isinstance(data, np.ndarray):
            data = {'_real': data}
        return {k: data[k] for k in ('_real', '_ratings', '_rating_vals', '_test_on', '_is_new_item')
                if k in data}


def _save(results, args):
    if args.save_results:
        print("saving results in '{}'".format(args.save_results))
        results['_args'] = args
        with open(args.save_results, 'wb') as f:
            pickle.dump(results, f)


def _results_path(args):
    if args.save_results is True:
        args.save_results = 'results.pkl'
    elif args.save_results:
        dirname = os.path.dirname(args.save_results)
        if dirname and not os.path.exists(dirname):
            os.makedirs(dirname)


def _check_keys(args, key_names):
    for k in args.keys:
        if k not in key_names:
            sys.stderr.write("Invalid key name %s; options are %s.\n" % (k, ', '.join(sorted(key_names))))
            sys.exit(1)
    if not args.keys:
        args.keys = sorted(key_names)


def run_cli(key_funcs, compare, kind):
    key_names = set(key_funcs)
    parser = argparse.ArgumentParser()
    model = parser.add_argument_group("Model Options")
    model.add_argument('--latent-d', '-D', type=int, default=5)
    model.add_argument('--discrete-integration', nargs='?', const=True, default=False)
    model.add_argument('--continuous-integration', action='store_false', dest='discrete_integration')
    _bool_opt(model, 'fit-sigmas', default=False)
    _bool_opt(model, 'refit-lookahead', default=False)
    model.add_argument('--fit', default='batch')
    for s in ('u', 'v'):
        model.add_argument('--sig-%s-mean' % s, type=float, default=0)
        model.add_argument('--sig-%s-var' % s, type=float, default=-1)
    model.add_argument('keys', nargs='*', help="Choices: {}.".format(', '.join(sorted(key_names))))
    if kind == 'mn':
        parser.add_argument('--load-data', default=None, metavar='FILE')
    else:
        prob = parser.add_argument_group("Problem Definiton")
        prob.add_argument('--load-data', default=None, metavar='FILE')
        prob.add_argument('--gen-rank', '-R', type=int, default=5)
        prob.add_argument('--type', default='float')
        prob.add_argument('--noise', '-n', type=float, default=.25)
        prob.add_argument('--num-users', '-N', type=int, default=10)
        prob.add_argument('--num-items', '-M', type=int, default=10)
        prob.add_argument('--u-mean', type=float, default=0)
        prob.add_argument('--u-std', type=float, default=2)
        prob.add_argument('--v-mean', type=float, default=0)
        prob.add_argument('--v-std', type=float, default=2)
        prob.add_argument('--mask', '-m', default=0)
        prob.add_argument('--load-model', default=None, metavar='FILE')
    running = parser.add_argument_group("Running")
    running.add_argument('--processes', '-P', type=int, default=None)
    _bool_opt(running, 'threading', True)
    running.add_argument('--steps', '-s', type=int, default=None)
    results = parser.add_argument_group("Results")
    results.add_argument('--save-results', default=True, metavar='FILE')
    results.add_argument('--no-save-results', action='store_false', dest='save_results')
    results.add_argument('--note', action='append')
    args = parser.parse_args()
    _check_keys(args, key_names)
    _results_path(args)

    common = dict(latent_d=args.latent_d, discrete_exp=args.discrete_integration,
                  refit_lookahead=args.refit_lookahead, fit_sigmas=args.fit_sigmas,
                  sig_u_mean=args.sig_u_mean, sig_u_var=args.sig_u_var,
                  sig_v_mean=args.sig_v_mean, sig_v_var=args.sig_v_var, steps=args.steps,
                  fit_type=parse_fit_type(args.fit), processes=args.processes,
                  do_threading=args.threading)
    if kind == 'mn':
        data = _load_data(args.load_data)
        real = data['_real']
        knowable = np.isfinite(real)
        knowable[real == 0] = False
        test_on = data.get('_test_on')
        if test_on is not None:
            print("using test set from file")
            knowable[test_on] = False
        res = compare(args.keys, real=real, ratings=data['_ratings'],
                      rating_vals=data.get('_rating_vals'), knowable=knowable,
                      test_on=test_on, **common)
    else:
        real_ratings_vals, apmf, knowable, gen = None, None, None, {}
        if args.load_data is not None:
            data = _load_data(args.load_data)
            real = data['_real']
            real_ratings_vals = (real, data['_ratings'], data.get('_rating_vals'))
            kn = np.isfinite(real)
            kn[real == 0] = False
            knowable = zip(*kn.nonzero())
        else:
            try:
                args.type = int(args.type)
            except ValueError:
                pass
            try:
                args.mask = float(args.mask)
            except ValueError:
                pass
            gen = dict(num_users=args.num_users, num_items=args.num_items, rank=args.gen_rank,
                       noise=args.noise, mask_type=args.mask, data_type=args.type,
                       u_mean=args.u_mean, u_std=args.u_std, v_mean=args.v_mean, v_std=args.v_std)
        if args.load_model:
            with open(args.load_model, 'rb') as f:
                apmf = restricted_load(f)['_initial_apmf']
        res = compare(args.keys, real_ratings_vals=real_ratings_vals, apmf=apmf, knowable=knowable,
                      **common, **gen)
    _save(res, args)


def run_bayes_cli(keys, compare_active):
    key_names = keys.keys()
    parser = argparse.ArgumentParser()
    parser.add_argument('--latent-d', '-D', type=int, default=5)
    parser.add_argument('--steps', '-s', type=int, default=None)
    parser.add_argument('--discrete', action='store_true', default=None)
    parser.add_argument('--no-discrete', action='store_false', dest='discrete')
    parser.add_argument('--subtract-mean', action='store_true', default=True)
    parser.add_argument('--no-subtract-mean', action='store_false', dest='subtract_mean')
    parser.add_argument('--fit', default='batch')
    parser.add_argument('--lookahead-fit', default='batch')
    parser.add_argument('--samps', '-S', type=int, default=128)
    parser.add_argument('--lookahead-samps', type=int, default=128)
    parser.add_argument('--threaded', action='store_true', default=True)
    parser.add_argument('--unthreaded', action='store_false', dest='threaded')
    parser.add_argument('--procs', '-P', type=int, default=None)
    parser.add_argument('--test-set', default='all')
    parser.add_argument('--load-data', required=True, metavar='FILE')
    parser.add_argument('--save-results', nargs='?', default=True, const=True, metavar='FILE')
    parser.add_argument('--no-save-results', action='store_false', dest='save_results')
    parser.add_argument('--note', action='append')
    parser.add_argument('keys', nargs='*', help="Choices: {}.".format(', '.join(sorted(key_names))))
    args = parser.parse_args()
    _check_keys(args, set(key_names))
    _results_path(args)
    data = _load_data(args.load_data)
    rating_vals = data.get('_rating_vals')
    if args.discrete is None:
        args.discrete = rating_vals is not None
    res = compare_active(key_names=args.keys, latent_d=args.latent_d, real=data['_real'],
                         ratings=data['_ratings'], rating_vals=rating_vals, test_set=args.test_set,
                         num_steps=args.steps, discrete=args.discrete,
                         subtract_mean=args.subtract_mean, fit_type=parse_fit_type(args.fit),
                         lookahead_fit=args.lookahead_fit, num_samps=args.samps,
                         lookahead_samps=args.lookahead_samps, procs=args.procs,
                         threaded=args.threaded)
    _save(res, args)
