import sys, time, os
sys.path.insert(0, '/root/repo')
os.environ['APMF_DEVICE_RNG'] = '1'
import numpy as np
from itertools import islice
import bench
from active_matrix_factorization_amd import bayes_pmf
ratings, rated = bench.synth_ratings(0)
np.random.seed(0)
b = bayes_pmf.BayesianPMF(ratings, latent_d=15)
print('device_rng', getattr(b, 'device_rng', None))
it = b.samples()
list(islice(it, 5))
t = time.perf_counter(); s = list(islice(it, 50)); dt = (time.perf_counter() - t) / 50
print('yield ms (device rng)', dt * 1e3)
