"""MI355X-native hot path of active probabilistic matrix factorization.

Drop-in Python mirror of the reference's python-pmf class surface
(ProbabilisticMatrixFactorization, MNActivePMF, ActivePMF, BayesianPMF) whose hot methods
call hand-written HIP kernels (csrc/) through the C-ABI of include/apmf.h via ctypes.
There is no CPU fallback: without libapmf_hip.so and a gfx950 device the hot methods raise.
"""
__version__ = '0.1.0'
