"""cProfile of the split-phase (Python log_prob_fn) path at 32 x 5: where do the ~100 us per step go?"""
import cProfile
import pstats
import sys

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402

N, D = 32, 5
p0 = np.random.RandomState(1).randn(N, D)
s = emcee_amd.EnsembleSampler(N, D, lambda x: -0.5 * np.sum(x * x, axis=1), vectorize=True)
s.run_mcmc(p0, 200)
pr = cProfile.Profile()
pr.enable()
s.run_mcmc(None, 3000)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
