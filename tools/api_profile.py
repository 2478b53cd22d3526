import cProfile, pstats, sys
import numpy as np
sys.path.insert(0, ".")
import emcee_amd
p0 = np.random.RandomState(1).randn(32, 5)
s = emcee_amd.EnsembleSampler(32, 5, lambda x: -0.5 * np.sum(x * x, axis=1), vectorize=True)
s.run_mcmc(p0, 50)
pr = cProfile.Profile()
pr.enable()
s.run_mcmc(None, 2000)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
