import cProfile, pstats, sys
import numpy as np
sys.path.insert(0, ".")
import emcee_amd
from emcee_amd import targets
p0 = np.random.RandomState(1).randn(32, 5)
for rng in ("philox", "mt19937"):
    s = emcee_amd.EnsembleSampler(32, 5, targets.IsoGaussian(), rng=rng)
    s.run_mcmc(p0, 50)
    pr = cProfile.Profile()
    pr.enable()
    for _ in s.sample(p0, iterations=3000):
        pass
    pr.disable()
    print("=====", rng)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
