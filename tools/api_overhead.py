"""Per-step cost of the Python surface on a small ensemble (32 x 5): run_mcmc (one native call) vs the sample()
generator (one native call per yielded state), both RNG modes."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402
from emcee_amd import targets  # noqa: E402

p0 = np.random.RandomState(1).randn(32, 5)
for rng in ("philox", "mt19937"):
    s = emcee_amd.EnsembleSampler(32, 5, targets.IsoGaussian(), rng=rng)
    s.run_mcmc(p0, 200)
    t0 = time.perf_counter()
    s.run_mcmc(None, 20000)
    a = (time.perf_counter() - t0) / 20000
    s = emcee_amd.EnsembleSampler(32, 5, targets.IsoGaussian(), rng=rng)
    n = 0
    t0 = time.perf_counter()
    for _ in s.sample(p0, iterations=4000):
        n += 1
    b = (time.perf_counter() - t0) / n
    s = emcee_amd.EnsembleSampler(32, 5, targets.IsoGaussian(), rng=rng)
    t0 = time.perf_counter()
    for _ in s.sample(p0, iterations=400, thin_by=10):
        pass
    c = (time.perf_counter() - t0) / 4000
    print("%-8s run_mcmc %.2f us/step | sample() %.1f us/step | sample(thin_by=10) %.1f us/step" % (rng, a * 1e6, b * 1e6, c * 1e6), flush=True)
