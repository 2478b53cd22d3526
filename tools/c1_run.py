"""BASELINE configs[0] (32 walkers x 5 dims, isotropic Gaussian, StretchMove): 40 000 native steps + 40 000 exact steps."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402
from emcee_amd import targets  # noqa: E402

p0 = np.random.RandomState(1234).randn(32, 5)
for rng in ("philox", "mt19937"):
    s = emcee_amd.EnsembleSampler(32, 5, targets.IsoGaussian(), rng=rng)
    s.run_mcmc(p0, 2000, store=False)
    t0 = time.perf_counter()
    s.run_mcmc(None, 40000, store=False)
    dt = time.perf_counter() - t0
    print("%s: %.2f us/step, %.0f steps/s" % (rng, dt / 40000 * 1e6, 40000 / dt), flush=True)
