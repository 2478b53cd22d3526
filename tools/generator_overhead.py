"""Cost per yielded step of the sample() generator at the headline size, both RNG modes, against one-call run_mcmc."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402
from bench import dense_gaussian  # noqa: E402

N, D = 65536, 64
mu, cov, icov = dense_gaussian(D)
p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
for rng in ("mt19937", "philox"):
    s = emcee_amd.EnsembleSampler(N, D, emcee_amd.targets.DenseGaussian(mu, icov), rng=rng)
    st = s.run_mcmc(p0, 50, store=False, skip_initial_state_check=True)
    t0 = time.perf_counter()
    st = s.run_mcmc(st, 400, store=False, skip_initial_state_check=True)
    t_run = (time.perf_counter() - t0) / 400
    t0 = time.perf_counter()
    n = 0
    for st in s.sample(st, iterations=400, store=False, skip_initial_state_check=True):
        n += 1
    t_gen = (time.perf_counter() - t0) / n
    print("%s: run_mcmc %.1f us/step, sample() generator %.1f us/step" % (rng, t_run * 1e6, t_gen * 1e6))
