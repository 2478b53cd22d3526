"""Per-step cost with an ordinary Python log_prob_fn (split-phase path: proposal and accept/commit on the GPU,
the callable on the host) on small ensembles -- the regime most emcee users live in."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402


def lp_vec(x):
    return -0.5 * np.sum(x * x, axis=1)


def lp_one(x):
    return -0.5 * np.sum(x * x)


for N, D in ((32, 5), (128, 10), (1024, 16)):
    p0 = np.random.RandomState(1).randn(N, D)
    for label, fn, vec in (("vectorize=True", lp_vec, True), ("per-walker", lp_one, False)):
        s = emcee_amd.EnsembleSampler(N, D, fn, vectorize=vec)
        s.run_mcmc(p0, 20)
        n = 1500 if vec else 300
        t0 = time.perf_counter()
        s.run_mcmc(None, n)
        dt = (time.perf_counter() - t0) / n
        print("emcee_amd split-phase %5d x %-3d %-15s %.1f us/step  %.0f steps/s" % (N, D, label, dt * 1e6, 1 / dt), flush=True)
