"""The reference's quickstart tutorial (docs/tutorials/quickstart.ipynb:76-327): 5-d correlated Gaussian, 32 walkers,
100 burn-in + 10 000 steps, StretchMove -- published outputs: mean acceptance 0.552, mean tau 57.112 steps."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402
from emcee_amd import targets  # noqa: E402

out = {}
for rng in ("mt19937", "philox"):
    np.random.seed(42)
    ndim = 5
    means = np.random.rand(ndim)
    cov = 0.5 - np.random.rand(ndim ** 2).reshape((ndim, ndim))
    cov = np.triu(cov)
    cov += cov.T - np.diag(cov.diagonal())
    cov = np.dot(cov, cov)
    nwalkers = 32
    p0 = np.random.rand(nwalkers, ndim)
    sampler = emcee_amd.EnsembleSampler(nwalkers, ndim, targets.DenseGaussian(means, np.linalg.inv(cov)), rng=rng)
    state = sampler.run_mcmc(p0, 100)
    sampler.reset()
    sampler.run_mcmc(state, 10000)
    acc = float(np.mean(sampler.acceptance_fraction))
    tau = float(np.mean(sampler.get_autocorr_time()))
    out[rng] = dict(acceptance=acc, tau=tau)
    print("%-8s mean acceptance %.3f (tutorial 0.552)   mean tau %.2f steps (tutorial 57.112)" % (rng, acc, tau), flush=True)
json.dump(out, open("gpurun_out/quickstart_check.json", "w"), indent=1)
