"""A user's torch log_prob_fn on the device (targets.DeviceCallable) against the fused target and the host callable, C2's shape."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402
from emcee_amd import targets  # noqa: E402
from bench import dense_gaussian  # noqa: E402
import torch  # noqa: E402

N, D = 65536, 64
mu, cov, icov = dense_gaussian(D)
p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
dev = torch.device("cuda", 0)
mu_t, icov_t = torch.as_tensor(mu, device=dev), torch.as_tensor(icov, device=dev)


def lp_torch(q):
    d = q - mu_t
    return -0.5 * ((d @ icov_t) * d).sum(1)


def lp_numpy(x):
    d = x - mu
    return -0.5 * np.einsum("ij,ij->i", d @ icov, d)


def measure(label, target, nst, **kw):
    s = emcee_amd.EnsembleSampler(N, D, target, rng="philox", **kw)
    s._random.seed(3)
    st = s.run_mcmc(p0, 10, store=False, skip_initial_state_check=True)
    t_end = time.perf_counter() + 0.3                   # clocks up
    while time.perf_counter() < t_end:
        st = s.run_mcmc(st, nst, store=False, skip_initial_state_check=True)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        st = s.run_mcmc(st, nst, store=False, skip_initial_state_check=True)
        if s._ens is not None:
            s._ens.sync()
        best = min(best, (time.perf_counter() - t0) / nst)
    print("%-40s %8.1f us/step" % (label, best * 1e6), flush=True)


measure("fused DenseGaussian target", targets.DenseGaussian(mu, icov), 400)
measure("DeviceCallable(torch fn), eager", targets.DeviceCallable(lp_torch), 400)
measure("DeviceCallable(torch fn), HIP graph", targets.DeviceCallable(lp_torch, graph=True), 400)
measure("host callable, vectorize=True (split-phase)", lp_numpy, 20, vectorize=True)
