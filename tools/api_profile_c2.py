"""Where does a run_mcmc call at the headline size spend its host time?  (cProfile, Philox mode, store=False)"""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402
from bench import dense_gaussian  # noqa: E402

N, D = 65536, 64
mu, cov, icov = dense_gaussian(D)
p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
s = emcee_amd.EnsembleSampler(N, D, emcee_amd.targets.DenseGaussian(mu, icov), rng="philox")
st = s.run_mcmc(p0, 50, store=False, skip_initial_state_check=True)
for nst in (400, 400, 4000):
    t0 = time.perf_counter()
    st = s.run_mcmc(st, nst, store=False, skip_initial_state_check=True)
    print("run_mcmc(%d), state handed back unread: %.1f us/step" % (nst, (time.perf_counter() - t0) / nst * 1e6))
t0 = time.perf_counter()
st = s.run_mcmc(emcee_amd.State(st.coords, log_prob=st.log_prob, random_state=st.random_state), 400, store=False, skip_initial_state_check=True)
st.coords
print("run_mcmc(400), arrays down and up again: %.1f us/step" % ((time.perf_counter() - t0) / 400 * 1e6))
pr = cProfile.Profile()
pr.enable()
st = s.run_mcmc(st, 400, store=False, skip_initial_state_check=True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
