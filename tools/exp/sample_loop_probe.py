"""sampler.sample() one step per iteration (the progress-bar / convergence-check loop) in exact mode at mid sizes: us per iteration
with the persistent exact path (persist_exact = 1) and without"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import emcee_amd
from emcee_amd import targets
for N, D in ((1024, 5), (4096, 16)):
    for pe in (1, 0):
        s = emcee_amd.EnsembleSampler(N, D, targets.IsotropicGaussian() if hasattr(targets, "IsotropicGaussian") else targets.DiagGaussian(np.zeros(D), np.ones(D)))
        s._device_ensemble().set_tuning("persist_exact", pe)
        p0 = np.random.RandomState(1).randn(N, D)
        st = s.run_mcmc(p0, 50, skip_initial_state_check=True, store=False)
        t0 = time.perf_counter()
        n = 0
        for _ in s.sample(st, iterations=1500, skip_initial_state_check=True, store=False):
            n += 1
        t1 = time.perf_counter()
        t2 = time.perf_counter()
        s.run_mcmc(None if False else st, 1500, skip_initial_state_check=True, store=False)
        t3 = time.perf_counter()
        print("N=%d D=%d persist_exact=%d: sample() %.1f us/iteration, run_mcmc %.1f us/step" % (N, D, pe, (t1 - t0) * 1e6 / n, (t3 - t2) * 1e6 / 1500), flush=True)
