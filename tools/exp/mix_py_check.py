"""the quality test's sampler run (DE 0.8 + snooker 0.2, 1024 x 64) through the Python layer in variations: acceptance with persist_mix = 1 / 0"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import emcee_amd  # noqa: E402
from tools.quick_bench import dense_params  # noqa: E402

N, D = 1024, 64
mu, cov, icov = dense_params(D)
p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
for burn, nst, thin in ((0, 50, 1), (0, 50, 4), (100, 50, 1), (100, 50, 4), (37, 50, 2)):
    res = []
    for mix in (1, 0):
        mv = [(emcee_amd.moves.DEMove(), 0.8), (emcee_amd.moves.DESnookerMove(), 0.2)]
        s = emcee_amd.EnsembleSampler(N, D, emcee_amd.targets.DenseGaussian(mu, icov), rng="philox", moves=mv)
        s._random.seed(12)
        s._device_ensemble().set_tuning("persist_mix", mix)
        st = p0
        if burn:
            st = s.run_mcmc(p0, burn, skip_initial_state_check=True, store=False)
        s.run_mcmc(st, nst, thin_by=thin, skip_initial_state_check=True)
        res.append((float(np.mean(s.acceptance_fraction)), s.get_chain()[-1].copy(), s._device_ensemble().persist_info()["launches"]))
    print("burn %4d, %d x %d: acceptance mix %.4f one-move %.4f   last stored step equal %s   launches %d / %d" % (
        burn, nst, thin, res[0][0], res[1][0], np.array_equal(res[0][1], res[1][1]), res[0][2], res[1][2]), flush=True)
