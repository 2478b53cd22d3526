import sys, time
import numpy as np
sys.path.insert(0, ".")
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from bench import dense_gaussian
for D in (64, 80, 96, 100, 112):
    for wide in (0, 1):
        N = 65536
        ens = DeviceEnsemble(N, D)
        mu, cov, icov = dense_gaussian(D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        ens.set_tuning("dense_wide", wide)
        p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
        ens.set_state(p0); ens.eval_state_log_prob()
        ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(7, 0)
        ens.run(50, 1, False); ens.sync()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); ens.run(200, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
        print("dense %d wide=%d: %.2f us/step" % (D, wide, best / 200 * 1e6), flush=True)
        ens.close()
