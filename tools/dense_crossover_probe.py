"""Fused dense kernel (ndim <= 112) against the wide-target path (tuning dense_wide = 1) on the ndims both can take."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from emcee_amd import _lib                      # noqa: E402
from emcee_amd.device import DeviceEnsemble    # noqa: E402
from bench import dense_gaussian               # noqa: E402

N = 65536
for D in (96, 112, 113, 120, 128):
    for wide in (0, 1):
        ens = DeviceEnsemble(N, D)
        mu, cov, icov = dense_gaussian(D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        ens.set_tuning("dense_wide", wide)
        ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
        ens.eval_state_log_prob()
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(7, 0)
        ens.run(50, 1, False)
        ens.sync()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            ens.run(200, 1, False)
            ens.sync()
            best = min(best, time.perf_counter() - t0)
        print("dense %d wide=%d: %.2f us/step" % (D, wide, best / 200 * 1e6), flush=True)
        ens.close()
