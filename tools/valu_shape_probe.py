"""General against LEAN instantiations of k_halfstep on shapes outside the bench configurations (EMX_NO_LEAN_X=1: general)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from emcee_amd import _lib                      # noqa: E402
from emcee_amd.device import DeviceEnsemble    # noqa: E402
from bench import dense_gaussian               # noqa: E402

for N, D, tgt in ((65536, 5, "iso"), (65536, 16, "rosen"), (65536, 64, "iso"), (65536, 5, "dense"), (65536, 32, "dense"), (65536, 100, "dense")):
    ens = DeviceEnsemble(N, D)
    rs = np.random.RandomState(1)
    if tgt == "iso":
        ens.set_target(_lib.TARGET_ISO)
        p0 = rs.randn(N, D)
    elif tgt == "dense":
        mu, cov, icov = dense_gaussian(D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        p0 = mu + rs.randn(N, D) @ np.linalg.cholesky(cov).T
    else:
        ens.set_target(_lib.TARGET_ROSENBROCK, scale=20.0)
        p0 = 1 + 0.1 * rs.randn(N, D)
    mvk = int(os.environ.get("MOVE", "0"))            # 0 stretch, 1 DE, 2 snooker
    if mvk:
        ens.set_moves([_lib.MoveDesc(mvk, 4 if mvk == 2 else 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * D), 1.7)], np.array([1.0]))
    ens.set_state(p0)
    ens.eval_state_log_prob()
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(7, 0)
    ens.run(50, 1, False)
    ens.sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        ens.run(400, 1, False)
        ens.sync()
        best = min(best, time.perf_counter() - t0)
    print("%s %dx%d move %d: %.2f us/step" % (tgt, N, D, mvk, best / 400 * 1e6), flush=True)
    ens.close()
