"""how long the host takes to enqueue one emx_run of K steps at C2's shape (the call returns when everything is enqueued), against the
GPU time of the same steps: the part of a driver-clock block (sync, run(K), sync) during which the device waits for the host"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402
from tools.quick_bench import dense_params  # noqa: E402

N, D = 65536, 64
mu, cov, icov = dense_params(D)
ens = DeviceEnsemble(N, D)
ens.set_target(_lib.TARGET_DENSE, mu, icov)
ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
ens.set_rng_mode(_lib.RNG_PHILOX)
ens.set_philox(7, 0)
ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
ens.eval_state_log_prob()
ens.run(200, 1, False)
ens.sync()
for K in (1, 4, 16, 20, 32):
    enq, tot = [], []
    for _ in range(40):
        ens.sync()
        t0 = time.perf_counter()
        ens.run(K, 1, False)
        t1 = time.perf_counter()
        ens.sync()
        t2 = time.perf_counter()
        enq.append(t1 - t0)
        tot.append(t2 - t0)
    print("K=%2d: enqueue %.1f us (min %.1f)   whole block %.1f us = %.2f us/step" % (K, np.median(enq) * 1e6, np.min(enq) * 1e6, np.median(tot) * 1e6, np.median(tot) * 1e6 / K), flush=True)
ens.close()
