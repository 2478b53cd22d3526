"""Exact mode on the one-XCD persistent kernels: k_plan_fetch keeping off the persistent launch's XCD (tuning fetch_avoid) against
not, alternating in one process; dense Gaussian ndim 64, stretch and DE 0.8 + snooker 0.2.  us/step of emx_run (400-step blocks)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params
D = 64
mu, cov, icov = dense_params(D)
for N in (1024, 4096, 8192):
    for name, moves, cdf in (("stretch", [_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], [1.0]),
                             ("de0.8+snooker0.2", [_lib.MoveDesc(1, 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * D), 1.7), _lib.MoveDesc(2, 4, 1, 0, 2.0, 1e-5, 0.2, 1.7)], [0.8, 1.0])):
        ens = DeviceEnsemble(N, D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        ens.set_moves(moves, np.array(cdf))
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(np.random.RandomState(5).get_state())
        ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
        ens.eval_state_log_prob()
        ens.run(200, 1, False); ens.sync()
        best = {0: 1e9, 1: 1e9}
        for _ in range(3):
            for v in (1, 0):
                ens.set_tuning("fetch_avoid", v)
                ens.run(100, 1, False); ens.sync()
                for _ in range(3):
                    t0 = time.perf_counter(); ens.run(400, 1, False); ens.sync(); best[v] = min(best[v], time.perf_counter() - t0)
        info = ens.persist_info()
        print("N=%d %s: fetch_avoid=1 %.2f us/step | fetch_avoid=0 %.2f us/step  (local launches %d of %d, status %d)" % (
            N, name, best[1] * 1e6 / 400, best[0] * 1e6 / 400, info["local_launches"], info["launches"], ens.status()), flush=True)
        ens.close()
