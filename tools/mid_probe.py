import sys, time
import numpy as np
sys.path.insert(0, ".")
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params

def mk(N, D, target):
    ens = DeviceEnsemble(N, D)
    rs = np.random.RandomState(1)
    mu, cov, icov = dense_params(D)
    if target == "dense":
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
    else:
        ens.set_target(_lib.TARGET_ISO)
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 0.0, 0.0, 0.0)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(1, 0)
    ens.set_state(mu + rs.randn(N, D) @ np.linalg.cholesky(cov).T); ens.eval_state_log_prob()
    ens.set_tuning("small_kernel", 0)
    return ens

for target in ("dense", "iso"):
    for N in (4096, 16384):
        ens = mk(N, 64, target)
        ens.run(200, 1, False); ens.sync()
        for steps in (100, 1000, 4000, 100):
            ens.timer_start(); t0 = time.perf_counter()
            ens.run(steps, 1, False)
            ms = ens.timer_stop(); wall = time.perf_counter() - t0
            print("%-5s N=%-6d steps=%-5d gpu %.2f us/step  wall %.2f us/step" % (target, N, steps, ms * 1e3 / steps, wall * 1e6 / steps), flush=True)
        for thr in (0, 16, 256):
            ens.set_tuning("throttle", thr)
            ens.timer_start(); t0 = time.perf_counter()
            ens.run(4000, 1, False)
            ms = ens.timer_stop(); wall = time.perf_counter() - t0
            print("%-5s N=%-6d throttle=%-4d gpu %.2f us/step  wall %.2f us/step" % (target, N, thr, ms * 1e3 / 4000, wall * 1e6 / 4000), flush=True)
        ens.close()
