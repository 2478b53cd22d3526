import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params
N = 65536
for D in (128, 112):
    mu, cov, icov = dense_params(D)
    for move in (0, 1):
        ens = DeviceEnsemble(N, D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        ens.set_moves([_lib.MoveDesc(move, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
        ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(1, 0)
        ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
        ens.eval_state_log_prob()
        ens.run(100, 1, False); ens.sync()
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter(); ens.run(200, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
        print("%s N=%d D=%d move=%d: %.2f us/step (%.3f of 8 TB/s by 24D+17)" % (os.environ.get("EMX_LIB", "shipped")[-20:], N, D, move, best * 1e6 / 200, N * (24 * D + 17) / (best / 200) / 8e12), flush=True)
        ens.close()
