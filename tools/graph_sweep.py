"""hipGraph replay of the native 8-step block vs plain launches across ensemble sizes (general path)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402
from tools.quick_bench import dense_params  # noqa: E402


def run(N, D, target, graph, steps):
    ens = DeviceEnsemble(N, D)
    rs = np.random.RandomState(1)
    if target == "dense":
        mu, cov, icov = dense_params(D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        p0 = mu + rs.randn(N, D) @ np.linalg.cholesky(cov).T
    else:
        ens.set_target(_lib.TARGET_ISO)
        p0 = rs.randn(N, D)
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 0.0, 0.0, 0.0)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(12345, 0)
    ens.set_state(p0)
    ens.eval_state_log_prob()
    ens.set_tuning("graph", graph)
    ens.set_tuning("small_kernel", 0)
    ens.run(steps // 4, 1, False)
    ens.sync()
    t0 = time.perf_counter()
    ens.run(steps, 1, False)
    ens.sync()
    dt = (time.perf_counter() - t0) / steps
    ens.close()
    return dt * 1e6


for N, D, target in [(512, 16, "iso"), (2048, 64, "dense"), (4096, 64, "dense"), (4096, 64, "iso"), (16384, 64, "dense"), (16384, 16, "iso")]:
    a, b = run(N, D, target, 0, 4000), run(N, D, target, 1, 4000)
    print("%6d x %-3d %-5s plain %.2f us/step | graph %.2f us/step" % (N, D, target, a, b), flush=True)
