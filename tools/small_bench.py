"""Throughput of small ensembles: whole-run-in-one-workgroup kernel (k_small_run) vs the general launch-per-half-step path."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402


def run(N, D, target, small, steps):
    ens = DeviceEnsemble(N, D)
    rs = np.random.RandomState(1)
    if target == "iso":
        ens.set_target(_lib.TARGET_ISO)
        p0 = rs.randn(N, D)
    elif target == "diag":
        iv = 1.0 / (0.1 + rs.rand(D))
        ens.set_target(_lib.TARGET_DIAG, np.zeros(D), iv)
        p0 = rs.randn(N, D) / np.sqrt(iv)
    elif target == "dense":
        A = rs.randn(D, D)
        cov = A @ A.T / D + 0.1 * np.eye(D)
        icov = np.linalg.inv(cov)
        ens.set_target(_lib.TARGET_DENSE, np.zeros(D), 0.5 * (icov + icov.T))
        p0 = rs.randn(N, D) @ np.linalg.cholesky(cov).T
    else:
        ens.set_target(_lib.TARGET_ROSENBROCK, scale=20.0)
        p0 = 1 + 0.1 * rs.randn(N, D)
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 0.0, 0.0, 0.0)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(12345, 0)
    ens.set_state(p0)
    ens.eval_state_log_prob()
    ens.set_tuning("small_kernel", small)
    ens.run(steps // 10, 1, False)
    ens.sync()
    t0 = time.perf_counter()
    ens.run(steps, 1, False)
    ens.sync()
    dt = time.perf_counter() - t0
    ens.close()
    return dt / steps * 1e6


if __name__ == "__main__":
    out = []
    for N, D, target in [(32, 5, "iso"), (32, 5, "dense"), (128, 64, "dense"), (512, 16, "dense"), (64, 8, "rosen"), (128, 16, "diag"), (256, 32, "iso"), (1024, 8, "iso"),
                         (1024, 16, "rosen"), (4096, 2, "iso"), (2048, 4, "diag")]:
        steps = 40000 if N <= 256 else 8000
        fast, slow = run(N, D, target, 1, steps), run(N, D, target, 0, steps)
        r = dict(N=N, D=D, target=target, small_us_per_step=fast, general_us_per_step=slow, speedup=slow / fast,
                 steps_per_s=1e6 / fast, wu_per_s=N * 1e6 / fast)
        print(json.dumps(r), flush=True)
        out.append(r)
    json.dump(out, open("gpurun_out/small_bench.json", "w"), indent=1)
