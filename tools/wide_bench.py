"""Wide dense Gaussian targets (ndim > 112, emx_wide.hip): time per step and f64 MFMA rate of the log-prob kernel.

  python tools/wide_bench.py [--steps 50]          (GPU box)

flop per walker-update = Dp (Dp + 16) -- the lower-triangular 16x16 blocks of L that k_wide_lp multiplies, 2 flop per MAC
(the algorithmic D^2 is the same thing without the padding).  Peak: 78.6 TFLOP/s f64 matrix (MI355X_MICROARCH.md)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from emcee_amd import _lib                      # noqa: E402
from emcee_amd.device import DeviceEnsemble    # noqa: E402


def dense_gaussian(D, seed=0):
    rs = np.random.RandomState(seed)
    mu = rs.randn(D)
    A = rs.randn(D, D)
    cov = A @ A.T / D + 0.1 * np.eye(D)
    icov = np.linalg.inv(cov)
    return mu, cov, 0.5 * (icov + icov.T)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--configs", default="65536x128,65536x256,65536x512,16384x1024,4096x512,262144x128")
    a = ap.parse_args()
    for cfg in a.configs.split(","):
        N, D = (int(v) for v in cfg.split("x"))
        mu, cov, icov = dense_gaussian(D)
        p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
        ens = DeviceEnsemble(N, D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        ens.set_state(p0)
        ens.eval_state_log_prob()
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(7, 0)
        ens.run(5, 1, False)
        ens.sync()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            ens.run(a.steps, 1, False)
            ens.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        Dp = (D + 15) // 16 * 16
        us = best / a.steps * 1e6
        flop = N * Dp * (Dp + 16)
        byts = N * (24 * D + 17)
        print("%7d x %4d: %9.1f us/step  %6.2f e9 wu/s  log-prob MFMA work %5.1f TFLOP/s of the WHOLE step (%.1f %% of 78.6)  "
              "algorithmic HBM %.2f TB/s  acc %.3f" % (N, D, us, N / us / 1e3, flop / us / 1e6, flop / us / 1e6 / 78.6 * 100,
                                                      byts / us / 1e6, float(np.mean(ens.accepted_mask()))), flush=True)
        assert ens.status() == 0
        ens.close()


if __name__ == "__main__":
    main()
