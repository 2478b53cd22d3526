"""Persistent half-steps (k_persist; tuning "persist"): step time against the launch-per-half-step path, per ensemble size.
  usage: python tools/exp/persist_check.py [nwalkers ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

import os
KEY = os.environ.get("PERSIST_CFG", "c2")          # bench.Workload key: c2 (dense 64) or c3 (Rosenbrock 32)
sizes = [int(a) for a in sys.argv[1:]] or [65536, 32768, 16384, 4096, 2048]


def run(ens, n, store):
    if store:
        ens.chain_reset()
    ens.run(n, 1, store)
    ens.sync()


for N in sizes:
    for store in (False, True):
        row = []
        for persist in (0, 1):
            wl = bench.Workload(KEY, N)
            if os.environ.get("PERSIST_D"):        # the dense target at another even ndim <= 64
                from emcee_amd import _lib
                Dn = int(os.environ["PERSIST_D"])
                mu, cov, icov = bench.dense_gaussian(Dn)
                wl.D = Dn
                wl.target = (_lib.TARGET_DENSE, mu, icov, 0.0)
                wl.p0 = mu + np.random.RandomState(1).randn(N, Dn) @ np.linalg.cholesky(cov).T
                wl.moves = [("stretch", _lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * Dn), 1.7))]
            ens = DeviceEnsemble(wl.N, wl.D, device=0)
            wl.install(ens, "philox")
            ens.set_tuning("persist", 3 * persist)
            if store:
                ens.chain_config(160)
            t_end = time.perf_counter() + 0.15
            while time.perf_counter() < t_end:
                run(ens, 20, store)
            ts = []
            for _ in range(40):
                t0 = time.perf_counter()
                run(ens, 20, store)
                ts.append((time.perf_counter() - t0) / 20)
            ts2 = []
            for _ in range(20):
                t0 = time.perf_counter()
                run(ens, 160, store)
                ts2.append((time.perf_counter() - t0) / 160)
            row.append((np.median(ts) * 1e6, np.median(ts2) * 1e6, ens.status(), ens.persist_info()["launches"]))
            ens.close()
        (a20, a160, s0, l0), (b20, b160, s1, l1) = row
        print(KEY + os.environ.get("PERSIST_D", "") + " N=%6d store=%d   K=20: %.2f -> %.2f us/step (%+.1f %%)   K=160: %.2f -> %.2f (%+.1f %%)   status %d/%d  launches %d/%d" % (
            N, store, a20, b20, (b20 / a20 - 1) * 100, a160, b160, (b160 / a160 - 1) * 100, s0, s1, l0, l1), flush=True)
