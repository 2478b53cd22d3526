"""The Gaussian Metropolis move on the fused dense target: per-step launches against k_persist_gauss (walkers in registers), us/step.
  usage: python tools/exp/persist_gauss_check.py [nwalkers ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [65536, 8192]
for N in sizes:
    for store in (False, True):
        row = []
        for persist in (0, 1):
            wl = bench.Workload("c2", N)
            ens = DeviceEnsemble(wl.N, wl.D, device=0)
            wl.install(ens, "philox")
            md = _lib.MoveDesc(3, 1, 1, 0, 0.0, 0.05, 0.0, 0.0)      # Gaussian move, vector mode, isotropic sigma
            ens.set_moves([md], np.array([1.0]))
            ens.set_tuning("persist", persist)
            if store:
                ens.chain_config(160)
            t_end = time.perf_counter() + 0.15
            while time.perf_counter() < t_end:
                if store:
                    ens.chain_reset()
                ens.run(20, 1, store)
                ens.sync()
            ts = []
            for _ in range(20):
                if store:
                    ens.chain_reset()
                t0 = time.perf_counter()
                ens.run(160, 1, store)
                ens.sync()
                ts.append((time.perf_counter() - t0) / 160)
            row.append((np.median(ts) * 1e6, ens.status(), ens.persist_info()["launches"], float(ens.accepted_mask().mean())))
            ens.close()
        (a, s0, l0, f0), (b, s1, l1, f1) = row
        print("gauss N=%6d store=%d  K=160: %.2f -> %.2f us/step (%+.1f %%)  status %d/%d launches %d/%d accept %.3f/%.3f" % (
            N, store, a, b, (b / a - 1) * 100, s0, s1, l0, l1, f0, f1), flush=True)
