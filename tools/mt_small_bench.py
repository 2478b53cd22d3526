"""Exact (MT19937) mode per-step cost for small and large ensembles (host plan + upload + 2 launches)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

for N, D, steps in [(32, 5, 20000), (256, 16, 10000), (4096, 16, 3000), (65536, 64, 100)]:
    ens = DeviceEnsemble(N, D)
    ens.set_target(_lib.TARGET_ISO)
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 0.0, 0.0, 0.0)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(np.random.RandomState(5).get_state())
    ens.set_state(np.random.RandomState(1).randn(N, D))
    ens.eval_state_log_prob()
    ens.run(steps // 10, 1, False)
    ens.sync()
    t0 = time.perf_counter()
    ens.run(steps, 1, False)
    ens.sync()
    dt = (time.perf_counter() - t0) / steps
    print("MT19937 mode %6d x %-3d  %.2f us/step  %.0f steps/s  %.3e wu/s" % (N, D, dt * 1e6, 1 / dt, N / dt), flush=True)
    ens.close()
