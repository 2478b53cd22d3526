"""k_persist_mix against the one-move instantiations: a DE + snooker schedule on the dense target with weights (1, 0), (0, 1), (0.8, 0.2),
(0.5, 0.5); us/step of emx_run(160 steps), best of 5.   usage: python tools/exp/mix_probe.py [nwalkers] [ndim]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402
from tools.quick_bench import dense_params  # noqa: E402

N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 64
mu, cov, icov = dense_params(D)
g0 = 2.38 / np.sqrt(2 * D)
for w in ([1.0, 0.0], [0.0, 1.0], [0.8, 0.2], [0.5, 0.5]):
    out = []
    for mix in (1, 0):
        ens = DeviceEnsemble(N, D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        ens.set_moves([_lib.MoveDesc(1, 2, 1, 0, 2.0, 1e-5, g0, 1.7), _lib.MoveDesc(2, 4, 1, 0, 2.0, 1e-5, g0, 1.7)], np.array(w))
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(7, 0)
        ens.set_tuning("persist_mix", mix)
        ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
        ens.eval_state_log_prob()
        ens.run(64, 1, False)
        ens.sync()
        h0 = ens.persist_info()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            ens.run(160, 1, False)
            ens.sync()
            best = min(best, time.perf_counter() - t0)
        h1 = ens.persist_info()
        hs = (h1["halfsteps"] - h0["halfsteps"]) / 5
        out.append("mix=%d %6.2f us/step = %5.2f us/half-step, %4.1f half-steps a launch" % (
            mix, best * 1e6 / 160, best * 1e6 / hs, (h1["halfsteps"] - h0["halfsteps"]) / max(1, h1["launches"] - h0["launches"])))
        ens.close()
    print("N=%d D=%d weights %s:  %s" % (N, D, w, "   |   ".join(out)), flush=True)
