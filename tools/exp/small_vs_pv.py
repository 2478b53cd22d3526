import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
for N, D in ((256, 5), (512, 5), (512, 10), (1024, 5), (512, 16), (256, 16)):
    for small in (1, 0):
        ens = DeviceEnsemble(N, D)
        ens.set_target(_lib.TARGET_ISO)
        ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
        ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(11, 0)
        ens.set_tuning("small_kernel", small)
        ens.set_state(np.random.RandomState(1).randn(N, D)); ens.eval_state_log_prob()
        ens.run(200, 1, False); ens.sync()
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter(); ens.run(800, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
        print("N=%d D=%d small_kernel=%d: %.2f us/step  %r" % (N, D, small, best * 1e6 / 800, ens.persist_info()), flush=True)
        ens.close()
