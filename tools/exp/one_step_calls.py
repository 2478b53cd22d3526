import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
N, D = 1024, 5
for pe in (1, 0, 1):
    ens = DeviceEnsemble(N, D)
    ens.set_target(_lib.TARGET_ISO)
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
    ens.set_tuning("persist_exact", pe)
    ens.set_state(np.random.RandomState(1).randn(N, D)); ens.eval_state_log_prob()
    ens.run(50, 1, False); ens.sync()
    i0 = ens.persist_info()
    t0 = time.perf_counter()
    for _ in range(1000):
        ens.run(1, 1, False)
    ens.sync()
    t1 = time.perf_counter()
    print("pe=%d: %.1f us per 1-step call; persist before %r after %r" % (pe, (t1 - t0) * 1e3, i0["launches"], ens.persist_info()["launches"]), flush=True)
    ens.close()
