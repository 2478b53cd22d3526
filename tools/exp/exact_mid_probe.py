"""exact (MT19937) mode at mid-size ensembles: us/step through emx_run, against Philox mode"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
for N, D in ((256, 5), (1024, 5), (1024, 64), (4096, 16), (4096, 64), (8192, 64), (16384, 64)):
    for rng in ("mt19937", "philox"):
        ens = DeviceEnsemble(N, D)
        ens.set_target(_lib.TARGET_ISO)
        ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
        if rng == "philox":
            ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(11, 0)
        else:
            ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
        ens.set_state(np.random.RandomState(1).randn(N, D)); ens.eval_state_log_prob()
        ens.run(100, 1, False); ens.sync()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); ens.run(400, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
        print("N=%5d D=%2d %-8s: %7.2f us/step  persist %r" % (N, D, rng, best * 1e6 / 400, ens.persist_info()["launches"]), flush=True)
        ens.close()
print("--- pipeline stage times (us per step) at mid sizes, exact mode")
for N, D in ((1024, 64), (4096, 64), (8192, 64)):
    ens = DeviceEnsemble(N, D)
    ens.set_target(_lib.TARGET_ISO)
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
    ens.set_state(np.random.RandomState(1).randn(N, D)); ens.eval_state_log_prob()
    ens.run(2000, 1, False); ens.sync()
    print(N, D, ens.pipeline_stats(), flush=True)
    import cProfile
    ens.close()
