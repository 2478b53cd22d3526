"""element-wise targets at low ndim (rows of 4 lanes): the one-XCD persistent kernel against the per-half-step launches, Philox and exact"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
for N, D in ((1024, 2), (1024, 4), (1024, 8), (4096, 3), (4096, 8), (8192, 4)):
    out = []
    for rng in ("philox", "mt19937"):
        for persist in (1, 0):
            ens = DeviceEnsemble(N, D)
            ens.set_target(_lib.TARGET_ISO)
            ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
            if rng == "philox":
                ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(11, 0)
            else:
                ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
            ens.set_tuning("persist", persist)
            ens.set_tuning("small_kernel", 0)
            ens.set_state(np.random.RandomState(1).randn(N, D)); ens.eval_state_log_prob()
            ens.run(100, 1, False); ens.sync()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); ens.run(400, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
            out.append("%s persist=%d %.2f" % (rng, persist, best * 1e6 / 400))
            ens.close()
    print("N=%d D=%d: %s" % (N, D, "   ".join(out)), flush=True)
