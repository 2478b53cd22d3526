"""exact mode, element-wise target (iso Gaussian), stretch: the device-wide form of k_persist_valu on the host pipeline's plans against the
per-half-step launches; us/step, best of 5 x 300 steps"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
for N, D in ((16384, 64), (16384, 5), (32768, 16), (8192, 64)):
    out = []
    for pe in (1, 0):
        ens = DeviceEnsemble(N, D)
        ens.set_target(_lib.TARGET_ISO)
        ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
        ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
        ens.set_tuning("mt_device", 0)
        ens.set_tuning("persist_exact", pe)
        ens.set_state(np.random.RandomState(1).randn(N, D)); ens.eval_state_log_prob()
        ens.run(100, 1, False); ens.sync()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); ens.run(300, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
        info = ens.persist_info()
        out.append("persist_exact=%d %.1f us/step (launches %d, local %d)" % (pe, best * 1e6 / 300, info["launches"], info["local_launches"]))
        ens.close()
    print("N=%6d D=%2d: %s" % (N, D, "   ".join(out)), flush=True)
