"""exact mode, mid sizes: us/step (best of 5 x 400 steps); env EMX_PIPE_SPIN_US = the stage threads' yield window"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
out = []
for N, D in ((1024, 5), (1024, 64), (4096, 64), (8192, 64), (16384, 64), (65536, 64)):
    ens = DeviceEnsemble(N, D)
    ens.set_target(_lib.TARGET_ISO)
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
    if os.environ.get('PES'): ens.set_tuning('persist_exact_steps', int(os.environ['PES']))
    ens.set_state(np.random.RandomState(1).randn(N, D)); ens.eval_state_log_prob()
    ens.run(100, 1, False); ens.sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); ens.run(400, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
    out.append("%dx%d %.2f" % (N, D, best * 1e6 / 400))
    ens.close()
print("steps/launch=%s :" % os.environ.get("PES", "default"), "  ".join(out), flush=True)
