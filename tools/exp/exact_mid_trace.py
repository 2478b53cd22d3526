"""exact mode at a mid size: host-side timeline of the persistent launches (EMX_TRACE_PERSIST=1)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["EMX_TRACE_PERSIST"] = "1"
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
N, D = int(sys.argv[1]), 64
ens = DeviceEnsemble(N, D)
ens.set_target(_lib.TARGET_ISO)
ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
ens.set_state(np.random.RandomState(1).randn(N, D)); ens.eval_state_log_prob()
t0 = time.perf_counter(); ens.run(400, 1, False); ens.sync(); print("400 steps: %.1f us/step" % ((time.perf_counter() - t0) * 1e6 / 400))
ens.close()
