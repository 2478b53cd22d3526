"""exact mode, host pipeline: mid sizes (persistent kernels) and big ones (per-half-step path) in one go; us/step, best of 5"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
out = []
for N, D, K in ((1024, 64, 400), (4096, 64, 400), (16384, 64, 300), (65536, 64, 200), (65536, 64, 200)):
    ens = DeviceEnsemble(N, D)
    ens.set_target(_lib.TARGET_ISO)
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
    ens.set_tuning("mt_device", 0)
    ens.set_state(np.random.RandomState(1).randn(N, D)); ens.eval_state_log_prob()
    ens.run(100, 1, False); ens.sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); ens.run(K, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
    st = ens.pipeline_stats()
    out.append("%dx%d %.1f (gen %.1f)" % (N, D, best * 1e6 / K, st["generator_us"]))
    ens.close()
print("spin=%s cpus=%d(%d usable): %s" % (os.environ.get("EMX_PIPE_SPIN_US", "default"), os.cpu_count(), len(os.sched_getaffinity(0)), "  ".join(out)), flush=True)
