"""exact mode on the per-half-step path (persist_exact = 0; an upload per step, events) at mid sizes: us/step in 200-step blocks as bench.py
times them; env EMX_PIPE_SPIN_US = the stage threads' spin window"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params
mu, cov, icov = dense_params(64)
out = []
for N, pe in ((1024, 0), (4096, 0), (4096, 1), (65536, 0)):
    ens = DeviceEnsemble(N, 64)
    ens.set_target(_lib.TARGET_DENSE, mu, icov)
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
    ens.set_tuning("mt_device", 0)
    ens.set_tuning("persist_exact", pe)
    ens.set_state(mu + np.random.RandomState(1).randn(N, 64) @ np.linalg.cholesky(cov).T); ens.eval_state_log_prob()
    ens.run(50, 1, False); ens.sync()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); ens.run(200, 1, False); ens.sync(); ts.append((time.perf_counter() - t0) * 1e6 / 200)
    out.append("%dx64 pe=%d median %.1f (min %.1f max %.1f)" % (N, pe, np.median(ts), min(ts), max(ts)))
    ens.close()
print("spin=%s usable cpus %d: %s" % (os.environ.get("EMX_PIPE_SPIN_US", "default"), len(os.sched_getaffinity(0)), "   ".join(out)), flush=True)
