"""exact mode, dense Gaussian ndim 64, stretch: the persistent kernels on the host pipeline's plans (persist_exact = 1) against the
per-half-step launches with an upload per step; us/step, best of 5 x 300 steps"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params
mu, cov, icov = dense_params(64)
for N in (4096, 8192, 16384, 32768, 65536):
    out = []
    for pe, mx in ((1, 65536), (0, 0)):
        ens = DeviceEnsemble(N, 64)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
        ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
        ens.set_tuning("mt_device", 0)
        ens.set_tuning("persist_exact", pe)
        if pe: ens.set_tuning("persist_exact_max_walkers", mx)
        ens.set_state(mu + np.random.RandomState(1).randn(N, 64) @ np.linalg.cholesky(cov).T); ens.eval_state_log_prob()
        ens.run(100, 1, False); ens.sync()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); ens.run(300, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
        st = ens.pipeline_stats()
        out.append("persist_exact=%d %.1f us/step (gen %.1f, launches %d)" % (pe, best * 1e6 / 300, st["generator_us"], ens.persist_info()["launches"]))
        ens.close()
    print("N=%6d: %s" % (N, "   ".join(out)), flush=True)
