"""Exact (MT19937) mode with a move mixture at mid sizes: us/step on the persistent kernels (round 5) against an upload per step
(tuning persist_exact_mix = 0), dense Gaussian target, DEMove 0.8 + DESnookerMove 0.2 and StretchMove 0.5 + DEMove 0.5."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params
for N in (1024, 4096, 16384, 65536):
    D = 64
    mu, cov, icov = dense_params(D)
    for name, moves, cdf in (("de0.8+snooker0.2", [_lib.MoveDesc(1, 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * D), 1.7), _lib.MoveDesc(2, 4, 1, 0, 2.0, 1e-5, 0.2, 1.7)], [0.8, 1.0]),
                             ("stretch0.5+de0.5", [_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7), _lib.MoveDesc(1, 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * D), 1.7)], [0.5, 1.0])):
        res = {}
        for mix in (1, 0):
            ens = DeviceEnsemble(N, D)
            ens.set_target(_lib.TARGET_DENSE, mu, icov)
            ens.set_moves(moves, np.array(cdf))
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(np.random.RandomState(5).get_state())
            ens.set_tuning("persist_exact_mix", mix)
            ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
            ens.eval_state_log_prob()
            ens.run(200, 1, False); ens.sync()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); ens.run(400, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
            st = ens.pipeline_stats() or {}
            res[mix] = (best * 1e6 / 400, ens.persist_info()["launches"], ens.status(), st)
            ens.close()
        print("N=%d D=%d %s: persistent %.2f us/step (%d launches, status %d) | an upload per step %.2f us/step (%d launches)" % (
            N, D, name, res[1][0], res[1][1], res[1][2], res[0][0], res[0][1]), flush=True)
        print("    pipeline us per produced step (persistent run): " + ", ".join("%s %.1f" % (k.replace("_us", ""), v) for k, v in res[1][3].items() if k.endswith("_us") or k.endswith("_summed")), flush=True)
