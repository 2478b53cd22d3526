"""Exact mode on the DEVICE-WIDE persistent kernels (8 192 < walkers <= 32 768): k_plan_fetch as few workgroups (tuning fetch_blocks)
against one per piece of 256 entries, alternating in one process, and the Philox rate of the shape.  us/step of emx_run."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params
D = 64
mu, cov, icov = dense_params(D)
vals = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["64", "0"])]
for N in (8192, 16384, 32768):
    for name, moves, cdf in (("stretch", [_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], [1.0]),
                             ("de0.8+snooker0.2", [_lib.MoveDesc(1, 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * D), 1.7), _lib.MoveDesc(2, 4, 1, 0, 2.0, 1e-5, 0.2, 1.7)], [0.8, 1.0])):
        out = {}
        for rng in ("mt", "philox"):
            ens = DeviceEnsemble(N, D)
            ens.set_target(_lib.TARGET_DENSE, mu, icov)
            ens.set_moves(moves, np.array(cdf))
            if rng == "mt":
                ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
            else:
                ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(1, 0)
            ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
            ens.eval_state_log_prob()
            ens.run(200, 1, False); ens.sync()
            best = {v: 1e9 for v in (vals if rng == "mt" else [0])}
            for _ in range(2):
                for v in best:
                    if rng == "mt":
                        ens.set_tuning("fetch_blocks", v)
                    ens.run(100, 1, False); ens.sync()
                    for _ in range(3):
                        t0 = time.perf_counter(); ens.run(400, 1, False); ens.sync(); best[v] = min(best[v], time.perf_counter() - t0)
            info = ens.persist_info()
            out[rng] = (best, info, ens.status())
            ens.close()
        print("N=%d %s: exact " % (N, name) + " | ".join("fetch_blocks=%d %.2f us/step" % (v, t * 1e6 / 400) for v, t in out["mt"][0].items())
              + " | philox %.2f us/step  (launches %d, local %d, status %d)" % (out["philox"][0][0] * 1e6 / 400, out["mt"][1]["launches"], out["mt"][1]["local_launches"], out["mt"][2]), flush=True)
