"""Slab kernel A/B on one box: us/step at 65 536 walkers, padded ndim 128 / 112, stretch and DE, for the values of one tuning key
(default: slab_skew 0 / 1), alternating.  usage: python tools/exp/slab_ab.py [key] [rounds] [values, e.g. 0,1,2,3,4]"""
import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params
key = sys.argv[1] if len(sys.argv) > 1 else "slab_skew"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
values = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1]
N = 65536
for D in (128, 112):
    mu, cov, icov = dense_params(D)
    for move in (0, 1):
        ens = DeviceEnsemble(N, D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        ens.set_moves([_lib.MoveDesc(move, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
        ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(1, 0)
        ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
        ens.eval_state_log_prob()
        ens.run(100, 1, False); ens.sync()
        best = {v: 1e9 for v in values}
        for _ in range(rounds):
            for v in values:
                ens.set_tuning(key, v)
                ens.run(50, 1, False); ens.sync()
                for _ in range(3):
                    t0 = time.perf_counter(); ens.run(200, 1, False); ens.sync(); best[v] = min(best[v], time.perf_counter() - t0)
        nb = (24 if move == 0 else 32) * D + 17
        print("N=%d D=%d move=%d: " % (N, D, move) + " | ".join("%s=%d %.2f us/step (%.3f of 8 TB/s)" % (
            key, v, best[v] * 1e6 / 200, N * nb / (best[v] / 200) / 8e12) for v in values), flush=True)
        ens.close()
