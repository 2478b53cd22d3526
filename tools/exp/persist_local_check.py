"""The one-XCD form of the persistent kernel (k_persist<..., LOCAL>: tuning "persist_local") against the device-wide form and the
per-half-step launches: same bits, us/step.   usage: python tools/exp/persist_local_check.py [ndim] [sizes...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402
from tools.quick_bench import dense_params  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SIZES = [int(a) for a in sys.argv[2:]] or [512, 1024, 2048, 4096, 8192, 16384]
mu, cov, icov = dense_params(D)
for N in SIZES:
    outs = {}
    for name, tune in (("local", {"persist_local": 1}), ("device-wide", {"persist_local": 0}), ("per-half-step", {"persist": 0})):
        for store in (0, 1):
            ens = DeviceEnsemble(N, D)
            ens.set_target(_lib.TARGET_DENSE, mu, icov)
            ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(7, 0)
            for k, v in tune.items():
                ens.set_tuning(k, v)
            ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
            ens.eval_state_log_prob()
            if store:
                ens.chain_config(64)
            ens.run(64, 1, bool(store))
            x, lp = ens.get_state()
            rec = [x, lp, ens.accepted_mask().copy()] + ([ens.chain_read(0, 0, 64)] if store else [])
            if store:
                ens.chain_reset()
            best = 1e9
            K = 160
            if not store:
                for _ in range(7):
                    ens.sync()
                    t0 = time.perf_counter()
                    ens.run(K, 1, False)
                    ens.sync()
                    best = min(best, time.perf_counter() - t0)
            outs[(name, store)] = (rec, best * 1e6 / K, ens.persist_info(), ens.status())
            ens.close()
    ref = outs[("per-half-step", 0)][0], outs[("per-half-step", 1)][0]
    line = "N=%6d D=%d:" % (N, D)
    for name in ("local", "device-wide", "per-half-step"):
        same = all(np.array_equal(a, b) for st in (0, 1) for a, b in zip(outs[(name, st)][0], ref[st]))
        pi = outs[(name, 0)][2]
        line += "  %s %.2f us/step%s [launches %d local %d status %d]" % (name, outs[(name, 0)][1], "" if same else " DIFFERS", pi["launches"],
                                                                            pi["local_launches"], outs[(name, 0)][3])
    print(line, flush=True)
