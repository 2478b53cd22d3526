"""The persistent kernel of the element-wise targets (csrc/emx_pvalu.hip, one-XCD form) against the per-half-step launches:
same bits, us/step.   usage: python tools/exp/persist_valu_check.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

CASES = [("iso", 10, 0), ("iso", 5, 0), ("rosen", 32, 0), ("diag", 64, 0), ("iso", 64, 1), ("rosen", 16, 2), ("box", 7, 0)]
for target, D, move in CASES:
    for N in (512, 2048, 8192):
        outs = {}
        for name, tune in (("persistent", {}), ("per-half-step", {"persist": 0})):
            for store in (0, 1):
                ens = DeviceEnsemble(N, D)
                rs = np.random.RandomState(D)
                if target == "iso":
                    ens.set_target(_lib.TARGET_ISO)
                elif target == "diag":
                    ens.set_target(_lib.TARGET_DIAG, rs.randn(D), 0.5 + rs.rand(D))
                elif target == "rosen":
                    ens.set_target(_lib.TARGET_ROSENBROCK, None, None, 20.0)
                else:
                    ens.set_target(_lib.TARGET_BOX)
                S = 4 if move == 2 else 2
                ens.set_moves([_lib.MoveDesc(move, S, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
                ens.set_rng_mode(_lib.RNG_PHILOX)
                ens.set_philox(11, 0)
                for k, v in tune.items():
                    ens.set_tuning(k, v)
                x0 = np.random.RandomState(1).rand(N, D) if target == "box" else np.random.RandomState(1).randn(N, D)
                ens.set_state(x0)
                ens.eval_state_log_prob()
                if store:
                    ens.chain_config(48)
                ens.run(48, 1, bool(store))
                x, lp = ens.get_state()
                rec = [x, lp, ens.accepted_mask().copy()] + ([ens.chain_read(0, 0, 48), ens.chain_read(1, 0, 48), ens.accepted_counts()] if store else [])
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
        same = all(np.array_equal(a, b) for st in (0, 1) for a, b in zip(outs[("persistent", st)][0], outs[("per-half-step", st)][0]))
        pi = outs[("persistent", 0)][2]
        print("%-5s D=%2d move=%d N=%5d: persistent %.2f us/step  per-half-step %.2f  %s [launches %d local %d recovered %d status %d]"
              % (target, D, move, N, outs[("persistent", 0)][1], outs[("per-half-step", 0)][1], "same bits" if same else "DIFFERS",
                 pi["launches"], pi["local_launches"], pi["recovered"], outs[("persistent", 0)][3]), flush=True)
