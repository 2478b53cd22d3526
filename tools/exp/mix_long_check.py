"""k_persist_mix against one-move launches and the per-half-step path over a LONG thinned, stored run (the quality test's shape):
first stored step at which the chains differ, acceptance fractions"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402
from tools.quick_bench import dense_params  # noqa: E402

N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 64
NST, THIN = int(sys.argv[3]) if len(sys.argv) > 3 else 600, int(sys.argv[4]) if len(sys.argv) > 4 else 4
BURN = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
mu, cov, icov = dense_params(D)
g0 = 2.38 / np.sqrt(2 * D)
out = []
for name, tune in (("mix", {"persist_mix": 1}), ("one-move", {"persist_mix": 0}), ("per-half-step", {"persist": 0})):
    ens = DeviceEnsemble(N, D)
    ens.set_target(_lib.TARGET_DENSE, mu, icov)
    ens.set_moves([_lib.MoveDesc(1, 2, 1, 0, 2.0, 1e-5, g0, 1.7), _lib.MoveDesc(2, 4, 1, 0, 2.0, 1e-5, g0, 1.7)], np.array([0.8, 0.2]))
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(12, 0)
    for k, v in tune.items():
        ens.set_tuning(k, v)
    ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
    ens.eval_state_log_prob()
    ens.run(BURN, 1, False)
    xb, lpb = ens.get_state()
    ens.chain_config(NST)
    ens.run(NST, THIN, True)
    chain = ens.chain_read(0, 0, NST)
    counts = ens.accepted_counts()
    print("%-14s status %d  acceptance %.4f  persist %r" % (name, ens.status(), counts.mean() / NST, ens.persist_info()), flush=True)
    out.append((xb, chain, counts))
    ens.close()
for k in (0, 1):
    a, b = out[k], out[2]
    print("%s vs per-half-step: burn-in state equal %s; " % (("mix", "one-move")[k], np.array_equal(a[0], b[0])), end="")
    d = np.nonzero((a[1] != b[1]).reshape(NST, -1).any(axis=1))[0]
    print("chains equal" if d.size == 0 else "first differing stored step %d (of %d), walkers differing there %d" % (d[0], NST, (a[1][d[0]] != b[1][d[0]]).any(axis=1).sum()),
          "; counts equal", np.array_equal(a[2], b[2]))
