"""long runs of the round-4 paths against the per-half-step path: final state, accept counters and generator state must be equal
(exact mode on the persistent kernels; DE + snooker mixtures in shared launches)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params

def run(N, D, tgt, moves, w, rng, steps, tune):
    ens = DeviceEnsemble(N, D)
    if tgt == "dense":
        mu, cov, icov = dense_params(D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
    else:
        ens.set_target(_lib.TARGET_ISO)
        p0 = np.random.RandomState(1).randn(N, D)
    ens.set_moves(moves, np.array(w))
    if rng == "mt":
        ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
    else:
        ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(3, 0)
    for k, v in tune.items():
        ens.set_tuning(k, v)
    ens.set_state(p0); ens.eval_state_log_prob()
    t0 = time.perf_counter()
    left = steps
    while left > 0:                      # calls of different lengths: launches of every size, pipeline carried across calls
        k = min(left, int(np.random.RandomState(left).randint(1, 700)))
        ens.run(k, 1, False)
        left -= k
    ens.sync()
    dt = time.perf_counter() - t0
    x, lp = ens.get_state()
    out = (x, lp, ens.get_mt19937()[1] if rng == "mt" else None, ens.status(), ens.persist_info(), dt)
    ens.close()
    return out

g0 = 2.38 / np.sqrt(128)
ST = [_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)]
MIX = [_lib.MoveDesc(1, 2, 1, 0, 2.0, 1e-5, g0, 0.0), _lib.MoveDesc(2, 4, 1, 0, 2.0, 1e-5, g0, 1.4)]
cases = [(1024, 64, "dense", ST, [1.0], "mt", 20000), (4096, 16, "iso", ST, [1.0], "mt", 12000), (8192, 8, "iso", ST, [1.0], "mt", 6000),
         (512, 5, "iso", ST, [1.0], "mt", 20000), (2048, 4, "iso", [_lib.MoveDesc(2, 4, 1, 0, 2.0, 1e-5, 0.3, 1.7)], [1.0], "mt", 8000),
         (65536, 64, "dense", MIX, [0.8, 0.2], "philox", 3000), (2048, 64, "dense", MIX, [0.5, 0.5], "philox", 12000), (1024, 32, "dense", MIX, [0.3, 0.7], "philox", 12000),
         # round 5: exact mode with move mixtures on the persistent kernels; exact mode at the headline size (device finish, an upload per step)
         (1024, 64, "dense", MIX, [0.8, 0.2], "mt", 6000), (4096, 64, "dense", ST + MIX[:1], [0.5, 0.5], "mt", 4000), (2048, 10, "iso", ST + MIX, [0.4, 0.4, 0.2], "mt", 4000),
         (65536, 64, "dense", ST, [1.0], "mt-finish", 1500)]
for N, D, tgt, mv, w, rng, steps in cases:
    if rng == "mt-finish":               # device finish (k_plan_raw) against the finisher threads' own conversions
        rng = "mt"
        a = run(N, D, tgt, mv, w, rng, steps, {})
        b = run(N, D, tgt, mv, w, rng, steps, {"mt_device_finish": 0})
    else:
        a = run(N, D, tgt, mv, w, rng, steps, {})
        b = run(N, D, tgt, mv, w, rng, steps, {"persist": 0})
    ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and (a[2] is None or np.array_equal(a[2], b[2])) and a[3] == 0 and b[3] == 0
    print("%6d x %2d %-5s %-6s %5d steps: %s  (persistent %d launches, %d recovered, %.2f s; per-half-step %.2f s)" % (
        N, D, tgt, rng, steps, "EQUAL" if ok else "DIFFERENT  status %r %r" % (a[3], b[3]), a[4]["launches"], a[4]["recovered"], a[5], b[5]), flush=True)
