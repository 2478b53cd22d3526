"""Workgroup size (waves per block) and blocks per CU of the dense-target half-step kernel, per move, at 65536 x 64."""
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
import quick_bench as qb
def run(move, wpb, bpc, N=65536, D=64, steps=300):
    ens = DeviceEnsemble(N, D)
    rs = np.random.RandomState(1)
    mu, cov, icov = qb.dense_params(D)
    ens.set_target(_lib.TARGET_DENSE, mu, icov)
    p0 = mu + rs.randn(N, D) @ np.linalg.cholesky(cov).T
    md = _lib.MoveDesc(3, 1, 0, 0, 0.0, 0.03, 0.0, 0.0) if move == 3 else _lib.MoveDesc(move, 4 if move == 2 else 2, 1, 0, 2.0, 1e-5, 2.38/np.sqrt(2*D), 1.7)
    ens.set_moves([md], np.array([1.0])); ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(1, 0)
    ens.set_state(p0); ens.eval_state_log_prob()
    ens.set_tuning("waves_per_block", wpb); ens.set_tuning("blocks_per_cu", bpc)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        ens.run(50, 1, False); ens.sync()
    out = []
    for _ in range(7):
        ens.sync(); ens.timer_start(); ens.run(steps, 1, False); out.append(ens.timer_stop() / steps * 1e3)
    ens.close()
    return float(np.median(out))
for move in (3, 0, 1, 2):
    for wpb, bpc in ((0, 2), (8, 2), (4, 2), (4, 4), (2, 4)):
        print("move", move, "wpb", wpb, "bpc", bpc, "%.2f us/step" % run(move, wpb, bpc), flush=True)
