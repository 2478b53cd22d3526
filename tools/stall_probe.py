"""Wall-vs-GPU time of long native runs: looks for host-side stalls of the launch loop."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params

def probe(move, throttle, steps=2000, N=65536, D=64):
    ens = DeviceEnsemble(N, D)
    mu, cov, icov = dense_params(D)
    ens.set_target(_lib.TARGET_DENSE, mu, icov)
    rs = np.random.RandomState(1)
    p0 = mu + rs.randn(N, D) @ np.linalg.cholesky(cov).T
    ens.set_moves([_lib.MoveDesc(move, 4 if move == 2 else 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(3, 0)
    ens.set_state(p0); ens.eval_state_log_prob()
    ens.set_tuning("throttle", throttle)
    ens.run(20, 1, False); ens.sync()
    out = []
    for rep in range(4):
        ens.timer_start(); t0 = time.perf_counter()
        ens.run(steps, 1, False)
        ms = ens.timer_stop(); wall = (time.perf_counter() - t0) * 1e3
        out.append("%.1f/%.1f" % (ms / steps * 1e3, wall / steps * 1e3))
    ens.close()
    return out

if __name__ == "__main__":
    for move in (0, 2):
        for thr in (0, 64, 16):
            print("move=%d throttle=%-3d gpu/wall us per step:" % (move, thr), "  ".join(probe(move, thr)), flush=True)
