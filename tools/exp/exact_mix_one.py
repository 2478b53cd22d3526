"""One exact-mode (or Philox) mixture run for a kernel trace: usage exact_mix_one.py N rng(mt|philox) steps"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params
N = int(sys.argv[1]); rng = sys.argv[2]; steps = int(sys.argv[3]); D = 64
mu, cov, icov = dense_params(D)
moves = [_lib.MoveDesc(1, 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * D), 1.7), _lib.MoveDesc(2, 4, 1, 0, 2.0, 1e-5, 0.2, 1.7)]
ens = DeviceEnsemble(N, D)
ens.set_target(_lib.TARGET_DENSE, mu, icov)
ens.set_moves(moves, np.array([0.8, 1.0]))
if rng == "mt":
    ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
else:
    ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(1, 0)
ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
ens.eval_state_log_prob()
ens.run(200, 1, False); ens.sync()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); ens.run(steps, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
print("N=%d %s: %.2f us/step; %s; %s" % (N, rng, best * 1e6 / steps, ens.persist_info(), ens.pipeline_stats()), flush=True)
ens.close()
