#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04m
O=$PWD/gpurun_out/r04m
timeout 900 python -m pytest tests/test_gpu_wide_dense.py -q -x -p no:cacheprovider > $O/wide_tests.log 2>&1; echo "wide tests rc=$?" | tee -a $O/summary.txt
tail -n 12 $O/wide_tests.log
for D in 128 112 96 80; do
 for slab in 1 0; do
  EMX_TUNE="slab=$slab" timeout 300 python - <<PY 2>&1 | tee -a $O/slab_ab.txt
import sys, time
sys.path.insert(0, ".")
import numpy as np
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params
N, D = 65536, $D
mu, cov, icov = dense_params(D)
for move in (0, 1):
    ens = DeviceEnsemble(N, D)
    ens.set_target(_lib.TARGET_DENSE, mu, icov)
    ens.set_moves([_lib.MoveDesc(move, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(1, 0)
    ens.set_state(mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T)
    ens.eval_state_log_prob()
    ens.run(100, 1, False); ens.sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); ens.run(200, 1, False); ens.sync(); best = min(best, time.perf_counter() - t0)
    acc = ens.accepted_mask().mean()
    bytes_per = 24 * D + 17
    print("slab=$slab N=%d D=%d move=%d: %.2f us/step  (%.3f of 8 TB/s by 24D+17; last accept %.3f) status %d" % (N, D, move, best * 1e6 / 200, N * bytes_per / (best / 200) / 8e12, acc, ens.status()))
    ens.close()
PY
 done
done
