"""Probe: D2H bandwidth of emx_chain_read (get_chain) for a ~2 GB device chain."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

N, D, K = 65536, 64, 64
ens = DeviceEnsemble(N, D)
ens.set_target(_lib.TARGET_ISO)
ens.set_rng_mode(_lib.RNG_PHILOX)
ens.set_philox(1, 0)
ens.set_state(np.random.RandomState(0).randn(N, D))
ens.eval_state_log_prob()
ens.chain_config(K)
ens.run(K, 1, True)
ens.sync()
gb = K * N * D * 8 / 1e9
for rep in range(3):
    t0 = time.perf_counter()
    c = ens.chain_read(0, 0, K)
    dt = time.perf_counter() - t0
    print("contiguous read %.2f GB in %.3f s = %.1f GB/s" % (gb, dt, gb / dt), flush=True)
t0 = time.perf_counter()
c2 = ens.chain_read(0, 1, K, 2)
dt = time.perf_counter() - t0
print("strided (thin=2) %.2f GB in %.3f s = %.1f GB/s" % (gb / 2, dt, gb / 2 / dt))
t0 = time.perf_counter()
z = np.empty((K, N, D))
z[:] = 1.0
print("host alloc+touch of the same size: %.3f s" % (time.perf_counter() - t0))
