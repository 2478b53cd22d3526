"""Host <-> device copies of the state at the headline size through the C ABI (pageable NumPy arrays)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

N, D = 65536, 64
ens = DeviceEnsemble(N, D)
ens.set_target(_lib.TARGET_ISO)
x = np.random.RandomState(0).randn(N, D)
mb = N * D * 8 / 1e6
for rep in range(3):
    t0 = time.perf_counter()
    ens.set_state(x)
    t1 = time.perf_counter()
    y, _ = ens.get_state()
    t2 = time.perf_counter()
    print("set_state %.1f MB: %.2f ms (%.1f GB/s) | get_state: %.2f ms (%.1f GB/s)" %
          (mb, (t1 - t0) * 1e3, mb / (t1 - t0) / 1e3, (t2 - t1) * 1e3, mb / (t2 - t1) / 1e3))
    assert np.array_equal(x, y)
