"""device exact-plan producer at C2's shape: wall time of emx_run(K) for several K, every repetition printed"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = bench.Workload("c2", N)
e = DeviceEnsemble(wl.N, wl.D, device=0)
wl.install(e, "mt19937")
e.set_tuning("mt_device", 2)
e.run(64, 1, False); e.sync()
for K in (160, 400, 800, 1600):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); e.run(K, 1, False); e.sync(); ts.append((time.perf_counter() - t0) * 1e6 / K)
    print("K=%4d: %s us/step; tok %r" % (K, " ".join("%.1f" % t for t in ts), {k: round(v, 1) if isinstance(v, float) else v for k, v in e.mtdev_tok_stats().items()}), flush=True)
e.close()
