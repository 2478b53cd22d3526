"""Exact mode at C2 with the plans made on the device vs by the host pipeline: us/step, where the producer's time goes.
  usage: python tools/mtdev_probe.py [nwalkers] [ndim] [steps] [modes, e.g. 1,0,1]"""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 400
key = "c2" if D == 64 else "c3"
wl = bench.Workload(key, N)
MODES = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 0, 1]
for dev in MODES:
    e = DeviceEnsemble(wl.N, wl.D, device=0)
    wl.install(e, "mt19937")
    e.set_tuning("mt_device", 2 if dev else 0)          # 2: the device producer at any size
    t0 = time.perf_counter()
    e.run(40, 1, False)
    e.sync()
    first = time.perf_counter() - t0
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        e.run(K, 1, False)
        e.sync()
        best = min(best, time.perf_counter() - t0)
    print("mt_device=%d  N=%d D=%d: first 40 steps %.2f ms, then %.2f us/step (best of 5 x %d); status %d; mtdev %r; tok %r; persist %r; pipeline %r"
          % (dev, N, D, first * 1e3, best * 1e6 / K, K, e.status(), e.mtdev_info(), e.mtdev_tok_stats() if dev else None, e.persist_info(),
             e.pipeline_stats()), flush=True)
    e.close()
