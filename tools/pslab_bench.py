"""Persistent slab kernel (k_persist_slab) against the per-half-step launches: us/step over ndim x walkers (verdict round 5, items
5 and 6).   usage: python tools/pslab_bench.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from emcee_amd import _lib  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200


class WL(bench.Workload):
    def __init__(self, N, D, move):
        self.key, self.N, self.D = "pslab", N, D
        mu, cov, icov = bench.dense_gaussian(D)
        self.target = (_lib.TARGET_DENSE, mu, icov, 0.0)
        self.p0 = mu + np.random.default_rng(1).standard_normal((N, D)) @ np.linalg.cholesky(cov).T
        kind = {"stretch": 0, "de": 1}[move]
        self.moves, self.weights = [(move, _lib.MoveDesc(kind, 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * D), 1.7))], [1.0]
        self.label = "%d x %d dense, %s" % (N, D, move)


print("%-28s %12s %12s %8s   roofline frac (persistent)" % ("shape", "persistent", "per-half-step", "gain"))
for move in ("stretch", "de"):
    for D in (128, 112, 96, 80, 66):
        for N in (1024, 4096, 8192, 16384, 32768, 65536):
            if move == "de" and D not in (128, 96):
                continue
            wl = WL(N, D, move)
            out = []
            for ps in (1, 0):
                r = bench.measure_single(wl, K, 10, want_kernel=False, spin_s=0.05, tuning={"persist_slab": 2 * ps})
                out.append((r["wall_s"] * 1e6 / K, r.get("halfsteps_per_launch", 1.0)))
            B = wl.bytes_per_update(False)
            print("%-28s %9.2f us %9.2f us %7.2fx   %.3f   (%.0f half-steps a launch)" % (
                wl.label, out[0][0], out[1][0], out[1][0] / out[0][0], N * B / (out[0][0] * 1e-6) / 8e12, out[0][1]), flush=True)
