"""The staggered partner loads (tuning persist_stagger) on the other shapes the dense ndim <= 64 persistent kernels take: odd ndim
(emx_podd.hip), the DE move alone, smaller device-wide ensembles.   usage: python tools/exp/stagger_shapes.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from emcee_amd import _lib  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20


class WL(bench.Workload):
    def __init__(self, N, D, move):
        self.key, self.N, self.D = "stg", N, D
        mu, cov, icov = bench.dense_gaussian(D)
        self.target = (_lib.TARGET_DENSE, mu, icov, 0.0)
        self.p0 = mu + np.random.default_rng(1).standard_normal((N, D)) @ np.linalg.cholesky(cov).T
        kind = {"stretch": 0, "de": 1, "snooker": 2}[move]
        self.moves, self.weights = [(move, _lib.MoveDesc(kind, 4 if kind == 2 else 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * D), 1.7))], [1.0]
        self.label = "%d x %d dense, %s" % (N, D, move)


SHAPES = [(int(a.split("x")[0]), int(a.split("x")[1]), a.split("x")[2]) for a in sys.argv[2:]]
for rep in (1, 2):
    for N, D, move in SHAPES or ((65536, 63, "stretch"), (65536, 33, "stretch"), (65536, 64, "de"), (32768, 64, "stretch"), (16384, 64, "stretch"), (65536, 32, "stretch"), (65536, 48, "stretch")):
        wl = WL(N, D, move)
        row = []
        for st in (0, 516, 528, 536, 1028, 1032):
            r = bench.measure_single(wl, K, 10, want_kernel=False, spin_s=0.05, tuning={"persist_stagger": st})
            row.append("%d: %.2f" % (st, r["wall_s"] * 1e6 / K))
        B = wl.bytes_per_update(False)
        row.append("frac at the first %.3f" % (wl.N * B / float(row[0].split()[1]) / 1e-6 / 8e12))
        print("%-30s %s" % (wl.label, "   ".join(row)), flush=True)
