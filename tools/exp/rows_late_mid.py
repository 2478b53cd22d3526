"""Stored chains at mid-size ensembles (the one-XCD persistent form) and at 16 384 / 32 768 walkers: k_persist<..., ROWS_LATE> or not.
usage: python tools/exp/rows_late_mid.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for rep in (1, 2):
    for N in (1024, 4096, 8192, 16384, 32768):
        wl = bench.Workload("c2", N)
        row = []
        for late in (0, 1, 2):
            r = bench.measure_single(wl, K, 10, want_kernel=False, spin_s=0.05, store=True, tuning={"persist_rows_late": late})
            row.append("%d: %.2f" % (late, r["wall_s"] * 1e6 / K))
        print("%6d x 64 dense, stretch, chain stored   persist_rows_late %s" % (N, "   ".join(row)), flush=True)
