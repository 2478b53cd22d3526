"""A longer bit-equality run of the stored-chain paths: k_persist<..., ROWS_LATE> (one-XCD and device-wide) against the per-half-step
launches, chain and log-probs of every stored step.   usage: python tools/exp/rows_late_long.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

for N, nsteps, thin in ((4096, 1500, 1), (8192, 600, 3), (65536, 60, 1), (1024, 3000, 2)):
    wl = bench.Workload("c2", N)
    out = []
    for persist in (1, 0):
        ens = DeviceEnsemble(wl.N, wl.D, device=0)
        wl.install(ens, "philox")
        ens.set_tuning("persist", persist)
        ens.chain_config(nsteps)
        for _ in range(3):
            ens.run(nsteps // 3, thin, True)
        assert ens.status() == 0
        x, lp = ens.get_state()
        info = ens.persist_info()
        out.append((x, lp, ens.chain_read(0, 0, nsteps, max(1, nsteps // 40)), ens.chain_read(1, 0, nsteps), ens.accepted_counts(), info))
        ens.close()
    same = all(np.array_equal(a, b) for a, b in zip(out[0][:5], out[1][:5]))
    print("%6d x 64, %d stored steps (thin_by %d): persistent launches %d (one-XCD %s) vs %d -- %s" % (
        N, nsteps, thin, out[0][5]["launches"], out[0][5].get("local_launches"), out[1][5]["launches"], "bit-identical" if same else "DIFFERENT"), flush=True)
    assert same
