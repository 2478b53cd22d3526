"""Soak of the exact-mode default path at the headline size (round 6: regen steps on the persistent kernel): long runs from several
seeds, the persistent path against the per-step path -- final coordinates, log-probs, accept counters and generator state must agree.
  usage: python tools/soak_exact.py [steps] [seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for N in (65536, 32768):
    wl = bench.Workload("c2", N)
    for seed in range(seeds):
        recs = []
        for pe in (1, 0):
            ens = DeviceEnsemble(wl.N, wl.D, device=0)
            wl.install(ens, "mt19937", seed=1000 + seed)
            ens.set_tuning("persist_exact", pe)
            n = steps if pe else steps // 4              # (the per-step control is slower: a quarter of the steps, compared there)
            out = {}
            done = 0
            for chunk in (steps // 4, steps - steps // 4):
                if done >= n:
                    break
                ens.run(chunk, 1, False)
                done += chunk
                x, lp = ens.get_state()
                out[done] = (x, lp, ens.accepted_counts(), ens.get_mt19937())
            assert ens.status() == 0
            info, hand = ens.persist_info(), ens.pipeline_handovers()
            ens.close()
            recs.append((out, info, hand))
        (a, ia, ha), (b, ib, hb) = recs
        k = steps // 4
        same = all(np.array_equal(a[k][q], b[k][q]) for q in range(3)) and np.array_equal(a[k][3][1], b[k][3][1]) and a[k][3][2] == b[k][3][2]
        print("N=%d seed %d: %d steps on the persistent kernel (%d launches, %d regen steps, recovered %d); equal to the per-step path after %d steps: %s"
              % (N, seed, steps, ia["launches"], ha["regen_steps"], ia["recovered"], k, same), flush=True)
        assert same and ia["recovered"] == 0
print("soak ok")
