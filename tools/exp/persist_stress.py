"""Stress the persistent kernel's parity: T trials x nsteps at C2's size, fresh Philox seed each, against persist=0.
  usage: python tools/exp/persist_stress.py [trials] [nsteps] [store] [nwalkers] [thin_by]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 37
store = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
N = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
thin = int(sys.argv[5]) if len(sys.argv) > 5 else 1
import os
wl = bench.Workload(os.environ.get("PERSIST_CFG", "c2"), N)
ens = []
for persist in (1, 0):
    e = DeviceEnsemble(wl.N, wl.D, device=0)
    wl.install(e, "philox")
    e.set_tuning("persist", 3 * persist)
    e.set_tuning("persist_timeout_ms", 50)
    if store:
        e.chain_config(nsteps)
    ens.append(e)
bad = 0
for t in range(T):
    got = []
    for e in ens:
        e.set_state(wl.p0)
        e.eval_state_log_prob()
        e.set_philox(1000 + t, 0)
        if store:
            e.chain_reset()
        e.run(nsteps, thin, store)
        x, lp = e.get_state()
        got.append((x, lp, e.status(), e.chain_read(0, 0, nsteps) if store else None, e.accepted_mask(), e.accepted_counts()))
    (x1, l1, s1, c1, a1, n1), (x0, l0, s0, c0, a0, n0) = got
    rows = int((x1 != x0).any(1).sum()) + int((a1 != a0).sum()) + int((n1 != n0).sum())
    first = -1
    if store and not np.array_equal(c1, c0):
        first = int(np.argmax((c1 != c0).any((1, 2))))
    if rows or s1 or s0 or first >= 0:
        bad += 1
    print("trial %2d: %d rows differ, %d log-probs differ, first differing stored step %d, status %d/%d" % (
        t, rows, int((l1 != l0).sum()), first, s1, s0), flush=True)
print("launches", ens[0].persist_info(), " BAD TRIALS: %d of %d" % (bad, T))
