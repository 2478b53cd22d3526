"""k_persist_p2p (no barrier between the half-steps) against k_persist (device-wide barrier) on one box, interleaved.

  usage: python tools/exp/p2p_ab.py [steps] [reps]     -> one line per (walkers, ndim, store): us/step both ways (hip events)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402


def make(key, N, p2p, store, steps):
    wl = bench.Workload(key, N)
    ens = DeviceEnsemble(wl.N, wl.D, device=0)
    wl.install(ens, "philox")
    ens.set_tuning(os.environ.get("AB_KEY", "persist_p2p"), p2p)
    for kv in os.environ.get("AB_FIX", "").split(","):
        if kv:
            ens.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
    if store:
        ens.chain_config(steps + 64)
    return ens


def timed(ens, steps, store):
    if store:
        ens.chain_reset()
    ens.sync()
    ens.timer_start()
    ens.run(steps, 1, bool(store))
    return ens.timer_stop() * 1e3 / steps


def main(steps=1600, reps=4, quick=0):
    cases = (("c2", 65536, 0), ("c2", 65536, 1), ("c2", 49152, 0), ("c2", 32768, 0), ("c2", 16384, 0), ("c2", 131072, 0))
    if quick:
        cases = (("c2", 65536, 0), ("c2", 32768, 0))
    print("library: %s   key (1 | 0): %s   fixed: %s" % (os.environ.get("EMX_LIB", "emcee_amd/libemx.so"), os.environ.get("AB_KEY", "persist_p2p"), os.environ.get("AB_FIX", "")))
    for key, N, store in cases:
        st = min(steps, 400) if store else steps
        e = {p: make(key, N, p, store, st) for p in (1, 0)}
        for p in e:
            e[p].run(64, 1, False)
        t = {1: [], 0: []}
        for _ in range(reps):
            for p in (1, 0):
                t[p].append(timed(e[p], st, store))
        info = {p: e[p].persist_info() for p in e}
        acc = float(e[1].accepted_mask().mean())
        stt = [e[p].status() for p in e]
        for p in e:
            e[p].close()
        print("%-3s %7d x 64 store=%d   on %6.2f us/step (min %6.2f)   off %6.2f (min %6.2f)   ratio %.3f   p2p launches %d/%d  acc %.3f status %s"
              % (key, N, store, np.median(t[1]), min(t[1]), np.median(t[0]), min(t[0]), np.median(t[1]) / np.median(t[0]),
                 info[1]["p2p_launches"], info[1]["launches"], acc, stt), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 1600, int(a[1]) if len(a) > 1 else 4, int(a[2]) if len(a) > 2 else 0)
