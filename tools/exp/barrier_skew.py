"""What a workgroup of k_persist waits for at the device-wide barrier: the instrumented build (-DEMX_OPT_STAMPS=1) leaves, per workgroup
and half-step, the 100 MHz wall clock when the workgroup ARRIVED (all of its waves at the barrier, before the arrival is counted) and
when it was RELEASED.  Per half-step: L = the last arrival; skew = L - arrival; mechanism = release - L.

  usage: python tools/exp/barrier_skew.py [nwalkers] [ndim]       env EMX_AB_TUNE='{"key": value}' sets tuning keys"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
_STAMPS = os.environ.get("EMX_STAMPS_LIB") or os.path.join(ROOT, "emcee_amd", "libemx_stamps.so")
if not os.path.exists(_STAMPS):
    subprocess.check_call(["bash", os.path.join(ROOT, "tools", "ab_variants.sh"), "stamps", "-DEMX_OPT_STAMPS=1"])
os.environ["EMX_LIB"] = _STAMPS
import bench  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402


def main(N=65536, D=64):
    import torch
    from emcee_amd.parallel import _DevView
    wl = bench.Workload("c2" if D == 64 else "c3", N)
    ens = DeviceEnsemble(wl.N, wl.D, device=0)
    wl.install(ens, "philox")
    for key, val in json.loads(os.environ.get("EMX_AB_TUNE", "{}")).items():
        ens.set_tuning(key, val)
    ens.run(200, 1, False)
    assert ens.persist_info()["launches"] > 0
    ens.set_tuning("phase_clock", 1)
    ng = N // 2 // 128
    recs = []
    for _ in range(30):
        ens.run(16, 1, False)
        ens.sync()
        ptr, nbytes = ens.device_ptr(7)
        t = torch.as_tensor(_DevView(ptr, nbytes // 8), device=torch.device("cuda", 0))
        raw = t.view(torch.int64).cpu().numpy()
        niter = int(raw[6])
        w = raw[4096:4096 + (niter - 1) * ng * 2].reshape(niter - 1, ng, 2).astype(np.float64) * 0.01       # us
        recs.append(w)
    ens.set_tuning("phase_clock", 0)
    ens.close()
    w = np.concatenate(recs)                       # (barriers, workgroups, [arrived, released])
    arr, rel = w[:, :, 0], w[:, :, 1]
    L = arr.max(axis=1, keepdims=True)
    first = arr.min(axis=1, keepdims=True)
    skew = L - arr
    mech = rel - L
    wait = rel - arr
    per = np.diff(L[:, 0])
    per = per[(per > 0) & (per < 100)]
    q = lambda a, p: np.percentile(a, p)
    print("tuning %s, library %s" % (os.environ.get("EMX_AB_TUNE", "{}"), os.path.basename(_STAMPS)))
    print("k_persist %d x %d: %d barriers x %d workgroups; last arrival to last arrival %.2f us (median)" % (N, D, w.shape[0], ng, np.median(per)))
    print("  a workgroup waits at the barrier (released - arrived)      median %5.2f us   p10 %5.2f   p90 %5.2f" % (np.median(wait), q(wait, 10), q(wait, 90)))
    print("  ... of that: for the last workgroup to arrive (skew)        median %5.2f us   p10 %5.2f   p90 %5.2f   mean %5.2f" % (np.median(skew), q(skew, 10), q(skew, 90), skew.mean()))
    print("  ... and from the last arrival to its own release            median %5.2f us   p10 %5.2f   p90 %5.2f   (first released %5.2f, last %5.2f)"
          % (np.median(mech), q(mech, 10), q(mech, 90), np.median(mech.min(axis=1)), np.median(mech.max(axis=1))))
    print("  first arrival to last arrival                              median %5.2f us   p90 %5.2f" % (np.median(L - first), q(L - first, 90)))
    # who is last: by workgroup and by XCD
    last = arr.argmax(axis=1)
    cnt = np.bincount(last, minlength=ng)
    print("  the last workgroup: by XCD %s; the ten most frequent workgroups %s" % (np.bincount(last & 7, minlength=8).tolist(), np.argsort(-cnt)[:10].tolist()))
    order = np.sort(skew, axis=1)           # per barrier: how far ahead of the last arrival the latest (0), second latest, ... workgroups were
    print("  ahead of the last arrival, k-th latest workgroup (median): " + "  ".join("k=%d %.2f" % (k, np.median(order[:, k])) for k in (1, 2, 4, 8, 16, 32, 64, 128, 255)))
    rel_arr = arr - np.median(arr, axis=1, keepdims=True)
    xcd = np.arange(ng) & 7
    print("  arrival against the barrier's median arrival, by XCD (mean us): " + "  ".join("%d: %+.2f" % (x, rel_arr[:, xcd == x].mean()) for x in range(8)))
    rel_rel = rel - np.median(rel, axis=1, keepdims=True)
    print("  release against the barrier's median release, by XCD (mean us): " + "  ".join("%d: %+.2f" % (x, rel_rel[:, xcd == x].mean()) for x in range(8)))
    # work between a release and the next arrival (inside a launch: consecutive barriers of one run() call; the stamps of a call are niter - 1 rows)
    nb = recs[0].shape[0]
    work = np.concatenate([r[1:, :, 0] - r[:-1, :, 1] for r in recs])
    print("  release -> next arrival (the half-step's work), by XCD (mean us): " + "  ".join("%d: %.2f" % (x, work[:, xcd == x].mean()) for x in range(8)))
    print("  ... by workgroup inside its XCD (mean over XCDs, workgroups 0-31 of each): " + " ".join("%+.2f" % rel_arr[:, (np.arange(ng) >> 3) == g].mean() for g in range(ng // 8)))


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 65536, int(a[1]) if len(a) > 1 else 64)
