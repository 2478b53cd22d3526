"""Where a half-step of the persistent kernel (k_persist) goes: in-kernel timestamps of the first wave of every workgroup, summed
over the half-steps of a launch (instrumented build, -DEMX_OPT_STAMPS=1: tools/ab_variants.sh stamps "-DEMX_OPT_STAMPS=1").

  usage: python tools/persist_phase_clock.py [nwalkers] [ndim] [store]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_STAMPS = os.environ.get("EMX_STAMPS_LIB") or os.path.join(ROOT, "emcee_amd", "libemx_stamps.so")
if not os.path.exists(_STAMPS):
    subprocess.check_call(["bash", os.path.join(ROOT, "tools", "ab_variants.sh"), "stamps", "-DEMX_OPT_STAMPS=1"])
os.environ["EMX_LIB"] = _STAMPS
import bench  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402

NAMES = ["partner rows + next plan entries arrive (sc1 round trip)", "proposals, tile written, next own rows issued",
         "LDS fragments + MFMA chain + row reductions", "decisions, commit stores issued", "stores acknowledged (vmcnt 0)",
         "device-wide barrier (arrive, poll)"]


def main(N=65536, D=64, store=0):
    import torch
    from emcee_amd.parallel import _DevView
    wl = bench.Workload("c2" if D == 64 else "c3", N)
    ens = DeviceEnsemble(wl.N, wl.D, device=0)
    wl.install(ens, "philox")
    import json
    for key, val in json.loads(os.environ.get("EMX_AB_TUNE", "{}")).items():
        ens.set_tuning(key, val)
    if store:
        ens.chain_config(4000)
    ens.run(200, 1, bool(store))
    assert ens.persist_info()["launches"] > 0, "this configuration does not take the persistent kernel"
    ens.set_tuning("phase_clock", 1)
    rows = []
    for _ in range(20):
        ens.run(16, 1, bool(store))
        ens.sync()
        ptr, nbytes = ens.device_ptr(7)
        t = torch.as_tensor(_DevView(ptr, nbytes // 8), device=torch.device("cuda", 0))
        raw = t.view(torch.int64).cpu().numpy().reshape(-1, 16)
        rows.append(raw[raw[:, 6] != 0].copy())
    info = ens.persist_info()
    ens.set_tuning("phase_clock", 0)
    ens.close()
    raw = np.concatenate(rows).astype(float)
    niter = raw[:, 6]
    wall_ns = (raw[:, 12] - raw[:, 11]) * 10.0
    ticks = raw[:, :6].sum(axis=1)
    ns_per_tick = np.median(wall_ns / ticks)
    per = raw[:, :6] / niter[:, None] * ns_per_tick / 1e3          # us per half-step
    # the barrier is passed niter - 1 times a launch, the other phases niter times
    per[:, 5] *= niter / np.maximum(niter - 1, 1)
    names = NAMES
    print("tuning %s, library %s" % (os.environ.get("EMX_AB_TUNE", "{}"), os.path.basename(_STAMPS)))
    print("k_persist %d x %d%s: %d workgroup-launch samples (%d half-steps a launch), counter tick %.2f ns, persist launches so far %d"
          % (N, D, ", stored chain" if store else "", len(raw), int(np.median(niter)), ns_per_tick, info["launches"]))
    print("  wave-0 lifetime per half-step: median %.2f us" % np.median(wall_ns / niter / 1e3))
    for k, name in enumerate(names):
        print("  %-62s median %6.2f us   p10 %6.2f   p90 %6.2f   (%4.1f %%)"
              % (name, np.median(per[:, k]), np.percentile(per[:, k], 10), np.percentile(per[:, k], 90),
                 100 * np.median(per[:, k]) / np.median(per.sum(axis=1))))
    print("  sum of medians %.2f us per half-step" % np.median(per, axis=0).sum())


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 65536, int(a[1]) if len(a) > 1 else 64, int(a[2]) if len(a) > 2 else 0)
