"""Per-phase latencies of the dense half-step kernel from in-kernel timestamps (wave 0 of every workgroup).

`phase_clock` tuning key: k_halfstep records s_memtime at its phase boundaries; this script reads the
samples of the LAST launch, converts them with the 100 MHz wall clock recorded at both ends, and prints
the median / p90 over workgroups of each phase."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, ".")
# the timestamps cost 1 % and are therefore compiled out of the shipped library: use an instrumented build
_STAMPS = os.path.abspath("emcee_amd/libemx_stamps.so")
if not os.path.exists(_STAMPS):
    subprocess.check_call(["bash", "tools/ab_variants.sh", "stamps", "-DEMX_OPT_STAMPS=1"])
os.environ["EMX_LIB"] = _STAMPS
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402
from tools.quick_bench import dense_params  # noqa: E402

NAMES = ["start -> plan loads issued", "plan arrived, row loads issued", "image published + barrier",
         "rows arrived, proposals, tile written", "A fragments loaded (LDS)", "MFMA chain", "row reductions (DPP)",
         "decisions (lp/logu arrived, stores issued)", "commit (tile -> X)", "store drain"]


def main(target="dense", N=65536, D=64):
    import torch
    from emcee_amd.parallel import _DevView
    ens = DeviceEnsemble(N, D)
    rs = np.random.RandomState(1)
    mu, cov, icov = dense_params(D)
    if target == "dense":
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
    else:
        ens.set_target(_lib.TARGET_ISO)
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(1, 0)
    ens.set_state(mu + rs.randn(N, D) @ np.linalg.cholesky(cov).T)
    ens.eval_state_log_prob()
    ens.run(200, 1, False)
    ens.set_tuning("phase_clock", 1)
    rows = []
    for _ in range(20):
        ens.run(3, 1, False)
        ens.sync()
        ptr, nbytes = ens.device_ptr(7)
        t = torch.as_tensor(_DevView(ptr, nbytes // 8), device=torch.device("cuda", 0))
        raw = t.view(torch.int64).cpu().numpy().reshape(-1, 16)
        raw = raw[raw[:, 0] != 0]
        rows.append(raw.copy())
    ens.set_tuning("phase_clock", 0)
    ens.close()
    raw = np.concatenate(rows)
    ticks = (raw[:, 10] - raw[:, 0]).astype(float)
    wall = (raw[:, 12] - raw[:, 11]).astype(float) * 10.0          # ns at 100 MHz
    ok = wall > 0
    ns_per_tick = np.median(wall[ok] / ticks[ok])
    print("%s %dx%d: %d workgroup samples, s_memtime tick = %.3f ns, wave-0 lifetime median %.2f us (p90 %.2f)" %
          (target, N, D, len(raw), ns_per_tick, np.median(ticks) * ns_per_tick / 1e3, np.percentile(ticks, 90) * ns_per_tick / 1e3))
    for k in range(10):
        a, b = raw[:, k], raw[:, k + 1]
        good = (a != 0) & (b != 0)
        if not good.any():
            continue
        d = (b[good] - a[good]) * ns_per_tick / 1e3
        print("  %-46s median %6.2f us   p90 %6.2f us" % (NAMES[k], np.median(d), np.percentile(d, 90)))
    a, b, c2 = raw[:, 1], raw[:, 13], raw[:, 2]
    good = (a != 0) & (b != 0)
    if good.any():
        print("  (instrumented) plan loads issued -> arrived   median %6.2f us;  arrived -> row loads issued  median %6.2f us" %
              (np.median((b[good] - a[good]) * ns_per_tick / 1e3), np.median((c2[good] - b[good]) * ns_per_tick / 1e3)))
    # spread of start / end times across workgroups of one launch (last sample set)
    last = rows[-1]
    s0 = (last[:, 11] - last[:, 11].min()) * 10.0 / 1e3
    e0 = (last[:, 12] - last[:, 11].min()) * 10.0 / 1e3
    print("  last launch: workgroup start spread %.2f us, first end %.2f us, last end %.2f us" % (s0.max(), e0.min(), e0.max()))


if __name__ == "__main__":
    main("dense")
    main("iso")
