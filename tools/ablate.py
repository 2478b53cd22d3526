"""Phase ablation of the dense half-step kernel (timing only; results are invalid by design)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
from tools.quick_bench import dense_params


def run(ablate, wpb=8, bpc=2, N=65536, D=64, steps=200, target="dense"):
    ens = DeviceEnsemble(N, D)
    rs = np.random.RandomState(1)
    mu, cov, icov = dense_params(D)
    if target == "dense":
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
    else:
        ens.set_target(_lib.TARGET_ISO)
    p0 = mu + rs.randn(N, D) @ np.linalg.cholesky(cov).T
    ens.set_moves([_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.2, 1.7)], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(1, 0)
    ens.set_state(p0)
    ens.eval_state_log_prob()
    ens.set_tuning("ablate", ablate)
    ens.set_tuning("waves_per_block", wpb)
    ens.set_tuning("blocks_per_cu", bpc)
    ens.run(20, 1, False)
    ens.sync()
    ens.timer_start()
    ens.run(steps, 1, False)
    ms = ens.timer_stop()
    ens.close()
    return ms / steps * 1e3


if __name__ == "__main__":
    names = {0: "full", 64: "empty kernels (launch floor)", 128: "plan+row loads only", 16: "no MFMA+decision stage", 1: "no MFMA loop", 8: "no commit"}
    for rep in range(2):
        for ab, nm in names.items():
            print("ablate=%-3d %-34s %.2f us/step" % (ab, nm, run(ab)), flush=True)
    for wpb, bpc in ((4, 2), (8, 1), (8, 2)):
        print("full wpb=%d bpc=%d  %.2f us/step" % (wpb, bpc, run(0, wpb, bpc)), flush=True)
    print("iso full %.2f %.2f us/step" % (run(0, target="iso"), run(0, target="iso")))
