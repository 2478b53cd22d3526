import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble
for N, D, tgt, kind, S in ((2048, 8, "diag", 1, 2), (2048, 10, "diag", 1, 2), (2048, 16, "diag", 1, 2), (1024, 8, "iso", 2, 4), (1024, 8, "iso", 0, 2), (1024, 16, "iso", 2, 4), (1024, 6, "iso", 0, 2), (1024, 5, "iso", 0, 2)):
    for rng in ("mt", "philox"):
        ens = DeviceEnsemble(N, D)
        if tgt == "iso":
            ens.set_target(_lib.TARGET_ISO)
        else:
            ens.set_target(_lib.TARGET_DIAG, np.zeros(D), np.ones(D))
        ens.set_moves([_lib.MoveDesc(kind, S, 1, 0, 2.0, 1e-5, 0.3, 1.7)], np.array([1.0]))
        if rng == "mt":
            ens.set_rng_mode(_lib.RNG_MT19937); ens.set_mt19937(np.random.RandomState(5).get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX); ens.set_philox(3, 0)
        ens.set_state(np.random.RandomState(1).randn(N, D)); ens.eval_state_log_prob()
        ens.run(32, 1, False); ens.sync()
        print(N, D, tgt, "kind", kind, rng, ens.persist_info())
        ens.close()
