"""Quick device-side throughput probe (not the contract bench): sweeps configs / spw."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402


def dense_params(D, seed=0):
    rs = np.random.RandomState(seed)
    mu = rs.randn(D)
    A = rs.randn(D, D)
    cov = A @ A.T / D + 0.1 * np.eye(D)
    icov = np.linalg.inv(cov)
    return mu, cov, 0.5 * (icov + icov.T)


def run(N, D, target, move=0, S=2, steps=200, spw=0, store=False, rng=_lib.RNG_PHILOX, bpc=2, gsigma=0.05, gmode=0):
    ens = DeviceEnsemble(N, D)
    rs = np.random.RandomState(1)
    if target == "dense":
        mu, cov, icov = dense_params(D)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        p0 = mu + rs.randn(N, D) @ np.linalg.cholesky(cov).T
    elif target == "iso":
        ens.set_target(_lib.TARGET_ISO)
        p0 = rs.randn(N, D)
    elif target == "diag":
        iv = 1.0 / (0.1 + rs.rand(D))
        ens.set_target(_lib.TARGET_DIAG, np.zeros(D), iv)
        p0 = rs.randn(N, D) / np.sqrt(iv)
    else:
        ens.set_target(_lib.TARGET_ROSENBROCK, scale=20.0)
        p0 = 1 + 0.1 * rs.randn(N, D)
    if move == 3:      # Gaussian Metropolis move, vector mode, step sized for ~25 % acceptance
        md = _lib.MoveDesc(3, 1, 0, gmode, 0.0, gsigma, 0.0, 0.0)
    else:
        md = _lib.MoveDesc(move, 4 if move == 2 else S, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * D), 1.7)
    ens.set_moves([md], np.array([1.0]))
    ens.set_rng_mode(rng)
    if rng == _lib.RNG_MT19937:
        ens.set_mt19937(np.random.RandomState(5).get_state())
    else:
        ens.set_philox(12345, 0)
    ens.set_state(p0)
    ens.eval_state_log_prob()
    if spw:
        ens.set_tuning("spw", spw)
    ens.set_tuning("blocks_per_cu", bpc)
    if store:
        ens.chain_config(steps + 20)
    ens.run(20, 1, store)
    ens.sync()
    ens.timer_start()
    t0 = time.perf_counter()
    ens.run(steps, 1, store)
    ms = ens.timer_stop()
    wall = time.perf_counter() - t0
    acc = ens.accepted_mask().mean()
    st = ens.status()
    nsp = md.nsplits
    ens.profile_enable(64)
    ens.run(32, 1, False)
    pl = ens.profile_read(64)
    ens.close()
    part = {0: 1, 1: 2, 2: 3, 3: 0}[move]
    B = (16 + 8 * part) * D + 17 + (8 * D + 8 if store else 0)
    wups = N * steps / (ms * 1e-3)
    return dict(N=N, D=D, target=target, move=move, spw=spw, store=store, rng=rng, ms_per_step=ms / steps,
                wall_ms_per_step=wall * 1e3 / steps, wu_per_s=wups, alg_GBps=wups * B / 1e9, frac=wups * B / 8e12,
                acc=float(acc), status=st, kernel_us=float(np.median(pl) * 1e3) if len(pl) else None,
                kernel_alg_GBps=float((N / nsp) * B / (np.median(pl) * 1e-3) / 1e9) if len(pl) else None)


if __name__ == "__main__":
    out = []
    cfgs = []
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for spw in (0, 16, 32, 64):
        cfgs.append(dict(N=65536, D=64, target="dense", spw=spw))
    for spw in (0, 4, 8, 16, 32, 64):
        cfgs.append(dict(N=65536, D=64, target="iso", spw=spw))
    cfgs += [dict(N=65536, D=64, target="dense", store=True, steps=100),
             dict(N=65536, D=64, target="dense", rng=_lib.RNG_MT19937, steps=50),
             dict(N=262144, D=32, target="rosen"),
             dict(N=16384, D=1024, target="diag"),
             dict(N=65536, D=64, target="dense", move=1),
             dict(N=65536, D=64, target="dense", move=2),
             dict(N=65536, D=64, target="dense", move=3, gsigma=0.03),
             dict(N=65536, D=64, target="iso", move=3, gsigma=0.25),
             dict(N=65536, D=64, target="iso", move=3, gsigma=1.0, gmode=1),
             dict(N=65536, D=64, target="dense", move=3, gsigma=0.03, rng=_lib.RNG_MT19937, steps=10),
             dict(N=32, D=5, target="iso", steps=2000),
             dict(N=1024, D=16, target="iso", steps=2000)]
    if only == "gauss":
        cfgs = [c for c in cfgs if c.get("move") == 3]
    if only == "moves":      # the dense 64-dim shape under every move
        cfgs = [c for c in cfgs if c.get("target") == "dense" and c.get("move", 0) in (1, 2, 3) and "rng" not in c]
    for c in cfgs:
        try:
            r = run(**c)
        except Exception as e:  # noqa: BLE001
            r = dict(cfg=c, error=str(e))
        print(json.dumps(r), flush=True)
        out.append(r)
    json.dump(out, open("gpurun_out/quick_bench%s.json" % ("_" + only if only else ""), "w"), indent=1)
