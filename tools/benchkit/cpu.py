"""bench.py: the CPU baseline -- reference emcee itself (oracle/_ref) or the NumPy port (oracle/), timed on the box's host cores."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools.benchkit.model import *  # noqa: F401,F403
from tools.benchkit.out import log



# ------------------------------------------------------------------------------------------------ CPU baseline
def usable_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a container on a 256-thread
    host may be limited to a handful: oversubscribing it makes every parallel leg slower than the serial one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:  # noqa: BLE001
            pass
    return n


def _port_leg(so, wl, fn, budget_s, label, cores):
    rs = np.random.RandomState(7)
    out = so.run(wl.p0, 1, fn, rs, store=False)                  # warm-up (page faults, BLAS initialisation)
    t0 = time.perf_counter()
    out = so.run(out["coords"], 1, fn, rs, store=False, log_prob0=out["lp"])
    t1 = time.perf_counter() - t0
    nst = int(min(1000, max(2, budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    so.run(out["coords"], nst, fn, rs, store=False, log_prob0=out["lp"])
    dt = time.perf_counter() - t0
    return {"mode": label, "wu_per_s": wl.N * nst / dt, "ms_per_step": dt * 1e3 / nst, "steps": nst, "seconds": dt, "cores": cores}


_POOL_MU = _POOL_ICOV = None


def _pool_init(mu, icov):
    global _POOL_MU, _POOL_ICOV
    _POOL_MU, _POOL_ICOV = mu, icov


def _pool_lp(x):
    d = x - _POOL_MU
    return -0.5 * float(np.dot(d, _POOL_ICOV @ d))


def cpu_baseline(wl, budget_s=14.0):
    """The host-core baseline of the headline workload (a reported number, not the optimisation target).

    Build container (/root/reference importable): reference emcee ITSELF, kind "reference".  GPU box: the NumPy port of
    its vectorize=True path (oracle/sampler_oracle.py, pinned to the reference by tests/golden), kind "port", in the
    reference's three documented modes, plus the committed reference timings from the build container."""
    from oracle import ref_shim
    from oracle import sampler_oracle as so
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # noqa: BLE001
        threadpool_limits = None
    mu, cov, icov = wl.params
    ncores = usable_cores()
    embedded = None
    path = os.path.join(ROOT, "profiles", "r02", "cpu_reference.json")
    if os.path.exists(path):
        try:
            embedded = json.load(open(path))
        except Exception:  # noqa: BLE001
            embedded = None

    if ref_shim.available():
        emcee = ref_shim.import_reference()

        def ref_leg(label, cores, **kw):
            s = emcee.EnsembleSampler(wl.N, wl.D, kw.pop("fn"), **kw)
            s._random.seed(7)
            st = s.run_mcmc(wl.p0, 1, skip_initial_state_check=True, store=False)
            t0 = time.perf_counter()
            st = s.run_mcmc(st, 1, skip_initial_state_check=True, store=False)
            t1 = time.perf_counter() - t0
            nst = int(min(1000, max(2, budget_s / max(t1, 1e-3))))
            t0 = time.perf_counter()
            s.run_mcmc(st, nst, skip_initial_state_check=True, store=False)
            dt = time.perf_counter() - t0
            return {"mode": label, "wu_per_s": wl.N * nst / dt, "ms_per_step": dt * 1e3 / nst, "steps": nst, "seconds": dt, "cores": cores}

        vec = lambda x: -0.5 * np.einsum("ij,ij->i", (x - mu) @ icov, x - mu)  # noqa: E731
        legs = []
        # the three ways the reference evaluates the ensemble's log-probs (ensemble.py:486-496): vectorize=True with the BLAS
        # form on one thread and on every core, and its documented parallel path, pool.map over walkers
        full_budget = budget_s
        if threadpool_limits is not None:
            with threadpool_limits(limits=1):
                legs.append(ref_leg("vectorize=True, 1 BLAS thread", 1, fn=vec, vectorize=True))
            budget_s = full_budget / 2
            with threadpool_limits(limits=ncores):
                legs.append(ref_leg("vectorize=True, %d BLAS threads" % ncores, ncores, fn=vec, vectorize=True))
        else:
            legs.append(ref_leg("vectorize=True, default BLAS threads", ncores, fn=vec, vectorize=True))
        budget_s = full_budget / 2
        try:
            import multiprocessing
            nproc = min(ncores, 32)
            _pool_init(mu, icov)
            with multiprocessing.Pool(nproc, initializer=_pool_init, initargs=(mu, icov)) as pool:
                legs.append(ref_leg("per-walker log_prob_fn, multiprocessing.Pool(%d)" % nproc, nproc, fn=_pool_lp, pool=pool))
        except Exception as e:  # noqa: BLE001
            legs.append({"mode": "per-walker log_prob_fn, multiprocessing.Pool", "error": repr(e)})
        legs_ok = [r for r in legs if "wu_per_s" in r]
        legs, all_legs = legs_ok, legs
        best = max(legs, key=lambda r: r["wu_per_s"])
        return {"value": best["wu_per_s"], "unit": "walker-updates/s", "cores": best["cores"], "kind": "reference",
                "sample": "reference emcee itself (%s) run_mcmc on %s; best mode '%s': %d steps, %.1f s; host has %d cores"
                          % (ref_shim.source(), wl.label, best["mode"], best["steps"], best["seconds"], ncores),
                "modes": all_legs}

    fn = lambda x: so.dense_gauss(x, mu, icov)  # noqa: E731
    legs = []
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            legs.append(_port_leg(so, wl, fn, budget_s, "port of vectorize=True, 1 BLAS thread", 1))
        with threadpool_limits(limits=ncores):
            legs.append(_port_leg(so, wl, fn, budget_s / 2, "port of vectorize=True, %d BLAS threads" % ncores, ncores))
    else:
        legs.append(_port_leg(so, wl, fn, budget_s, "port of vectorize=True, default BLAS threads", ncores))
    # the reference's documented parallel path: per-walker log_prob_fn through pool.map (ensemble.py:492-496)
    try:
        import multiprocessing
        nproc = min(ncores, 32)
        with multiprocessing.Pool(nproc, initializer=_pool_init, initargs=(mu, icov)) as pool:
            chunk = max(1, wl.N // 2 // (4 * nproc))
            pfn = lambda x: np.asarray(pool.map(_pool_lp, x, chunksize=chunk))  # noqa: E731
            if threadpool_limits is not None:
                with threadpool_limits(limits=1):
                    legs.append(_port_leg(so, wl, pfn, budget_s / 2, "port, per-walker log_prob_fn via multiprocessing.Pool(%d)" % nproc, nproc))
            else:
                legs.append(_port_leg(so, wl, pfn, budget_s / 2, "port, per-walker log_prob_fn via multiprocessing.Pool(%d)" % nproc, nproc))
    except Exception as e:  # noqa: BLE001
        legs.append({"mode": "port, multiprocessing.Pool", "error": repr(e)})
    head = legs[0]
    out = {"value": head["wu_per_s"], "unit": "walker-updates/s", "cores": head["cores"], "kind": "port",
           "sample": "oracle/sampler_oracle.py (NumPy restatement of emcee's vectorize=True path; /root/reference is absent on this "
                     "box), %d steps of %s, %.1f s, BLAS threads=%d, host has %d cores"
                     % (head["steps"], wl.label, head["seconds"], head["cores"], ncores),
           "modes": legs}
    if embedded is not None:
        out["reference_build_container"] = {"source": "profiles/r02/cpu_reference.json (tools/cpu_reference.py; static: measured in the "
                                                      "build container, not on this box)",
                                            "host": embedded.get("host"),
                                            "modes": {k: {"wu_per_s": v["wu_per_s"], "ms_per_step": v["ms_per_step"], "cores": v["cores"]}
                                                      for k, v in embedded.get("modes", {}).items()}}
    return out
