"""Reference emcee ITSELF timed on the host cores of the box this runs on (modes a-c).

Build-container only: /root/reference does not travel to the GPU box, so `bench.py` embeds the JSON this
script writes (profiles/r02/cpu_reference.json) next to the port it can time there.  Workload = BASELINE
configs[1] (65536 x 64 correlated Gaussian, StretchMove a=2, store=False, initial log-prob excluded):

  (a) vectorize=True, BLAS form of the log-prob, 1 BLAS thread and all cores
  (b) per-walker log_prob_fn through multiprocessing.Pool(ncores)   (docs/tutorials/parallel.ipynb:152-171)
  (c) per-walker serial map                                          (ensemble.py:492-496)

Usage: python tools/cpu_reference.py [--out profiles/r02/cpu_reference.json] [--budget 20]
"""
import argparse
import json
import multiprocessing
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MU = ICOV = None


def dense_gaussian(ndim, seed=0):
    rs = np.random.RandomState(seed)
    mu = rs.randn(ndim)
    A = rs.randn(ndim, ndim)
    cov = A @ A.T / ndim + 0.1 * np.eye(ndim)
    icov = np.linalg.inv(cov)
    return mu, cov, 0.5 * (icov + icov.T)


def lp_vector(x):
    d = x - MU
    return -0.5 * np.einsum("ij,ij->i", d @ ICOV, d)


def lp_walker(x):
    d = x - MU
    return -0.5 * np.dot(d, ICOV @ d)


def time_mode(emcee, N, D, p0, budget_s, **kw):
    s = emcee.EnsembleSampler(N, D, kw.pop("fn"), **kw)
    s._random.seed(7)
    st = s.run_mcmc(p0, 1, skip_initial_state_check=True, store=False)      # pays the initial log-prob
    t0 = time.perf_counter()
    st = s.run_mcmc(st, 1, skip_initial_state_check=True, store=False)
    t1 = time.perf_counter() - t0
    n = int(max(3, min(200, budget_s / max(t1, 1e-3))))
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        st = s.run_mcmc(st, n, skip_initial_state_check=True, store=False)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"steps": n, "seconds": best, "ms_per_step": best * 1e3 / n, "wu_per_s": N * n / best,
            "accept_frac_last_run": float(np.mean(s.acceptance_fraction)) if s.iteration else None}


def main():
    global MU, ICOV
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02", "cpu_reference.json"))
    ap.add_argument("--budget", type=float, default=15.0, help="seconds of CPU work per mode")
    ap.add_argument("--nwalkers", type=int, default=65536)
    ap.add_argument("--ndim", type=int, default=64)
    a = ap.parse_args()
    from oracle import ref_shim
    emcee = ref_shim.import_reference()
    from threadpoolctl import threadpool_limits
    N, D = a.nwalkers, a.ndim
    MU, cov, ICOV = dense_gaussian(D)
    p0 = MU + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
    from bench import usable_cores
    ncores = usable_cores()
    modes = {}
    with threadpool_limits(limits=1):
        modes["vectorize_1thread"] = dict(time_mode(emcee, N, D, p0, a.budget, fn=lp_vector, vectorize=True), cores=1)
        modes["per_walker_serial_map"] = dict(time_mode(emcee, N, D, p0, a.budget, fn=lp_walker), cores=1)
        with multiprocessing.Pool(ncores) as pool:
            modes["per_walker_pool"] = dict(time_mode(emcee, N, D, p0, a.budget, fn=lp_walker, pool=pool), cores=ncores)
    modes["vectorize_allthreads"] = dict(time_mode(emcee, N, D, p0, a.budget, fn=lp_vector, vectorize=True), cores=ncores)
    cpu = platform.processor()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {"what": "reference emcee (dfm/emcee at /root/reference/src) on host cores; configs[1] 65536x64 dense Gaussian, "
                   "StretchMove a=2, store=False, initial log-prob excluded, best of 2",
           "host": {"cpu": cpu, "cores": ncores, "numpy": np.__version__, "python": platform.python_version(),
                    "where": "build container (no GPU); /root/reference does not exist on the GPU box"},
           "nwalkers": N, "ndim": D, "modes": modes,
           "best_mode": max(modes, key=lambda k: modes[k]["wu_per_s"])}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
