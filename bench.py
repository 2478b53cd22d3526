#!/usr/bin/env python
"""Contract bench: walker-updates/sec of the fused red/blue half-step path on MI355X.

`python bench.py --gpus N --steps K --warmup W`  (N>1: launched by torch.distributed.run, one rank per GPU).

A "step" is one full ensemble step (every half-step of the move chosen for it) over an ensemble of synthetic
walkers resident in HBM.  The HEADLINE (`metric`, `value`, `ms_per_step`, `roofline`) is BASELINE.json configs[1]:
nwalkers=65536 per GPU, ndim=64, correlated Gaussian with a dense precision matrix, StretchMove a=2.0, counter-based
RNG, float64 -- weak-scaled over the GPUs.  The same JSON line also carries (rank 0, N=1):

  configs     C3 262144x32 Rosenbrock, C4 65536x64 DE+snooker mixture, C5 16384x1024 diagonal Gaussian and C2 with
              the chain stored every step: ms_per_step, wu_per_s, roofline fraction (SURVEY.md 8d bytes formulas)
  exact_mode  C2 under rng=mt19937 (the mode that reproduces reference emcee's chain for a seed)
  quality     acceptance fraction and integrated autocorrelation time of a 1024x64 run 68 tau long, next to the
              reference's numbers for the same configuration (tests/golden/quality_ref.json, build container)
  cpu_baseline reference emcee itself when /root/reference is importable (build container), otherwise the NumPy
              port (oracle/) timed here + the committed reference timings (profiles/r02/cpu_reference.json)

and at N>1 `multi_gpu`: C2 weak, C3 (262144 walkers sharded) and C5 (16384x1024, strong scaling), each under every
exchange protocol (emcee_amd/parallel.py), the fastest valid one reported per config.

Timing: after W warm-up steps, blocks of EXACTLY K steps are timed, each bracketed by barrier + synchronize, until
>= 50 ms have been measured; the MEDIAN block is reported (max over ranks per block).  `--single-block` restores the
one-block protocol.  stdout carries exactly one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X spec (MI355X_MICROARCH.md); ~6300 achievable
HBM_ACHIEVABLE_GBPS = 6300.0   # what a streaming kernel sustains on this part (same guide): the second yardstick of the beyond-MALL configs
MFMA_F64_PEAK_TFLOPS = 78.6    # dense f64 matrix peak (v_mfma_f64_16x16x4_f64: 256 flop x 4 SIMD x 256 CU x 2.4 GHz / 8 passes)
MIN_TIMED_MS = 50.0
MAX_BLOCKS = 400


# ------------------------------------------------------------------------------------------------ workloads
def dense_gaussian(ndim, seed=0):
    """Sigma = A A^T / D + 0.1 I, dense Sigma^-1."""
    rs = np.random.RandomState(seed)
    mu = rs.randn(ndim)
    A = rs.randn(ndim, ndim)
    cov = A @ A.T / ndim + 0.1 * np.eye(ndim)
    icov = np.linalg.inv(cov)
    return mu, cov, 0.5 * (icov + icov.T)


def partner_rows(kind):
    return {"stretch": 1, "de": 2, "snooker": 3}[kind]


def algorithmic_bytes(ndim, kind, store):
    """read x_k + partner rows, write x_k', log-prob in/out, accepted flag (+ chain row and log-prob)."""
    return (16 + 8 * partner_rows(kind)) * ndim + 17 + ((8 * ndim + 8) if store else 0)


def moved_bytes(ndim, kind, store, accept_frac):
    """Bytes that must actually MOVE per walker-update: SURVEY.md 8d's formula counts the 8*D write of x_k' for every proposal, but a
    rejected proposal writes nothing back (`Move.update` commits accepted rows only, move.py:12-45; the kernels do the same) -- so
    the coordinate write is weighted by the measured acceptance fraction: (8 + 8*partners)*D + 8*D*acc + 17 (+ chain row)."""
    return (8 + 8 * partner_rows(kind)) * ndim + 8 * ndim * accept_frac + 17 + ((8 * ndim + 8) if store else 0)


def roofline_audit(rl, wl, store, accept_frac, updates_per_s, traffic_bytes_per_launch=None, launch_s=None):
    """Make a roofline entry auditable (round-3 verdict): next to `achieved` (SURVEY 8d's nominal bytes) the acceptance-aware
    rate and, when PMC traffic is known, the rate of the bytes HBM really served.  An entry whose nominal rate exceeds what
    the memory system can deliver says which bytes never moved."""
    w = np.asarray(wl.weights) / np.sum(wl.weights)
    Bm = float(sum(wi * moved_bytes(wl.D, kind, store, accept_frac) for wi, (kind, _) in zip(w, wl.moves)))
    rl["moved_bytes_per_walker_update"] = Bm
    rl["achieved_moved"] = updates_per_s * Bm / 1e9
    rl["frac_moved"] = rl["achieved_moved"] / HBM_PEAK_GBPS
    rl["moved_is"] = "16*D + 8*D*accept_frac + 17 for the stretch move (%.3f accepted): rejected proposals write no row back" % accept_frac
    if traffic_bytes_per_launch and launch_s:
        rl["traffic_rate"] = traffic_bytes_per_launch / launch_s / 1e9
        rl["frac_traffic"] = rl["traffic_rate"] / HBM_PEAK_GBPS
    else:
        rl["frac_traffic"] = None
    nominal = rl.get("achieved")
    if nominal is not None and nominal > HBM_ACHIEVABLE_GBPS:
        B = wl.bytes_per_update(store)
        rl["above_achievable_because"] = (
            "nominal rate %.0f GB/s > the %.0f GB/s the memory system delivers: SURVEY 8d's %.0f B/update count %.0f B of coordinate "
            "writes per update that never happen at acceptance %.3f (moved: %.0f B/update -> %.0f GB/s)%s"
            % (nominal, HBM_ACHIEVABLE_GBPS, B, 8 * wl.D * (1 - accept_frac), accept_frac, Bm, rl["achieved_moved"],
               "" if wl.N * wl.D * 8 / 1e6 > 256.0 else "; the state also fits the 256 MB Infinity Cache, so part of the rest is not HBM traffic either"))
    return rl


class Workload(object):
    """One BASELINE.json configuration: synthetic inputs + how to install it on a DeviceEnsemble."""

    def __init__(self, key, nwalkers, make_p0=True):
        """make_p0=False: the description only (sizes, moves, byte formulas) -- the N > 1 orchestrators never touch the ensemble,
        and a weak-scaled start state is gigabytes of host normals per process"""
        from emcee_amd import _lib
        self.key = key
        self.N = int(nwalkers)
        self._make_p0 = make_p0
        std = lambda kind, D, S=2: _lib.MoveDesc({"stretch": 0, "de": 1, "snooker": 2}[kind], 4 if kind == "snooker" else S, 1, 0,  # noqa: E731
                                                 2.0, 1e-5, 2.38 / np.sqrt(2 * D), 1.7)
        rs = np.random.RandomState(1)
        if key in ("c2", "c4"):
            self.D = 64
            mu, cov, icov = dense_gaussian(self.D)
            self.params = (mu, cov, icov)
            self.target = (_lib.TARGET_DENSE, mu, icov, 0.0)
            self.p0 = mu + rs.randn(self.N, self.D) @ np.linalg.cholesky(cov).T if make_p0 else None     # equilibrium start
            if key == "c2":
                self.moves, self.weights = [("stretch", std("stretch", 64))], [1.0]
                self.label = "configs[1]: nwalkers=%d, ndim=64, dense-precision Gaussian, StretchMove a=2.0, nsplits=2" % self.N
            else:
                self.moves = [("de", std("de", 64)), ("snooker", std("snooker", 64))]
                self.weights = [0.8, 0.2]
                self.label = "configs[3]: nwalkers=%d, ndim=64, dense-precision Gaussian, DEMove 0.8 + DESnookerMove 0.2" % self.N
        elif key == "c3":
            self.D = 32
            self.target = (_lib.TARGET_ROSENBROCK, None, None, 20.0)
            self.p0 = 1.0 + 0.1 * rs.randn(self.N, self.D) if make_p0 else None
            self.moves, self.weights = [("stretch", std("stretch", 32))], [1.0]
            self.label = "configs[2]: nwalkers=%d, ndim=32, Rosenbrock/20, StretchMove a=2.0" % self.N
        elif key in ("hbm_dense", "w512", "w128"):
            # C2's target at other sizes: hbm_dense = 1 048 576 x 64 (537 MB of coordinates: past the 256 MB Infinity Cache);
            # w512 = 65 536 walkers on a 512-dimensional dense Gaussian (the MFMA-bound wide path, emx_wide.hip); w128 = on a
            # 128-dimensional one: the widest target the FUSED half-step kernel takes (round 3; the wide path before)
            self.D = {"hbm_dense": 64, "w512": 512, "w128": 128}[key]
            mu, cov, icov = dense_gaussian(self.D)
            self.params = (mu, cov, icov)
            self.target = (_lib.TARGET_DENSE, mu, icov, 0.0)
            self.p0 = mu + np.random.default_rng(1).standard_normal((self.N, self.D)) @ np.linalg.cholesky(cov).T if make_p0 else None
            self.moves, self.weights = [("stretch", std("stretch", self.D))], [1.0]
            self.label = "nwalkers=%d, ndim=%d, dense-precision Gaussian, StretchMove a=2.0" % (self.N, self.D)
        elif key in ("c5", "hbm_wide"):
            self.D = 1024
            ivar = 1.0 / np.random.RandomState(0).rand(self.D)                        # docs/index.rst:41-45
            self.target = (_lib.TARGET_DIAG, np.zeros(self.D), ivar, 0.0)
            if not make_p0:
                self.p0 = None
                self.label = ("configs[4]: " if key == "c5" else "") + "nwalkers=%d, ndim=1024, diagonal Gaussian, StretchMove a=2.0" % self.N
            elif key == "c5":
                self.p0 = rs.randn(self.N, self.D) / np.sqrt(ivar)
                self.label = "configs[4]: nwalkers=%d, ndim=1024, diagonal Gaussian, StretchMove a=2.0" % self.N
            else:       # 262 144 x 1024: 2.1 GB of coordinates, nothing of it cache resident
                self.p0 = np.random.default_rng(1).standard_normal((self.N, self.D)) / np.sqrt(ivar)
                self.label = "nwalkers=%d, ndim=1024, diagonal Gaussian, StretchMove a=2.0" % self.N
            self.moves, self.weights = [("stretch", std("stretch", 1024))], [1.0]
        else:
            raise ValueError(key)

    def bytes_per_update(self, store):
        w = np.asarray(self.weights) / np.sum(self.weights)
        return float(sum(wi * algorithmic_bytes(self.D, kind, store) for wi, (kind, _) in zip(w, self.moves)))

    def launches_per_step(self):
        w = np.asarray(self.weights) / np.sum(self.weights)
        return float(sum(wi * d.nsplits for wi, (_, d) in zip(w, self.moves)))

    def install(self, ens, rng, seed=20260923):
        from emcee_amd import _lib
        kind, p0, p1, scale = self.target
        ens.set_target(kind, p0, p1, scale)
        cdf = np.cumsum(self.weights) / np.sum(self.weights)
        ens.set_moves([d for _, d in self.moves], cdf)
        if rng == "philox":
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(seed, 0)
        else:
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(np.random.RandomState(seed).get_state())
        ens.set_state(self.p0)
        ens.eval_state_log_prob()


# ------------------------------------------------------------------------------------------------ stdout
# The contract is ONE JSON line on stdout.  Libraries in the process write there too (gloo announces its mesh, RCCL its
# version ...), so file descriptor 1 is pointed at stderr for the whole run and the line goes to a private copy of the
# original stdout.
_REAL_STDOUT = None


def _claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _REAL_STDOUT


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


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


# ------------------------------------------------------------------------------------------------ single-GPU measurement
def measure_single(wl, K, W, device=0, rng="philox", store=False, single_block=False, want_kernel=True, spin_s=0.15, tuning=None):
    """W warm-up steps, then K-step blocks (each: sync, hipEvent + wall clock around emx_run(K), sync) until >= 50 ms;
    median block.  Returns per-step times, per-launch event duration of the half-step kernel, accept fraction."""
    from emcee_amd.device import DeviceEnsemble
    ens = DeviceEnsemble(wl.N, wl.D, device=device)
    wl.install(ens, rng)
    for key, val in (tuning or {}).items():
        ens.set_tuning(key, val)
    if store:
        ens.chain_config(max(K, W))
    # untimed spin-up: the first ~50 ms on a fresh context run slower (clock ramp, first touch of the plan ring, lazy
    # code-object loading); tools/stall_probe.py
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < spin_s:
        ens.run(min(50, max(1, K)), 1, False)
        ens.sync()
    if store:
        ens.chain_reset()
    ens.run(W, 1, store)
    ens.sync()
    walls, gpus = [], []
    total = 0.0
    pinfo0 = ens.persist_info()
    while True:
        if store:
            ens.chain_reset()
        ens.sync()
        ens.timer_start()
        t0 = time.perf_counter()
        ens.run(K, 1, store)
        gpu_ms = ens.timer_stop()          # hipEvents on the stream the kernels are launched on; synchronises
        ens.sync()
        wall = time.perf_counter() - t0
        walls.append(wall)
        gpus.append(gpu_ms)
        total += wall * 1e3
        if single_block or (total >= MIN_TIMED_MS and len(walls) >= 3) or len(walls) >= MAX_BLOCKS:
            break
    wall = float(np.median(walls))
    gpu_ms = float(np.median(gpus))
    res = {"wall_s": wall, "gpu_ms": gpu_ms, "blocks": len(walls), "wall_min_s": float(np.min(walls)),
           "accept_frac": float(ens.accepted_mask().mean()), "status": ens.status(), "per_launch_us": None,
           "walls_s": [float(w) for w in walls], "halfsteps_per_launch": 1.0}
    pinfo1 = ens.persist_info()
    persistent = pinfo1["launches"] > pinfo0["launches"]
    if persistent:      # k_persist: several half-steps per launch (16 steps when a call is aligned with the plan batches)
        res["halfsteps_per_launch"] = (pinfo1["halfsteps"] - pinfo0["halfsteps"]) / float(pinfo1["launches"] - pinfo0["launches"])
    if rng == "mt19937":
        try:
            res["pipeline"] = ens.pipeline_stats()
        except Exception as e:  # noqa: BLE001
            log("pipeline stats unavailable:", e)
        try:
            res["mtdev"] = dict(ens.mtdev_info(), tokenizer=ens.mtdev_tok_stats())
        except Exception as e:  # noqa: BLE001
            log("device producer stats unavailable:", e)
    if want_kernel:
        # per-launch hipEvent durations of the half-step kernel (separate pass: event records perturb)
        if persistent and rng == "philox":
            seed, step = ens.get_philox()
            ens.set_philox(seed, step)          # forget the plans evaluated ahead: the launches below are whole 16-step batches
        pinfo2 = ens.persist_info()
        ens.profile_enable(128)
        ens.run(48, 1, False)
        pl = ens.profile_read(128)
        if len(pl):
            res["per_launch_us"] = float(np.median(pl) * 1e3)
            pinfo3 = ens.persist_info()
            if pinfo3["launches"] > pinfo2["launches"]:
                res["per_launch_halfsteps"] = (pinfo3["halfsteps"] - pinfo2["halfsteps"]) / float(pinfo3["launches"] - pinfo2["launches"])
    res["persist_total"] = ens.persist_info()
    ens.close()
    return res


def wide_entry(wl, res, K):
    """Dense Gaussian beyond ndim 112 (emx_wide.hip): propose -> k_wide_lp -> commit.  The MFMA log-prob kernel dominates and is
    bound by the f64 matrix pipe, not by HBM: D^2 + 3 D flop per walker-update in the Cholesky form, against
    24 D + 17 bytes."""
    D, N = wl.D, wl.N
    flops = float(D) * D + 3.0 * D
    ms = res["wall_s"] * 1e3 / K
    wu = N * K / res["wall_s"]
    out = {"workload": wl.label, "nwalkers": N, "ndim": D, "ms_per_step": ms, "wu_per_s": wu, "steps_per_s": K / res["wall_s"],
           "blocks_timed": res["blocks"], "accept_frac": res["accept_frac"], "device_status": res["status"]}
    rl = {"bound": "mfma_f64", "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "algorithmic_flops_per_walker_update": flops,
          "walker_updates_per_launch": N / 2.0, "kernel": "emx::k_wide_lp* (Y = R L by v_mfma_f64_16x16x4_f64, L streamed through LDS)",
          "frac_wall_clock": wu * flops / 1e12 / MFMA_F64_PEAK_TFLOPS,
          "hbm_frac_wall_clock": wu * wl.bytes_per_update(False) / 1e9 / HBM_PEAK_GBPS}
    if res["per_launch_us"]:
        rl["avg_launch_us"] = res["per_launch_us"]
        rl["achieved"] = (N / 2.0) * flops / (res["per_launch_us"] * 1e-6) / 1e12
        rl["frac"] = rl["achieved"] / MFMA_F64_PEAK_TFLOPS
        rl["note"] = "avg_launch_us = hipEvents around single k_wide_lp launches (median of 128); frac_wall_clock prices the WHOLE step " \
                     "(propose + log-prob + commit passes) against the matrix peak"
    out["roofline"] = rl
    return out


def config_entry(wl, res, K, store):
    B = wl.bytes_per_update(store)
    lps = wl.launches_per_step()
    ms = res["wall_s"] * 1e3 / K
    wu = wl.N * K / res["wall_s"]
    ev_ms = res["gpu_ms"] / K
    out = {"workload": wl.label + (", chain stored every step" if store else ""), "nwalkers": wl.N, "ndim": wl.D,
           "ms_per_step": ms, "wu_per_s": wu, "steps_per_s": K / res["wall_s"], "blocks_timed": res["blocks"],
           "accept_frac": res["accept_frac"], "device_status": res["status"],
           "roofline": {"bound": "hbm", "algorithmic_bytes_per_walker_update": B,
                        "achieved": wl.N * B / (ev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": wl.N * B / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "frac_wall_clock": wu * B / 1e9 / HBM_PEAK_GBPS,
                        "avg_launch_us": ev_ms * 1e3 / lps, "per_launch_event_us": res["per_launch_us"],
                        "launches_per_step": lps}}
    hpl = res.get("halfsteps_per_launch", 1.0)
    if hpl > 1.0 and len(wl.moves) == 1:
        # every step through the persistent kernel: a launch is hpl half-steps
        out["roofline"].update({"kernel": "emx::k_persist<8,2,4,DPB=4>: %.1f half-steps per launch" % hpl,
                                "avg_launch_us": ev_ms * 1e3 / lps * hpl, "launches_per_step": lps / hpl, "avg_halfstep_us": ev_ms * 1e3 / lps,
                                "per_launch_event_halfsteps": res.get("per_launch_halfsteps")})
    elif hpl > 1.0:
        # a mixture: the consecutive steps of one move share a persistent launch (k_persist<..., MOVE_DE / MOVE_SNOOKER>: two half-steps
        # per DE step, four per snooker step); avg_launch_us stays the timed region / half-steps
        out["roofline"]["kernel"] = ("emx::k_persist<8,2,4,DPB=4,MOVE_DE> and <...,MOVE_SNOOKER>: %.1f half-steps per launch (a run of consecutive "
                                     "steps of one move)" % hpl)
        out["roofline"]["avg_launch_us_is"] = "hipEvent time of the timed region / half-steps (persistent launches counted by their half-steps)"
    state_mb = wl.N * wl.D * 8 / 1e6
    traffic = None
    if state_mb > 256.0:
        traffic = hbm_traffic(wl.key)
        out["roofline"].update({"state_MB": state_mb, "beyond_infinity_cache": True,
                                "frac_of_achievable_6300": wl.N * B / (ev_ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBPS,
                                "traffic": traffic, "traffic_source": "profiles/pmc_traffic.json (static; rocprofv3 PMC passes)"})
    roofline_audit(out["roofline"], wl, store, res["accept_frac"], wl.N / (ev_ms * 1e-3), traffic, ev_ms * 1e-3 / lps)
    if state_mb > 256.0:
        out["roofline"]["frac_moved_of_achievable_6300"] = out["roofline"]["achieved_moved"] / HBM_ACHIEVABLE_GBPS
    return out


def hbm_traffic(key):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key + "_bytes_per_launch")
    except Exception:  # noqa: BLE001
        return None


def exact_mode_entry(wl, K, W, device):
    """C2 under rng=mt19937 (same seed => reference emcee's chain).  The host produces every draw of the step from the
    serial NumPy-legacy stream; host_plan_ms times that producer alone (no GPU involved)."""
    from emcee_amd import _lib
    Kx = max(100, min(K, 400))          # one emx_run per block: long enough that the pipeline's thread start-up (0.2 ms) is amortised
    res = measure_single(wl, Kx, max(W, 10), device=device, rng="mt19937", spin_s=0.05)
    lib = _lib.load()
    host_ms = None
    try:
        st = np.random.RandomState(5).get_state()
        key = np.ascontiguousarray(st[1], dtype=np.uint32)
        m = lib.emx_mt_create(key, int(st[2]), int(st[3]), float(st[4]))
        N = wl.N
        off = np.zeros(3, dtype=np.int32)
        order, p0, p1, p2 = (np.empty(N, dtype=np.int32) for _ in range(4))
        s0, ua = np.empty(N), np.empty(N)
        mv = wl.moves[0][1]
        import ctypes as C
        lib.emx_host_plan_mt(m, N, wl.D, C.byref(mv), off, order, p0, p1, p2, s0, ua)
        t0 = time.perf_counter()
        for _ in range(20):
            lib.emx_host_plan_mt(m, N, wl.D, C.byref(mv), off, order, p0, p1, p2, s0, ua)
        host_ms = (time.perf_counter() - t0) * 1e3 / 20
        lib.emx_mt_destroy(m)
    except Exception as e:  # noqa: BLE001
        log("host plan timing failed:", e)
    B = wl.bytes_per_update(False)
    wu = wl.N * Kx / res["wall_s"]
    return {"workload": wl.label + ", rng=mt19937 (NumPy legacy stream, chain identical to reference emcee's)",
            "steps": Kx, "blocks_timed": res["blocks"], "ms_per_step": res["wall_s"] * 1e3 / Kx, "wu_per_s": wu,
            "best_block_ms_per_step": res["wall_min_s"] * 1e3 / Kx,
            "host_plan_ms": host_ms, "kernel_us": res["per_launch_us"], "accept_frac": res["accept_frac"],
            "roofline_frac_wall_clock": wu * B / 1e9 / HBM_PEAK_GBPS,
            "block_spread": float(np.max(res["walls_s"]) / np.min(res["walls_s"])),
            "pipeline_stage_us_per_step": res.get("pipeline"),
            "note": "host_plan_ms = one step's plan made inline by ONE host thread (emx_host_plan_mt, no GPU) -- round 1's path; emx_run "
                    "now takes its plans from the host pipeline (csrc/emx_mtpipe.cpp: MT19937 generator thread, tokenizer thread, "
                    "finisher threads -- six where the L3 domain has room -- confined to that domain, uploads on a side stream), so ms_per_step is the pipeline's rate; "
                    "pipeline_stage_us_per_step says which stage bounds it on THIS host (the stages run concurrently: the largest of "
                    "generator / tokenizer / finishers-summed over the thread count is the pipeline's floor)"}


def exact_mode_large_entry(K, W, device):
    """rng=mt19937 at C3's size (262 144 x 32 Rosenbrock): the ensemble size from which the plans of the reference's own stream are
    made ON THE DEVICE (csrc/emx_mtdev.hpp: jump-ahead MT19937 segments, tokenizer and finisher kernels; no host thread touches a
    draw), next to the host pipeline on the same box (tuning mt_device = 0)."""
    wl = Workload("c3", 262144)
    Kx = max(50, min(K, 200))
    out = {"workload": wl.label + ", rng=mt19937 (NumPy legacy stream, chain identical to reference emcee's)", "steps": Kx}
    B = wl.bytes_per_update(False)
    for name, tune in (("device_producer", {"mt_device": 1}), ("host_pipeline", {"mt_device": 0})):
        res = measure_single(wl, Kx, max(W, 10), device=device, rng="mt19937", spin_s=0.05, want_kernel=False, tuning=tune)
        wu = wl.N * Kx / res["wall_s"]
        e = {"ms_per_step": res["wall_s"] * 1e3 / Kx, "wu_per_s": wu, "blocks_timed": res["blocks"], "device_status": res["status"],
             "roofline_frac_wall_clock": wu * B / 1e9 / HBM_PEAK_GBPS, "accept_frac": res["accept_frac"]}
        md = res.get("mtdev") or {}
        if name == "device_producer":
            e["host_threads"] = 0
            e["producer_used"] = bool(md.get("steps", 0) > 0)
            tk = md.get("tokenizer") or {}
            if tk.get("windows"):
                steps = max(1, md.get("steps", 1))
                e["tokenizer_us_per_step"] = tk["kernel_us"] / steps
                e["tokenizer_rounds_per_window"] = tk["rounds"] / float(tk["windows"])
        else:
            e["pipeline_stage_us_per_step"] = res.get("pipeline")
        out[name] = e
    if "ms_per_step" in out.get("device_producer", {}) and "ms_per_step" in out.get("host_pipeline", {}):
        out["speedup_device_over_host"] = out["host_pipeline"]["ms_per_step"] / out["device_producer"]["ms_per_step"]
    out["note"] = ("the tokenizer (the masked rejection of random.shuffle, red_blue.py:80: the one serial part of a step) bounds the device "
                   "producer; below ~10^5 walkers the host pipeline is faster and stays the default (profiles/r04/mtdev_sizes.txt)")
    return out


def quality_entry(device, rng="philox"):
    """Acceptance fraction and integrated autocorrelation time (reference estimator, c=5) of the 64-dim correlated
    Gaussian, StretchMove a=2, in the configuration the reference itself was run in (tests/golden/quality_ref.json, made by
    `tools/quality.py --ref` in the build container: 1024 walkers, 2000 burn-in + 100 000 steps = 68 tau, thin_by 25), next to
    the reference's numbers.  Different random streams: the comparison is statistical (2 % bar, BASELINE.json)."""
    import emcee_amd
    D = 64
    ref = None
    path = os.path.join(ROOT, "tests", "golden", "quality_ref.json")
    try:
        ref = json.load(open(path))
        cfg = ref["config"]
        nwalkers, nsteps, thin_by, burn = cfg["nwalkers"], cfg["nsteps"], cfg["thin_by"], cfg["burn"]
    except Exception:  # noqa: BLE001
        nwalkers, nsteps, thin_by, burn = 1024, 4000, 25, 2000
    mu, cov, icov = dense_gaussian(D)
    p0 = mu + np.random.RandomState(1).randn(nwalkers, D) @ np.linalg.cholesky(cov).T
    s = emcee_amd.EnsembleSampler(nwalkers, D, emcee_amd.targets.DenseGaussian(mu, icov), rng=rng, device=device)
    s._random.seed(12)
    t0 = time.perf_counter()
    st = s.run_mcmc(p0, burn, skip_initial_state_check=True, store=False)
    s.run_mcmc(st, nsteps, thin_by=thin_by, skip_initial_state_check=True)
    t_run = time.perf_counter() - t0
    t0 = time.perf_counter()
    tau = np.asarray(s.get_autocorr_time(quiet=True)) * thin_by          # the backend counts in stored samples
    t_tau = time.perf_counter() - t0
    acc = float(np.mean(s.acceptance_fraction))
    out = {"workload": "%d walkers x 64-dim correlated Gaussian, StretchMove a=2, %d burn-in + %d steps, thin_by=%d, rng=%s"
                       % (nwalkers, burn, nsteps * thin_by, thin_by, rng),
           "accept": acc, "tau_mean": float(np.mean(tau)), "tau_min": float(np.min(tau)), "tau_max": float(np.max(tau)),
           "nsteps_over_tau": float(nsteps * thin_by / np.mean(tau)), "run_seconds": t_run, "tau_seconds": t_tau}
    if ref is not None:
        r = ref["results"][0]
        out["reference"] = {"source": "tests/golden/quality_ref.json (reference emcee in the build container, tools/quality.py --ref; static)",
                            "accept": r["accept_mean"], "tau_mean": r["tau_mean"], "nsteps_over_tau": r["chain_over_tau"],
                            "seconds": r["seconds"]}
        out["accept_rel_diff"] = acc / r["accept_mean"] - 1.0
        out["tau_rel_diff"] = float(np.mean(tau)) / r["tau_mean"] - 1.0
        out["within_2pct"] = bool(abs(out["accept_rel_diff"]) < 0.02 and abs(out["tau_rel_diff"]) < 0.02)
    return out


# ------------------------------------------------------------------------------------------------ multi-GPU measurement
EXCHANGES = ("allgather", "pull", "direct", "replay", "replay_push")      # measured by default at N > 1
# "logprob" (proposal / commit replicated, log-prob evaluations shared out: for targets that dominate the step) is measured
# on request only: on the closed-form BASELINE targets the replicated part is most of the step
ALL_EXCHANGES = EXCHANGES + ("logprob",)
# the compute-heavy configuration (65 536 x 512 dense per GPU, weak scaling): the protocols that share out the evaluation
HEAVY_EXCHANGES = ("replay", "replay_push", "logprob")


_NCCL_GROUP = {}


def _torch_nccl_group(dist):
    """torch.distributed's own RCCL communicator, next to the gloo bootstrap group (fallback data path)"""
    if "g" not in _NCCL_GROUP:
        _NCCL_GROUP["g"] = dist.new_group(backend="nccl")
    return _NCCL_GROUP["g"]


def measure_sharded(wl, K, W, exchange, rank, world, local_rank, dist, comm_mode, single_block=False, direct_timeout_ms=None):
    """One sharded measurement (fresh context): spin-up, W warm-up steps, K-step blocks."""
    import torch
    from emcee_amd.device import DeviceEnsemble
    ens = DeviceEnsemble(wl.N, wl.D, device=local_rank)
    wl.install(ens, "philox")
    push = exchange == "replay_push"          # the replay exchange with the decisions stored into the peers' buffers (no collective)
    if push:
        exchange = "replay"
    ens.set_exchange(exchange)
    if direct_timeout_ms:
        ens.set_tuning("direct_timeout_ms", int(direct_timeout_ms))
    comm_used = None
    if comm_mode == "torch" and not push:
        if exchange == "direct":
            raise RuntimeError("the direct exchange is driven by libemx itself (--comm rccl)")
        from emcee_amd.parallel import DeviceEngine, PullStepper, ShardedStepper
        ens.set_stream(torch.cuda.current_stream().cuda_stream)   # kernels + RCCL ordered on one stream
        eng = DeviceEngine(ens, rank, world, torch.device("cuda", local_rank), exchange=exchange)
        grp = None if dist.get_backend() == "nccl" else _torch_nccl_group(dist)
        gather = lambda out, inp: dist.all_gather_into_tensor(out, inp, group=grp)  # noqa: E731
        if exchange == "logprob":
            from emcee_amd.parallel import LogProbStepper
            stepper = LogProbStepper(eng, gather)
        elif exchange == "replay":
            from emcee_amd.parallel import ReplayStepper
            stepper = ReplayStepper(eng, gather)
        elif exchange == "pull":
            stepper = PullStepper(eng, lambda out, inp: dist.all_to_all_single(out, inp, group=grp), gather)
        else:
            stepper = ShardedStepper(eng, gather)
        run = lambda k: stepper.run(k, 1, False)  # noqa: E731
        comm_used = "torch.distributed(nccl)"
    elif push:
        # no collective library anywhere on this path: the ranks map each other's receive buffers and barrier flags (hipIpc
        # handles over the gloo bootstrap group) and emx_run exchanges the decisions with plain stores + the device-side barrier
        from emcee_amd.parallel import import_direct_peers
        ens.set_shard(rank, world)
        import_direct_peers(ens, dist)
        run = lambda k: ens.run(k, 1, False)  # noqa: E731
        comm_used = "hipIpc stores + device-side barrier (no collective library)"
    else:
        uid = [DeviceEnsemble.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ens.comm_init(rank, world, uid[0])      # ncclCommInitRank; emx_run now exchanges per half-step
        if exchange == "direct":                # map the peers' coordinate arrays and barrier flags (IPC handles over gloo)
            from emcee_amd.parallel import import_direct_peers
            import_direct_peers(ens, dist)
        run = lambda k: ens.run(k, 1, False)  # noqa: E731
        comm_used = "libemx->RCCL"

    def fence():
        ens.sync()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    dist.barrier()                # the ranks enter the first step together (the device-side barriers are bounded, not patient)
    for _ in range(6):            # a FIXED count: every rank must issue the same collectives
        run(5)
        ens.sync()
    run(W)
    fence()
    walls, gpus = [], []
    total, nblk = 0.0, 0
    digest, every = None, None
    while True:
        fence()
        ens.timer_start()
        t0 = time.perf_counter()
        run(K)
        gpu_ms = ens.timer_stop()
        fence()
        wall = time.perf_counter() - t0
        t = torch.tensor([wall, gpu_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # gloo (bootstrap group) or nccl: both fine for 2 doubles
        walls.append(float(t[0]))
        gpus.append(float(t[1]))
        total += float(t[0]) * 1e3
        nblk += 1
        if digest is None:
            # checksum of the ensemble after a FIXED number of steps (spin-up + W + K): every rank and every exchange
            # protocol must arrive at the same state
            x, lp = ens.get_state()
            digest = "%.17g/%.17g" % (float(np.sum(x * np.arange(1, wl.D + 1))), float(np.sum(lp)))
            every = [None] * world
            dist.all_gather_object(every, digest)
        if single_block or (total >= MIN_TIMED_MS and nblk >= 3) or nblk >= 60:     # same decision on every rank: t is reduced
            break
    res = {"wall_s": float(np.median(walls)), "gpu_ms": float(np.median(gpus)), "blocks": nblk, "comm": comm_used,
           "exchange": "replay_push" if push else exchange, "accept_frac": float(ens.accepted_mask().mean()), "status": ens.status(),
           "digest": digest, "replicas_agree": len(set(every)) == 1}
    res.update(_rank_census(ens, dist, "peers" if push else comm_mode, local_rank))
    if comm_mode != "torch" and not push:
        ens.comm_destroy()
    if push:
        dist.barrier()                          # nobody unmaps while a peer may still store into its buffers
    ens.close()
    return res


def _rank_census(ens, dist, comm_mode, local_rank):
    """How many ranks the communicator that carried the exchange really has (ncclCommCount of libemx's communicator, or the
    torch process group's size) and how many DISTINCT devices the ranks sit on: n_gpus = N is only claimed when both say N."""
    import torch
    try:
        if comm_mode == "peers":               # device-side replay exchange: the ranks whose buffers this rank mapped (itself included)
            ranks = dist.get_world_size()
        else:
            ranks = ens.comm_count() if comm_mode != "torch" else dist.get_world_size()
    except Exception as e:  # noqa: BLE001
        log("comm_count failed:", e)
        ranks = None
    try:
        p = torch.cuda.get_device_properties(local_rank)
        ident = "%s/%s" % (getattr(p, "uuid", None), "%x:%x:%x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0),
                                                                    getattr(p, "pci_device_id", 0)))
    except Exception:  # noqa: BLE001
        ident = "device%d" % local_rank
    every = [None] * dist.get_world_size()
    dist.all_gather_object(every, ident)
    return {"rccl_ranks": ranks, "distinct_devices": len(set(every))}


class TinyWorkload(Workload):
    """Preflight: a few thousand walkers, isotropic Gaussian, StretchMove -- one millisecond of work per protocol."""

    def __init__(self, world):
        from emcee_amd import _lib
        self.key = "preflight"
        self.N, self.D = 4096 * world, 16
        self.target = (_lib.TARGET_ISO, None, None, 0.0)
        self.p0 = np.random.RandomState(3).randn(self.N, self.D)
        self.moves = [("stretch", _lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.4, 1.7))]
        self.weights = [1.0]
        self.label = "preflight: %d x %d isotropic Gaussian" % (self.N, self.D)


def preflight_child(args, rank, world, local_rank, dist):
    """`--child preflight:<exchange,...>`: first contact with the node, seconds per item instead of a 120 s watchdog each.
    Checks, in order: peer access between the devices, then every exchange protocol asked for on a tiny ensemble (8 steps,
    ensembles compared across the ranks).  One line per finished item goes out immediately, so a hang is attributed to the item
    in flight."""
    import torch
    out = _claim_stdout()

    def say(item, verdict):
        out.write("EMX_PREFLIGHT %s %s\n" % (item, json.dumps(verdict)))
        out.flush()

    res = {}
    try:
        ndev = torch.cuda.device_count()
        peers = [bool(torch.cuda.can_device_access_peer(local_rank, q)) for q in range(min(ndev, world)) if q != local_rank] \
            if args.all_on_device is None else []
        res["p2p"] = {"ok": all(peers), "devices_visible": ndev, "peer_access": peers}
    except Exception as e:  # noqa: BLE001
        res["p2p"] = {"ok": False, "error": repr(e)}
    say("p2p", res["p2p"])
    wl = TinyWorkload(world)
    for ex in args.child.split(":", 1)[1].split(","):
        if not ex:
            continue
        t0 = time.perf_counter()
        try:
            r = measure_sharded(wl, 8, 2, ex, rank, world, local_rank, dist, args.comm, single_block=True, direct_timeout_ms=2000)
            ok = r["status"] == 0 and r["replicas_agree"]
            res[ex] = {"ok": bool(ok), "seconds": time.perf_counter() - t0, "device_status": r["status"], "replicas_agree": r["replicas_agree"],
                       "digest": r["digest"], "rccl_ranks": r.get("rccl_ranks"), "distinct_devices": r.get("distinct_devices")}
        except Exception as e:  # noqa: BLE001
            res[ex] = {"ok": False, "seconds": time.perf_counter() - t0, "error": repr(e)[:300]}
        allok = torch_all_ok(dist, res[ex]["ok"])
        if not allok and res[ex]["ok"]:
            res[ex] = {"ok": False, "error": "failed on another rank"}
        say(ex, res[ex])
    return res


def run_preflight(args, world, dist, port0, exchanges):
    """-> {item: verdict}.  A child that hangs is killed after --preflight-timeout; what it had finished counts, the item in
    flight is marked failed and the rest is tried again in a fresh child."""
    verdicts = {}
    todo = list(exchanges)
    attempt = 0
    while True:
        r = run_child(args, "preflight", ",".join(todo), port0 + attempt, args.preflight_timeout + (180.0 if attempt == 0 else 0.0),
                      keep_partial=True)
        attempt += 1
        done = r.get("preflight", {}) if isinstance(r, dict) else {}
        for k, v in done.items():
            verdicts.setdefault(k, v)
        left = [e for e in todo if e not in verdicts]
        # every rank must take the same decision: agree on the shortest list of finished items
        n_done = len(todo) - len(left)
        import torch
        t = torch.tensor([n_done])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        n_done = int(t[0])
        for e in todo[n_done:]:
            verdicts.pop(e, None)
        left = todo[n_done:]
        if not left or attempt >= 4:
            for e in left:
                verdicts[e] = {"ok": False, "error": "not reached"}
            break
        verdicts[left[0]] = {"ok": False, "error": r.get("error") or "hung or crashed during the preflight (child killed)"}
        todo = left[1:]
        if not todo:
            break
    return verdicts


def sharded_workload(key, world, args, make_p0=True):
    scaling = {"c2": "weak", "c3": "strong", "c5": "strong", "w512": "weak"}[key] if args.scaling == "auto" else args.scaling
    base = {"c2": 65536, "c3": 262144, "c5": 16384, "w512": 65536}[key]
    return Workload(key, base * world if scaling == "weak" else base, make_p0=make_p0), scaling


def child_main(args, rank, world, local_rank):
    """One (configuration, exchange) measurement in a process of its own: a protocol that crashes the GPU runtime or hangs in a
    collective takes this child with it, not the rank's orchestrating parent (which never touches the GPU at N > 1)."""
    import torch
    import torch.distributed as dist
    key, ex = args.child.split(":", 1)
    torch.cuda.set_device(local_rank)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % args.child_port, rank=rank, world_size=world)
    if key == "preflight":
        out = {"preflight": preflight_child(args, rank, world, local_rank, dist)}
        _claim_stdout().write("EMX_CHILD_RESULT " + json.dumps(out) + "\n")
        _claim_stdout().flush()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        return
    wl, _ = sharded_workload(key, world, args)
    out = {"error": None}
    try:
        out = measure_sharded(wl, args.steps, args.warmup, ex, rank, world, local_rank, dist, args.comm, args.single_block)
    except Exception as e:  # noqa: BLE001
        first = repr(e)
        log("rank %d: exchange '%s' on %s failed: %s" % (rank, ex, key, first))
        out = {"error": first}
    if args.comm == "rccl" and ex not in ("direct", "replay_push"):
        # library-driven RCCL unavailable on some rank: the same protocol over torch.distributed's communicator
        if not torch_all_ok(dist, out.get("error") is None):
            try:
                out = measure_sharded(wl, args.steps, args.warmup, ex, rank, world, local_rank, dist, "torch", args.single_block)
            except Exception as e:  # noqa: BLE001
                out = {"error": (out.get("error") or "failed on another rank") + " | torch.distributed fallback: " + repr(e)}
    sys.stdout.flush()
    _claim_stdout().write("EMX_CHILD_RESULT " + json.dumps(out) + "\n")
    _claim_stdout().flush()
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


def _partial_preflight(text):
    done = {}
    for line in (text or "").splitlines():
        if line.startswith("EMX_PREFLIGHT "):
            try:
                _, item, verdict = line.split(" ", 2)
                done[item] = json.loads(verdict)
            except Exception:  # noqa: BLE001
                pass
    return done


def run_child(args, key, ex, port, timeout_s, keep_partial=False):
    """-> the child's result dict, or {"error": ...} (non-zero exit, no result line, or the timeout)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--child", "%s:%s" % (key, ex), "--child-port", str(port), "--comm", args.comm, "--scaling", args.scaling]
    if args.single_block:
        cmd.append("--single-block")
    if args.all_on_device is not None:
        cmd += ["--all-on-device", str(args.all_on_device)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=None, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired as e:
        out = {"error": "no result within %.0f s (hung; child killed)" % timeout_s}
        if keep_partial:
            txt = e.stdout.decode() if isinstance(e.stdout, bytes) else e.stdout
            out["preflight"] = _partial_preflight(txt)
        return out
    for line in (r.stdout or "").splitlines():
        if line.startswith("EMX_CHILD_RESULT "):
            try:
                return json.loads(line[len("EMX_CHILD_RESULT "):])
            except Exception as e:  # noqa: BLE001
                return {"error": "unreadable child result: %r" % (e,)}
    out = {"error": "child exited with code %d and no result" % r.returncode}
    if keep_partial:
        out["preflight"] = _partial_preflight(r.stdout)
    return out


_CHILDREN_RUN = []
_DEADLINE = [None]          # N > 1: perf_counter value by which the orchestrator wants to be done (--time-budget)


def agreed_remaining(dist):
    """seconds left of the time budget, the same number on every rank (the minimum over their clocks); None without a budget"""
    if _DEADLINE[0] is None:
        return None
    import torch
    t = torch.tensor([_DEADLINE[0] - time.perf_counter()], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t[0])


def worst_case_seconds(args, keys, exchanges):
    """what the watchdogs alone would allow: every preflight attempt and every (configuration, exchange) child running into its
    timeout -- the number the time budget exists to cut down"""
    pre = 0.0 if args.no_preflight else (args.preflight_timeout + 180.0) + 3 * args.preflight_timeout
    total = 0.0
    first = True
    for key in keys:
        for ex in ((HEAVY_EXCHANGES if key == "w512" else EXCHANGES) if args.exchange == "all" else (args.exchange,)):
            total += args.exchange_timeout + (180.0 if first else 0.0) + (180.0 if key == "w512" else 0.0)
            first = False
    return {"preflight_s": pre, "measurements_s": total, "unbounded_s": pre + total, "time_budget_s": args.time_budget,
            "note": "unbounded = every watchdog firing (a protocol that failed once is not tried again, so at most one timeout per "
                    "protocol in practice); the orchestrator stops starting children once the budget is spent and says what it skipped"}


XGMI_INGRESS_GBPS = 7 * 76.8        # MI355X: 7 links x 153.6 GB/s bidirectional = 76.8 GB/s per direction each (MI355X_MICROARCH.md)

# DESIGN.md section 6, "What to expect": microseconds per step of the protocol expected to win, written down BEFORE the first
# multi-GPU run so that the first curve can be read against a prediction (world size -> us/step)
PREDICTED_US_PER_STEP = {      # the device-side replay exchange (replay_push); the model and its inputs are in DESIGN.md section 6
    "c2": {2: 42.0, 4: 50.0, 8: 62.0},          # weak, 65 536 walkers per GPU: 1.1x / 1.9x / 3.1x one GPU's per-half-step 23.7 us
                                                # (1.0x / 1.7x / 2.7x the persistent kernel's 20.8 us, which the sharded paths do not use)
    "c3": {2: 44.0, 4: 41.0, 8: 40.0},          # strong, 38.6 us on one GPU: no G > 1 is expected to be faster (0.9-0.97x)
    "c5": {2: 57.0, 4: 44.0, 8: 39.0},          # strong, 53.6 us on one GPU: 0.94x / 1.2x / 1.4x
    "w512": {2: 536.0, 4: 558.0, 8: 600.0},     # weak, 504 us on one GPU: 1.9x / 3.6x / 6.7x -- the workload that reaches 6x
}


def xgmi_bytes_per_update(wl, ex, world, accept_frac=None):
    """Bytes a GPU receives over xGMI per walker-update of the whole ensemble's step."""
    G, D = world, wl.D
    w = np.asarray(wl.weights) / np.sum(wl.weights)
    npart = float(sum(wi * partner_rows(kind) for wi, (kind, _) in zip(w, wl.moves)))
    if ex == "allgather":
        return (G - 1) * 8.0 * (D + 2)
    if ex == "pull":
        return npart * (G - 1) / G * 8.0 * (D + 1) * 1.2
    if ex == "direct":
        return npart * (G - 1) / G * 8.0 * D
    if ex in ("logprob", "replay", "replay_push"):
        return (G - 1) * 8.0
    return float("nan")


def sharded_config(key, world, K, rank, dist, args, port0, skip):
    """Every exchange protocol on one workload, each in its own child process per rank; the fastest whose final ensemble
    agrees on all ranks (and with the first protocol's) is reported.  `skip`: protocols that already failed on an earlier
    configuration (not tried again)."""
    wl, scaling = sharded_workload(key, world, args, make_p0=False)       # the orchestrator only needs the description
    results, errors = {}, {}
    exchanges = (HEAVY_EXCHANGES if key == "w512" else EXCHANGES) if args.exchange == "all" else (args.exchange,)
    for n, ex in enumerate(exchanges):
        if ex in skip:
            errors[ex] = "skipped: failed on an earlier configuration (%s)" % skip[ex]
            continue
        # the very first child also pays for cold caches (kernel modules, code objects, a slower first torch import)
        first = not _CHILDREN_RUN
        # (the weak-scaled 512-dimensional ensemble is 2 GB of start state per rank to generate and upload: give it time)
        timeout_s = args.exchange_timeout + (180.0 if first else 0.0) + (180.0 if key == "w512" else 0.0)
        rem = agreed_remaining(dist)
        if rem is not None:
            if rem < 30.0:
                errors[ex] = "skipped: the run's time budget (--time-budget %.0f s) is spent" % args.time_budget
                continue
            timeout_s = min(timeout_s, rem - 10.0)
        _CHILDREN_RUN.append((key, ex))
        r = run_child(args, key, ex, port0 + n, timeout_s)
        ok = r.get("error") is None and "wall_s" in r
        if not torch_all_ok(dist, ok):             # the parents' own gloo group: CPU only
            errors[ex] = r.get("error") or "failed on another rank"
            skip[ex] = "%s: %s" % (key, errors[ex][:200])
            log("rank %d: exchange '%s' on %s: %s" % (rank, ex, key, errors[ex]))
        else:
            results[ex] = r
    ref_digest = None
    best = None
    summary = {}
    for ex in exchanges:
        r = results.get(ex)
        if r is None:
            summary[ex] = {"error": errors.get(ex, "?")}
            continue
        if ref_digest is None:
            ref_digest = r["digest"]
        census_ok = args.all_on_device is not None or (r.get("rccl_ranks") == world and r.get("distinct_devices") == world)
        valid = r["status"] == 0 and r["replicas_agree"] and r["digest"] == ref_digest and census_ok
        wu = wl.N * K / r["wall_s"]
        xb = xgmi_bytes_per_update(wl, ex, world, r.get("accept_frac"))
        B = wl.bytes_per_update(False)
        summary[ex] = {"ms_per_step": r["wall_s"] * 1e3 / K, "wu_per_s": wu, "comm": r["comm"],
                       "device_status": r["status"], "replicas_agree": r["replicas_agree"],
                       "same_final_state_as_first": r["digest"] == ref_digest, "blocks_timed": r["blocks"],
                       "rccl_ranks": r.get("rccl_ranks"), "distinct_devices": r.get("distinct_devices"),
                       "roofline_frac_per_gpu": wu * B / 1e9 / HBM_PEAK_GBPS / world,
                       # bytes every GPU RECEIVES over xGMI per step, and the rate that is against the 7-link ingress cap
                       "xgmi_bytes_per_walker_update": xb, "xgmi_bytes_per_step_per_gpu": xb * wl.N / world,
                       "xgmi_ingress_GBps_per_gpu": xb * wu / world / 1e9,
                       "xgmi_ingress_frac_of_cap": xb * wu / world / 1e9 / XGMI_INGRESS_GBPS}
        if not census_ok:
            summary[ex]["error"] = "rank census failed: %s RCCL ranks on %s distinct devices, expected %d" % (
                r.get("rccl_ranks"), r.get("distinct_devices"), world)
        if valid and (best is None or r["wall_s"] < best["wall_s"]):
            best = r
    entry = {"workload": wl.label, "nwalkers": wl.N, "ndim": wl.D, "scaling": scaling, "exchange": summary}
    pred = PREDICTED_US_PER_STEP.get(key, {}).get(world)
    if pred:
        entry["predicted_us_per_step"] = {"value": pred, "source": "DESIGN.md section 6 (written before any N>1 run)"}
    if best is not None:
        B = wl.bytes_per_update(False)
        wu = wl.N * K / best["wall_s"]
        if key == "w512":      # MFMA-bound: D^2 + 3 D flop per walker-update against the f64 matrix peak
            entry["mfma_frac_per_gpu"] = wu * (float(wl.D) ** 2 + 3.0 * wl.D) / 1e12 / MFMA_F64_PEAK_TFLOPS / world
        entry.update({"reported": best["exchange"], "ms_per_step": best["wall_s"] * 1e3 / K, "wu_per_s": wu,
                      "steps_per_s": K / best["wall_s"], "accept_frac": best["accept_frac"],
                      "rccl_ranks": best.get("rccl_ranks"), "distinct_devices": best.get("distinct_devices"),
                      "roofline_frac_per_gpu": wu * B / 1e9 / HBM_PEAK_GBPS / world})
    return wl, best, entry


def torch_all_ok(dist, ok):
    import torch
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag[0]) == 1



# ------------------------------------------------------------------------------------------------ --pmc
def refresh_pmc_traffic(args):
    """`--pmc`: HBM traffic of the headline kernel measured NOW instead of read from profiles/pmc_traffic.json -- this command is
    re-run under rocprofv3 with FETCH_SIZE and with WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md, HBM section: the two
    counters are not collected together; FETCH_SIZE is doubled on gfx950), medians per launch of the stretch / dense half-step.
    -> (bytes per launch -- per half-step when the kernel is the persistent one --, source text, per_halfstep) or (None, reason, False)."""
    import collections
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if not prof:
        return None, "rocprofv3 not on PATH", False
    med = {}
    per_halfstep = False
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="emx_pmc_")
        cmd = [prof, "--pmc", ctr, "-d", d, "-o", "p", "-f", "csv", "--", sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "10",
               "--warmup", "2", "--no-cpu-baseline", "--no-extras"]
        env = dict(os.environ)
        env.setdefault("TMPDIR", "/tmp")
        try:
            cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, env=env, cwd=ROOT)
            child = None
            for ln in cp.stdout.decode(errors="replace").splitlines():
                if ln.startswith("{") and '"metric"' in ln:
                    child = json.loads(ln)
            ptot = (child or {}).get("persist") or {}
            agg = collections.defaultdict(list)
            persist_sum = 0.0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != ctr:
                        continue
                    if "k_halfstep<8, 2, 4, 0, 4, 1>" in r["Kernel_Name"]:
                        agg[r.get("Grid_Size", "")].append(float(r["Counter_Value"]))
                    elif "k_persist<" in r["Kernel_Name"]:
                        persist_sum += float(r["Counter_Value"])
            if ptot.get("halfsteps", 0) > 0 and persist_sum > 0.0:
                # the persistent kernel: launches run different numbers of half-steps, so the sum over every launch of the
                # process / the half-steps they ran (the child's own count), per HALF-STEP
                med[ctr] = persist_sum / float(ptot["halfsteps"])
                per_halfstep = True
                continue
            vals = max(agg.values(), key=len) if agg else []
            if len(vals) < 8:
                return None, "rocprofv3 --pmc %s produced no samples of the half-step kernel" % ctr, False
            med[ctr] = statistics.median(vals)
        except Exception as e:  # noqa: BLE001
            return None, "rocprofv3 --pmc %s failed: %r" % (ctr, e), False
        finally:
            shutil.rmtree(d, ignore_errors=True)
    nbytes = (2.0 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024.0
    return nbytes, ("measured by this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of this command, %s: "
                    "%.1f KB / %.1f KB; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, FETCH_SIZE doubled per the gfx950 correction)"
                    % ("sum over the k_persist launches / the half-steps they ran, i.e. per HALF-STEP" if per_halfstep else "medians per launch",
                       med["FETCH_SIZE"], med["WRITE_SIZE"])), per_halfstep


# ------------------------------------------------------------------------------------------------ self-launch (N > 1)
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def error_line(args, msg, extra=None):
    line = {"metric": "walker-updates/sec (whole node), 64-dim correlated Gaussian, StretchMove a=2", "value": None,
            "unit": "walker-updates/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "error": msg}
    if extra:
        line.update(extra)
    return line


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks HERE, one process per GPU, with the
    environment torch.distributed.run would have given them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), and pass
    rank 0's single JSON line through.  This process never touches a GPU.  Under torchrun (WORLD_SIZE already set) main() takes
    the rank path directly, so both ways of starting an N-GPU run execute the same code."""
    import subprocess
    out = _claim_stdout()
    N = args.gpus
    ndev = None
    if args.all_on_device is None and not os.environ.get("EMX_BENCH_STUB"):
        try:
            from emcee_amd import _lib
            ndev = _lib.device_count()
        except Exception as e:  # noqa: BLE001
            log("device count unavailable:", e)
        if ndev is not None and ndev < N:
            out.write(json.dumps(error_line(args, "--gpus %d but only %d HIP device(s) are visible" % (N, ndev),
                                            {"devices_visible": ndev})) + "\n")
            out.flush()
            return 2
    port = _free_port()
    env0 = dict(os.environ)
    env0.update({"WORLD_SIZE": str(N), "LOCAL_WORLD_SIZE": str(N), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                 "EMX_BENCH_SELF_LAUNCHED": "1"})
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL and hipIpc between processes need it
    procs = []
    for r in range(N):
        env = dict(env0)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
        # rank 0's stdout carries the line; the other ranks' goes to stderr (they print nothing there by contract)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None, text=(r == 0),
                                      start_new_session=True))
    log("self-launch: %d ranks (pids %s), rendezvous 127.0.0.1:%d" % (N, [p.pid for p in procs], port))
    deadline = time.time() + args.launch_timeout
    line0 = None
    try:
        try:
            line0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            line0 = None
        for p in procs[1:]:
            try:
                p.wait(timeout=max(1.0, min(60.0, deadline - time.time())))
            except subprocess.TimeoutExpired:
                pass
    finally:
        for p in procs:                       # exactly the process groups started above
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 9)
                except Exception:  # noqa: BLE001
                    pass
    rcs = [p.returncode for p in procs]
    text = [ln for ln in (line0 or "").splitlines() if ln.strip().startswith("{")]
    if not text:
        out.write(json.dumps(error_line(args, "the ranks produced no result line (exit codes %s%s)"
                                        % (rcs, "; timed out after %.0f s" % args.launch_timeout if line0 is None else ""))) + "\n")
        out.flush()
        return 1
    try:
        line = json.loads(text[-1])
        line["launcher"] = "bench.py self-launch: %d rank processes, one per GPU (no torchrun around it)" % N
        out.write(json.dumps(line) + "\n")
    except Exception:  # noqa: BLE001
        out.write(text[-1] + "\n")
    out.flush()
    return 0 if all(rc == 0 for rc in rcs) else 1


# ------------------------------------------------------------------------------------------------ main
def emit_line(line):
    out = _claim_stdout()
    out.write(json.dumps(line) + "\n")
    out.flush()


def main(argv=None):
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", default="all", choices=["all", "c2", "c3", "c4", "c5", "hbm_dense", "hbm_wide", "w512", "w128"],
                    help="which BASELINE configuration(s) to measure beside the C2 headline (N>1: c2, c3, c5)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="N>1: auto = C2 weak (65536 walkers per GPU), C3 / C5 strong (BASELINE's fixed totals)")
    ap.add_argument("--store", action="store_true", help="headline with the chain appended every step (32D+25 bytes)")
    ap.add_argument("--rng", default="philox", choices=["philox", "mt19937"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (no configs / exact_mode / quality)")
    ap.add_argument("--pmc", action="store_true",
                    help="N=1: re-measure roofline.traffic (HBM bytes per launch) with two rocprofv3 --pmc passes of this command "
                         "instead of reading profiles/pmc_traffic.json")
    ap.add_argument("--single-block", action="store_true", help="time one K-step block instead of the median of >= 50 ms of blocks")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded RCCL path even at world size 1 (testing)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="sharded runs: collectives enqueued by libemx itself (default) or torch.distributed")
    ap.add_argument("--exchange", default="all", choices=["all"] + list(ALL_EXCHANGES))
    ap.add_argument("--all-on-device", type=int, default=None,
                    help="testing on a one-GPU box: every rank uses this device (RCCL refuses duplicate devices, so only the "
                         "control flow -- failure handling, watchdogs, the emitted line -- is exercised)")
    ap.add_argument("--exchange-timeout", type=float, default=120.0,
                    help="seconds after which a stuck exchange measurement (a child process per rank) is killed")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="--gpus N > 1 started without torchrun: seconds after which the rank processes started here are killed")
    ap.add_argument("--preflight-timeout", type=float, default=45.0,
                    help="N > 1: seconds a preflight child (peer access + every exchange on a tiny ensemble) may take")
    ap.add_argument("--time-budget", type=float, default=840.0,
                    help="N > 1: wall-clock seconds the whole run may take; what does not fit (the later configurations' slower "
                         "protocols) is skipped and named in the line -- the watchdogs alone would allow far more")
    ap.add_argument("--no-preflight", action="store_true")
    ap.add_argument("--preflight", action="store_true", help="N > 1: run the preflight only and print its verdicts")
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)          # internal: "<config>:<exchange>"
    ap.add_argument("--child-port", type=int, default=0, help=argparse.SUPPRESS)
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)

    stub = os.environ.get("EMX_BENCH_STUB")       # tests only: a module whose install(bench) replaces the GPU legs
    if stub:
        import importlib
        importlib.import_module(stub).install(sys.modules[__name__])

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.child:
        # started as plain `python bench.py --gpus N`: this process becomes the launcher of the N ranks
        return self_launch(args, argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.all_on_device is not None:
        local_rank = args.all_on_device
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # the line would claim n_gpus = --gpus while measuring WORLD_SIZE ranks: refuse instead of printing a wrong number
        if rank == 0:
            emit_line(error_line(args, "--gpus %d but WORLD_SIZE=%d: start one rank per GPU (torch.distributed.run "
                                       "--nproc-per-node %d) or run bench.py without a launcher" % (args.gpus, world, args.gpus)))
        return 2
    K, W = args.steps, args.warmup
    if args.child:
        return child_main(args, rank, world, local_rank)

    sharded = world > 1 or args.force_dist
    if not sharded:
        import torch
        torch.cuda.set_device(local_rank)
    # HBM traffic of the headline kernel: per launch of k_halfstep, per HALF-STEP of k_persist (whose launches differ in length)
    traffic_static = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic_static = json.load(open(tpath))
        except Exception:  # noqa: BLE001
            traffic_static = {}
    traffic_fresh = None          # (bytes, per_halfstep)
    traffic_source = ("profiles/pmc_traffic.json (static: rocprofv3 PMC passes of an earlier run of this command, FETCH_SIZE x2 gfx950 "
                      "correction + WRITE_SIZE; not re-measured here; `bench.py --pmc` re-measures)")
    if args.pmc and not sharded:
        fresh, why, per_hs = refresh_pmc_traffic(args)
        if fresh is not None:
            traffic_fresh, traffic_source = (fresh, per_hs), why
        else:
            log("--pmc:", why)
            traffic_source += " [--pmc failed: %s]" % why

    def traffic_per_launch(hpl):
        if traffic_fresh is not None:
            nbytes, per_hs = traffic_fresh
            return nbytes * hpl if per_hs else (nbytes if hpl == 1.0 else None)
        if hpl > 1.0:
            per_hs = traffic_static.get("c2_persist_bytes_per_halfstep")
            return per_hs * hpl if per_hs else None
        return traffic_static.get("c2_stretch_dense_bytes_per_launch")

    def headline(wl, wall_s, gpu_ms, per_launch_us, accept, status, how, extra, hpl=1.0, event_hpl=None, persist_total=None):
        B = wl.bytes_per_update(args.store)
        lps = wl.launches_per_step() / hpl                            # hpl: half-steps per launch (k_persist; 1 otherwise)
        slots_per_launch = (wl.N // world) / lps                      # per GPU
        avg_launch_s = gpu_ms * 1e-3 / (K * lps)                       # timed-region events / launches
        achieved = slots_per_launch * B / avg_launch_s / 1e9
        line = {
            "metric": "walker-updates/sec (whole node), 64-dim correlated Gaussian, StretchMove a=2",
            "value": wl.N * K / wall_s, "unit": "walker-updates/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall_s * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.label + ", rng=%s, store=%s" % (args.rng, args.store),
                       "nwalkers": wl.N, "ndim": wl.D, "parallelism": "walker-sharded x%d%s" % (world, how)},
            "timing": "median of K-step blocks, each bracketed by barrier + synchronize, >= %.0f ms measured" % MIN_TIMED_MS
                      if not args.single_block else "one K-step block",
            "steps_per_s": K / wall_s, "accept_frac": accept, "device_status": status,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic_per_launch(hpl),
                         "traffic_source": traffic_source,
                         "kernel": ("emx::k_persist<8,2,4,DPB=4> (persistent: %.1f half-steps per launch, a device-wide barrier between them; "
                                    "G=8 lanes/walker, V=2, CH=4; f64 MFMA dense target)" % hpl) if hpl > 1.0 else
                                   "emx::k_halfstep<8,2,4,STRETCH,DPB=4,LEAN> (G=8 lanes/walker, V=2, CH=4; f64 MFMA dense target)",
                         "halfsteps_per_launch": hpl, "avg_halfstep_us": avg_launch_s * 1e6 / hpl,
                         "per_launch_event_halfsteps": event_hpl,
                         "algorithmic_bytes_per_walker_update": B, "walker_updates_per_launch": slots_per_launch,
                         "avg_launch_us": avg_launch_s * 1e6, "per_launch_event_us": per_launch_us,
                         "note": "avg_launch_us = hipEvent time of the timed region / launches of the kernel: it includes the "
                                 "inter-kernel gaps and the batched plan kernel (k_native_plan_batch, 1 launch per 16 steps)"
                                 + (" and, on sharded runs, the exchange" if sharded else "") +
                                 "; per_launch_event_us brackets single launches (of per_launch_event_halfsteps half-steps when "
                                 "persistent) with hipEvents"},
        }
        roofline_audit(line["roofline"], wl, args.store, accept, slots_per_launch / avg_launch_s, line["roofline"]["traffic"], avg_launch_s)
        line.update(extra)
        if persist_total:
            line["persist"] = persist_total
        if args.all_on_device is not None:
            line["test_mode"] = ("--all-on-device %d: every rank shares ONE GPU -- a control-flow / protocol test of the N > 1 path, NOT a "
                                 "multi-GPU measurement" % args.all_on_device)
        return line

    def emit(line):
        out = _claim_stdout()
        out.write(json.dumps(line) + "\n")
        out.flush()

    if not sharded:
        wl = Workload("c2", 65536)
        res = measure_single(wl, K, W, device=local_rank, rng=args.rng, store=args.store, single_block=args.single_block)
        extra = {"timed_blocks": res["blocks"], "best_block_ms_per_step": res["wall_min_s"] * 1e3 / K}
        line = headline(wl, res["wall_s"], res["gpu_ms"], res["per_launch_us"], res["accept_frac"], res["status"], "", extra,
                        hpl=res.get("halfsteps_per_launch", 1.0), event_hpl=res.get("per_launch_halfsteps"),
                        persist_total=res.get("persist_total"))
        if not args.no_extras:
            cfgs = {}
            plan = [("c3", 262144, False), ("c4", 65536, False), ("c5", 16384, False), ("c2", 65536, True),
                    # beyond the BASELINE list: two ensembles no cache can hold (the HBM roofline taken literally) and the
                    # MFMA-bound wide dense targets
                    ("hbm_dense", 1048576, False), ("hbm_wide", 262144, False), ("w512", 65536, False), ("w128", 65536, False)]
            for key, n, st in plan:
                if args.config not in ("all", key):
                    continue
                name = {"c3": "c3_262144x32_rosen", "c4": "c4_de_snooker", "c5": "c5_16384x1024_diag", "c2": "c2_store",
                        "hbm_dense": "hbm_1048576x64_dense", "hbm_wide": "hbm_262144x1024_diag", "w512": "wide_65536x512_dense",
                        "w128": "dense_65536x128_fused"}[key]
                try:
                    w2 = wl if key == "c2" else Workload(key, n)
                    Ks = K if not st else min(K, 200)          # stored chain: 33.5 MB per step
                    if key in ("hbm_dense", "hbm_wide", "w512"):
                        Ks = max(4, min(K, 20))                # 0.2 - 1.5 ms per step: a few steps fill the timed 50 ms
                    r2 = measure_single(w2, Ks, min(W, Ks), device=local_rank, rng="philox", store=st, single_block=args.single_block,
                                        spin_s=0.05 if key.startswith(("hbm", "w")) else 0.15)
                    cfgs[name] = wide_entry(w2, r2, Ks) if key == "w512" else config_entry(w2, r2, Ks, st)
                    if key == "w128":     # fused: HBM roofline as for C2; the contraction is 5.4 flop per byte here (C2: 2.7)
                        fl = float(w2.D) ** 2 + 3.0 * w2.D
                        cfgs[name]["roofline"]["mfma_f64_frac_wall_clock"] = cfgs[name]["wu_per_s"] * fl / 1e12 / MFMA_F64_PEAK_TFLOPS
                        cfgs[name]["roofline"]["kernel"] = ("emx::k_halfstep_slab<DPB=8,STRETCH> (csrc/emx_slab.hip: the tile's proposals in registers, a "
                                                            "32-column LDS slab, eight waves a CU; 144 f64 MFMAs per 16-row tile)")
                except Exception as e:  # noqa: BLE001
                    cfgs[name] = {"error": repr(e)}
                    log("config %s failed: %r" % (name, e))
            line["configs"] = cfgs
            if args.config == "all":          # (a single --config: that configuration only, e.g. under rocprofv3)
                try:
                    line["exact_mode"] = exact_mode_entry(wl, K, W, local_rank)
                except Exception as e:  # noqa: BLE001
                    line["exact_mode"] = {"error": repr(e)}
                try:
                    line["exact_mode_c3"] = exact_mode_large_entry(K, W, local_rank)
                except Exception as e:  # noqa: BLE001
                    line["exact_mode_c3"] = {"error": repr(e)}
                try:
                    line["quality"] = quality_entry(local_rank)
                except Exception as e:  # noqa: BLE001
                    line["quality"] = {"error": repr(e)}
        line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(wl)
        emit(line)
        return

    # N > 1: this process only orchestrates (CPU; a gloo group of the parents for agreement on what failed).  Every
    # (configuration, exchange) measurement is a child process per rank with a rendezvous of its own.
    import torch.distributed as dist
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    port_base = int(os.environ["MASTER_PORT"]) + 17

    keys = ["c2", "c3", "c5", "w512"] if args.config == "all" else [args.config if args.config in ("c2", "c3", "c5", "w512") else "c2"]
    if "c2" not in keys:
        keys = ["c2"] + keys                 # the headline is always C2
    multi = {}
    line = None
    skip = {}
    exchanges = (EXCHANGES + ("logprob",) if "w512" in keys else EXCHANGES) if args.exchange == "all" else (args.exchange,)
    t_start = time.perf_counter()
    _DEADLINE[0] = t_start + args.time_budget if args.time_budget > 0 else None
    budget = worst_case_seconds(args, keys, exchanges)
    log("rank %d: time budget %.0f s (the watchdogs alone would allow %.0f s)" % (rank, args.time_budget, budget["unbounded_s"]))
    pre = None
    if not args.no_preflight or args.preflight:
        t0 = time.perf_counter()
        pre = run_preflight(args, world, dist, port_base + 200, exchanges)
        pre_s = time.perf_counter() - t0
        for ex in exchanges:
            v = pre.get(ex, {})
            if not v.get("ok"):
                skip[ex] = "preflight: " + str(v.get("error") or "final ensembles differ / device status %s" % v.get("device_status"))[:200]
        if not pre.get("p2p", {}).get("ok", True):
            log("preflight: no peer access between the devices -- the direct exchange cannot work")
            skip.setdefault("direct", "preflight: hipDeviceCanAccessPeer is false for some pair of devices")
        pre = {"seconds": pre_s, "items": pre, "disabled": dict(skip), "time_budget": budget}
        log("rank %d preflight (%.1f s): %s" % (rank, pre_s, {k: v.get("ok") for k, v in pre["items"].items()}))
        if args.preflight:
            if rank == 0:
                emit_line({"preflight": pre, "n_gpus": world})
            dist.barrier()
            dist.destroy_process_group()
            return 0
    for kn, key in enumerate(keys):
        wl, best, entry = sharded_config(key, world, K, rank, dist, args, port_base + 8 * kn, skip)
        multi[{"c2": "c2_weak_65536_per_gpu", "c3": "c3_262144x32_rosen_sharded", "c5": "c5_16384x1024_strong",
               "w512": "wide_65536x512_dense_weak"}[key]
              if args.scaling == "auto" else "%s_%s" % (key, entry["scaling"])] = entry
        if key == "c2":
            if best is None:
                if rank == 0:
                    emit({"metric": "walker-updates/sec (whole node), 64-dim correlated Gaussian, StretchMove a=2", "value": None,
                          "unit": "walker-updates/s", "n_gpus": world, "steps": K, "warmup": W, "error": entry["exchange"]})
                dist.barrier()
                dist.destroy_process_group()
                return
            how = ", %s via %s" % ({"pull": "all-to-all of the partner rows (pull exchange)",
                                    "allgather": "all-gather of the updated rows",
                                    "direct": "partner rows read in place from the peers' HBM (direct exchange)",
                                    "replay": "all-gather of the decisions, accepted updates recomputed on every replica (replay exchange)",
                                    "replay_push": "decisions stored into the peers' buffers, accepted updates recomputed on every replica "
                                                   "(device-side replay exchange)"}.get(
                                        best["exchange"], best["exchange"]), best["comm"])
            line = headline(wl, best["wall_s"], best["gpu_ms"], None, best["accept_frac"], best["status"], how,
                            {"timed_blocks": best["blocks"], "rccl_ranks": best.get("rccl_ranks"),
                             "distinct_devices": best.get("distinct_devices")})
            line["scaling"] = entry["scaling"]
            line["cpu_baseline"] = None
            line["multi_gpu"] = multi
    if rank == 0 and line is not None:
        line["multi_gpu"] = multi
        budget["used_s"] = time.perf_counter() - t_start
        line["time_budget"] = budget
        if pre is not None:
            line["preflight"] = pre
        emit(line)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
