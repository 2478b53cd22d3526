"""bench.py: the workloads and the roofline / xGMI byte models (SURVEY.md section 8d formulas; DESIGN.md sections 4-6)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X spec (MI355X_MICROARCH.md); ~6300 achievable
HBM_ACHIEVABLE_GBPS = 6300.0   # what a streaming kernel sustains on this part (same guide): the second yardstick of the beyond-MALL configs
MFMA_F64_PEAK_TFLOPS = 78.6    # dense f64 matrix peak (v_mfma_f64_16x16x4_f64: 256 flop x 4 SIMD x 256 CU x 2.4 GHz / 8 passes)
MIN_TIMED_MS = 1000.0       # of timed K-step blocks per measurement (round-5 verdict: 50 ms was 1.8 % of the command's time)
MAX_BLOCKS = 4000


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
