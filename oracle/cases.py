"""Parity cases shared by oracle/gen_golden.py and the tests (TEST INFRASTRUCTURE).

A case is fully determined by its name: ``build(name)`` returns the synthetic
inputs (p0, target description, moves, weights, nsteps, thin_by, seed).  The
target description is a plain dict so that the same case can be fed to the
reference (as a NumPy callable), to the oracle, and to the device library
(as a target descriptor).
"""
import numpy as np

from . import sampler_oracle as so


def _dense_params(D, seed):
    """SURVEY.md 8d, C2 construction: Sigma = A A^T / D + 0.1 I."""
    rs = np.random.RandomState(seed)
    mu = rs.randn(D)
    A = rs.randn(D, D)
    cov = A @ A.T / D + 0.1 * np.eye(D)
    icov = np.linalg.inv(cov)
    icov = 0.5 * (icov + icov.T)
    return mu, cov, icov


def make_target(desc):
    """desc -> vectorised numpy callable (n, D) -> (n,)."""
    k = desc["kind"]
    if k == "iso":
        return so.iso_gauss
    if k == "diag":
        return lambda x: so.diag_gauss(x, desc["mu"], desc["ivar"])
    if k == "dense":
        return lambda x: so.dense_gauss(x, desc["mu"], desc["icov"])
    if k == "rosenbrock":
        return so.rosenbrock
    if k == "box":
        def box(x):
            x = np.atleast_2d(x)
            bad = np.any((x > 1) | (x < 0), axis=1)
            return np.where(bad, -np.inf, 0.0)
        return box
    raise ValueError(k)


# name -> (N, D, target kind, moves, weights, nsteps, thin_by, seed, p0 kind)
_S = so.MoveSpec
CASES = {
    # BASELINE config 1 (plumbing case), reference run per-walker (vectorize=False)
    "c1_stretch_32x5_iso": dict(N=32, D=5, target="iso", moves=[_S("stretch")], nsteps=50, seed=1234, per_walker=True),
    # non power-of-two halves: randint / shuffle rejection paths
    "stretch_50x3_iso": dict(N=50, D=3, target="iso", moves=[_S("stretch")], nsteps=30, seed=7),
    "stretch_a3_66x7_diag": dict(N=66, D=7, target="diag", moves=[_S("stretch", a=3.0)], nsteps=20, seed=11),
    "stretch_256x16_dense": dict(N=256, D=16, target="dense", moves=[_S("stretch")], nsteps=10, seed=21),
    "stretch_128x64_dense": dict(N=128, D=64, target="dense", moves=[_S("stretch")], nsteps=6, seed=22),
    "stretch_128x8_rosen": dict(N=128, D=8, target="rosenbrock", moves=[_S("stretch")], nsteps=10, seed=31, p0="rosen"),
    "stretch_nsplits3_45x2": dict(N=45, D=2, target="iso", moves=[_S("stretch", nsplits=3)], nsteps=15, seed=41),
    "stretch_nsplits5_fixed_40x2": dict(N=40, D=2, target="iso", moves=[_S("stretch", nsplits=5, randomize_split=False)], nsteps=15, seed=42),
    "stretch_box_32x1": dict(N=32, D=1, target="box", moves=[_S("stretch")], nsteps=40, seed=51, p0="uniform"),
    "stretch_thin3_32x2": dict(N=32, D=2, target="iso", moves=[_S("stretch")], nsteps=8, thin_by=3, seed=61),
    "stretch_wide_16x130_live": dict(N=16, D=130, target="diag", moves=[_S("stretch", live_dangerously=True)], nsteps=6, seed=71),
    # dense precision matrices wider than LDS (padded ndim > 112): the propose -> MFMA log-prob -> commit path of emx_wide.hip
    "stretch_48x130_dense": dict(N=48, D=130, target="dense", moves=[_S("stretch", live_dangerously=True)], nsteps=5, seed=81),
    "mix_de_snooker_40x113_dense": dict(N=40, D=113, target="dense", nsteps=8, seed=82, weights=[0.7, 0.3],
                                        moves=[_S("de", live_dangerously=True), _S("snooker", live_dangerously=True)]),
    "de_64x4_iso": dict(N=64, D=4, target="iso", moves=[_S("de")], nsteps=20, seed=101),
    "de_g1_s01_30x3": dict(N=30, D=3, target="iso", moves=[_S("de", gamma0=1.0, sigma=0.1)], nsteps=20, seed=102),
    "snooker_64x4_iso": dict(N=64, D=4, target="iso", moves=[_S("snooker")], nsteps=20, seed=201),
    "snooker_38x3_diag": dict(N=38, D=3, target="diag", moves=[_S("snooker", gammas=1.2)], nsteps=15, seed=202),
    # BASELINE config 4 shape at oracle-feasible size
    "mix_de_snooker_128x8_dense": dict(N=128, D=8, target="dense", moves=[_S("de"), _S("snooker")], weights=[0.8, 0.2], nsteps=25, seed=301),
    "mix_stretch_de_64x5": dict(N=64, D=5, target="iso", moves=[_S("stretch"), _S("de")], nsteps=20, seed=302),
}

# SURVEY.md 8(f) widening: moves whose proposal is host code in the reference and in the product
# (MHMove/GaussianMove: whole ensemble at once; WalkMove/KDEMove: split-ensemble with a custom get_proposal).
HOST_MOVE_CASES = {
    "gauss_iso_vector_40x3": dict(N=40, D=3, target="iso", moves=[_S("gaussian", cov=0.25)], nsteps=20, seed=501),
    "gauss_diag_random_factor_30x4": dict(N=30, D=4, target="diag", nsteps=20, seed=502,
                                          moves=[_S("gaussian", cov=[0.1, 0.2, 0.3, 0.4], mode="random", factor=2.0)]),
    "gauss_iso_sequential_24x3": dict(N=24, D=3, target="iso", moves=[_S("gaussian", cov=0.5, mode="sequential")], nsteps=14, seed=503),
    "gauss_full_32x3_dense": dict(N=32, D=3, target="dense", nsteps=15, seed=504,
                                  moves=[_S("gaussian", cov=[[0.3, 0.1, 0.0], [0.1, 0.2, 0.05], [0.0, 0.05, 0.4]])]),
    # unit/test_sampler.py:26-28 mixes Stretch + Gaussian
    "mix_stretch_gauss_32x3": dict(N=32, D=3, target="iso", moves=[_S("stretch"), _S("gaussian", cov=0.3)], weights=[0.6, 0.4], nsteps=25, seed=505),
    "walk_24x2_iso": dict(N=24, D=2, target="iso", moves=[_S("walk")], nsteps=8, seed=511),
    "walk_s5_30x3_diag": dict(N=30, D=3, target="diag", moves=[_S("walk", s=5)], nsteps=8, seed=512),
    "kde_40x2_iso": dict(N=40, D=2, target="iso", moves=[_S("kde")], nsteps=8, seed=521),
}

# Larger cases: only a digest of the reference output is committed.  Their start state must be the SAME BITS on every CPU, or the
# reference digest cannot be asserted there (round-5 verdict: the dense cases built p0 through a BLAS matmul with the Cholesky
# factor, the digest was applied "if p0 matches" and silently was not on the GPU box).  So the dense cases here start from
# p0 = mu + randn (p0="mu_randn": element-wise arithmetic only, no BLAS / LAPACK in the start state); the equilibrium start
# (through the Cholesky factor) is covered by the fixtures above, whose p0 is stored.
DIGEST_CASES = {
    "stretch_4096x64_dense": dict(N=4096, D=64, target="dense", p0="mu_randn", moves=[_S("stretch")], nsteps=3, seed=401),
    "stretch_2048x32_rosen": dict(N=2048, D=32, target="rosenbrock", moves=[_S("stretch")], nsteps=3, seed=402, p0="rosen"),
    "stretch_2048x1024_diag": dict(N=2048, D=1024, target="diag", moves=[_S("stretch")], nsteps=2, seed=403),
    "stretch_1024x256_dense": dict(N=1024, D=256, target="dense", p0="mu_randn", moves=[_S("stretch")], nsteps=2, seed=405),
    "stretch_1100x520_dense": dict(N=1100, D=520, target="dense", p0="mu_randn", moves=[_S("stretch")], nsteps=2, seed=406),
    "mix_de_snooker_1024x64_dense": dict(N=1024, D=64, target="dense", p0="mu_randn", moves=[_S("de"), _S("snooker")], weights=[0.8, 0.2], nsteps=6, seed=404),
    # round 4: long enough for several launches of the persistent kernels in exact mode (sixteen steps each: k_plan_fetch)
    "stretch_1024x16_iso_40": dict(N=1024, D=16, target="iso", moves=[_S("stretch")], nsteps=40, seed=411),
    "de_2048x8_diag_24": dict(N=2048, D=8, target="diag", moves=[_S("de")], nsteps=24, seed=412),
    "snooker_1024x8_iso_20": dict(N=1024, D=8, target="iso", moves=[_S("snooker")], nsteps=20, seed=413),
    "stretch_512x64_dense_35": dict(N=512, D=64, target="dense", p0="mu_randn", moves=[_S("stretch")], nsteps=35, seed=414),
}


def build(name):
    spec = dict(CASES.get(name) or HOST_MOVE_CASES.get(name) or DIGEST_CASES[name])
    N, D, seed = spec["N"], spec["D"], spec["seed"]
    kind = spec["target"]
    desc = {"kind": kind}
    if kind == "diag":
        rs = np.random.RandomState(seed + 1000)
        desc["mu"] = rs.randn(D)
        desc["ivar"] = 1.0 / (0.1 + rs.rand(D))
    elif kind == "dense":
        mu, cov, icov = _dense_params(D, seed + 1000)
        desc.update(mu=mu, cov=cov, icov=icov)
    rs = np.random.RandomState(seed)
    p0kind = spec.get("p0", "randn")
    if p0kind == "uniform":
        p0 = rs.rand(N, D)
    elif p0kind == "rosen":
        p0 = 1.0 + 0.1 * rs.randn(N, D)
    elif p0kind == "mu_randn":
        p0 = desc["mu"] + rs.randn(N, D)
    elif kind == "dense":
        p0 = desc["mu"] + rs.randn(N, D) @ np.linalg.cholesky(desc["cov"]).T
    elif kind == "diag":
        p0 = desc["mu"] + rs.randn(N, D) / np.sqrt(desc["ivar"])
    else:
        p0 = rs.randn(N, D)
    spec.update(p0=p0, desc=desc, weights=spec.get("weights"),
                thin_by=spec.get("thin_by", 1), per_walker=spec.get("per_walker", False),
                rng_seed=seed + 5000)
    return spec
