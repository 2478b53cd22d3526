"""k_small_run: whole runs of a small ensemble inside one workgroup (state in LDS, barriers instead of kernel
boundaries) must give exactly the bits of the general launch-per-half-step path."""
import time

import numpy as np
import pytest

from emcee_amd import _lib
from oracle import cases
from oracle import sampler_oracle as so

from test_gpu_parity import make_ens

pytestmark = pytest.mark.gpu


def build(N, D, target, nsplits, a, seed, kind="stretch"):
    mv = so.MoveSpec(kind, nsplits=nsplits, a=a, live_dangerously=True, sigma=0.05, gammas=1.4)
    cases.DIGEST_CASES["_sm"] = dict(N=N, D=D, target=target, moves=[mv], nsteps=1, seed=seed,
                                     p0={"rosenbrock": "rosen", "box": "uniform"}.get(target, "randn"))
    spec = cases.build("_sm")
    del cases.DIGEST_CASES["_sm"]
    return spec


def run(spec, small, nsteps, thin_by, store, seed=4242, step0=0, chunks=1):
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(seed, step0)
    ens.set_tuning("small_kernel", small)
    if store:
        ens.chain_config(nsteps * chunks)
    for _ in range(chunks):
        ens.run(nsteps, thin_by, store)
    assert ens.status() == 0
    out = dict(state=ens.get_state(), acc=ens.accepted_mask(), step=ens.get_philox()[1], it=ens.iteration())
    if store:
        out.update(chain=ens.chain_read(0, 0, nsteps * chunks), lp=ens.chain_read(1, 0, nsteps * chunks), cnt=ens.accepted_counts())
    ens.close()
    return out


SHAPES = [(32, 5, "iso", 2, 2.0), (50, 3, "iso", 2, 2.0), (45, 2, "iso", 3, 2.0), (64, 8, "rosenbrock", 2, 2.0),
          (66, 7, "diag", 2, 3.0), (32, 1, "box", 2, 2.0), (128, 16, "diag", 2, 2.0), (100, 33, "iso", 4, 2.0),
          (256, 32, "rosenbrock", 2, 2.0), (40, 130, "diag", 5, 2.0), (1024, 8, "iso", 2, 2.0), (4096, 2, "iso", 2, 1.5),
          (16, 256, "iso", 2, 2.0), (2, 1, "iso", 2, 2.0)]


@pytest.mark.parametrize("N,D,target,nsplits,a", SHAPES)
def test_small_run_equals_general_path(N, D, target, nsplits, a):
    spec = build(N, D, target, nsplits, a, seed=N + D)
    for nsteps, thin_by, store in ((7, 1, True), (4, 3, True), (9, 2, False)):
        fast = run(spec, 1, nsteps, thin_by, store)
        slow = run(spec, 0, nsteps, thin_by, store)
        assert fast["step"] == slow["step"] == nsteps * thin_by and fast["it"] == slow["it"]
        assert np.array_equal(fast["state"][0], slow["state"][0]) and np.array_equal(fast["state"][1], slow["state"][1])
        assert np.array_equal(fast["acc"], slow["acc"])
        if store:
            assert np.array_equal(fast["chain"], slow["chain"]) and np.array_equal(fast["lp"], slow["lp"])
            assert np.array_equal(fast["cnt"], slow["cnt"])
            assert fast["cnt"].sum() > 0


@pytest.mark.parametrize("kind,N,D,target,nsplits", [
    ("de", 64, 4, "iso", 2), ("de", 45, 7, "diag", 3), ("de", 256, 16, "rosenbrock", 2), ("de", 30, 33, "iso", 2),
    ("snooker", 64, 4, "iso", 4), ("snooker", 38, 3, "diag", 4), ("snooker", 200, 16, "rosenbrock", 4),
    ("snooker", 40, 65, "iso", 4),
])
def test_small_run_de_and_snooker_equal_general_path(kind, N, D, target, nsplits):
    spec = build(N, D, target, nsplits, 2.0, seed=N * 3 + D, kind=kind)
    for nsteps, thin_by, store in ((6, 1, True), (3, 2, True)):
        fast = run(spec, 1, nsteps, thin_by, store)
        slow = run(spec, 0, nsteps, thin_by, store)
        assert np.array_equal(fast["chain"], slow["chain"]) and np.array_equal(fast["lp"], slow["lp"])
        assert np.array_equal(fast["cnt"], slow["cnt"]) and np.array_equal(fast["acc"], slow["acc"])
        assert np.array_equal(fast["state"][0], slow["state"][0]) and fast["step"] == slow["step"]
        assert fast["cnt"].sum() > 0


def run_exact(spec, small, nsteps, thin_by, store, chunks=1):
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(np.random.RandomState(77).get_state())
    ens.set_tuning("small_kernel", small)
    if store:
        ens.chain_config(nsteps * chunks)
    for _ in range(chunks):
        ens.run(nsteps, thin_by, store)
    assert ens.status() == 0
    st = ens.get_mt19937()
    out = dict(state=ens.get_state(), acc=ens.accepted_mask(), mt=(st[1].copy(), st[2], st[3], st[4]))
    if store:
        out.update(chain=ens.chain_read(0, 0, nsteps * chunks), lp=ens.chain_read(1, 0, nsteps * chunks), cnt=ens.accepted_counts())
    ens.close()
    return out


@pytest.mark.parametrize("kind,N,D,target,nsplits", [
    ("stretch", 32, 5, "iso", 2), ("stretch", 50, 3, "iso", 2), ("stretch", 45, 2, "iso", 3), ("stretch", 66, 7, "diag", 2),
    ("stretch", 256, 32, "rosenbrock", 2), ("stretch", 1024, 8, "iso", 2), ("de", 64, 4, "iso", 2), ("de", 30, 3, "diag", 3),
    ("snooker", 64, 4, "iso", 4), ("snooker", 38, 3, "diag", 4),
])
def test_small_run_exact_mode_equals_general_path(kind, N, D, target, nsplits):
    """MT19937 mode: plans of many steps made by the host twin in one go, consumed by the one-workgroup kernel --
    same chain, same accept counts and the same final generator state as the per-step path."""
    spec = build(N, D, target, nsplits, 2.0, seed=N + 7 * D, kind=kind)
    for nsteps, thin_by, store, chunks in ((9, 1, True, 1), (4, 3, True, 2), (300, 1, False, 1)):
        fast = run_exact(spec, 1, nsteps, thin_by, store, chunks)
        slow = run_exact(spec, 0, nsteps, thin_by, store, chunks)
        assert np.array_equal(fast["state"][0], slow["state"][0]) and np.array_equal(fast["state"][1], slow["state"][1])
        assert np.array_equal(fast["acc"], slow["acc"])
        assert np.array_equal(fast["mt"][0], slow["mt"][0]) and fast["mt"][1:] == slow["mt"][1:]
        if store:
            assert np.array_equal(fast["chain"], slow["chain"]) and np.array_equal(fast["lp"], slow["lp"])
            assert np.array_equal(fast["cnt"], slow["cnt"])


@pytest.mark.parametrize("rng", ["philox", "mt"])
@pytest.mark.parametrize("N,D,target,kinds,weights", [
    (64, 5, "iso", ["stretch", "de"], [0.5, 0.5]), (128, 8, "diag", ["de", "snooker"], [0.8, 0.2]),
    (48, 3, "rosenbrock", ["stretch", "de", "snooker"], [0.2, 0.5, 0.3]), (40, 17, "iso", ["snooker", "stretch"], [0.3, 0.7]),
])
def test_small_run_move_schedules_equal_general_path(N, D, target, kinds, weights, rng):
    """weighted mixtures: the move of each step comes from the same draw as on the general path (the host's MT19937
    choice() in exact mode, one Philox draw against the cdf in native mode)"""
    from emx_testlib import cdf_of
    mvs = [so.MoveSpec(k, nsplits=2, live_dangerously=True, sigma=0.05, gammas=1.4) for k in kinds]
    cases.DIGEST_CASES["_sm"] = dict(N=N, D=D, target=target, moves=mvs, weights=weights, nsteps=1, seed=N + D,
                                     p0="rosen" if target == "rosenbrock" else "randn")
    spec = cases.build("_sm")
    del cases.DIGEST_CASES["_sm"]
    outs = []
    for small in (1, 0):
        ens = make_ens(spec, spec["p0"])
        if rng == "mt":
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(np.random.RandomState(5).get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(2718, 0)
        ens.set_tuning("small_kernel", small)
        ens.chain_config(40)
        ens.run(25, 1, True)
        ens.run(5, 3, True)
        assert ens.status() == 0
        outs.append((ens.chain_read(0, 0, 30), ens.chain_read(1, 0, 30), ens.accepted_counts(), ens.get_state()[0],
                     ens.get_mt19937()[1] if rng == "mt" else np.array(ens.get_philox())))
        ens.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert outs[0][2].sum() > 0


@pytest.mark.parametrize("rng", ["philox", "mt"])
@pytest.mark.parametrize("N,D,kinds", [
    (32, 5, ["stretch"]), (16, 64, ["stretch"]), (64, 16, ["stretch"]), (50, 17, ["stretch"]), (60, 33, ["stretch"]),
    (5, 100, ["stretch"]), (64, 8, ["de"]), (128, 8, ["de", "snooker"]), (24, 48, ["stretch", "de"]), (4, 112, ["stretch"]),
    (128, 64, ["stretch"]),       # too much contraction for one CU: both runs take the general path
])
def test_small_run_dense_target_equals_general_path(N, D, kinds, rng):
    """dense Gaussian target inside the one-workgroup kernel: the tile / MFMA / reduction sequence is k_halfstep's, so the
    log-probs, and with them every decision, are the same bits"""
    mvs = [so.MoveSpec(k, live_dangerously=True, sigma=0.05, gammas=1.4) for k in kinds]
    cases.DIGEST_CASES["_sm"] = dict(N=N, D=D, target="dense", moves=mvs, weights=None, nsteps=1, seed=N + D)
    spec = cases.build("_sm")
    del cases.DIGEST_CASES["_sm"]
    outs = []
    for small in (1, 0):
        ens = make_ens(spec, spec["p0"])
        if rng == "mt":
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(np.random.RandomState(5).get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(31, 0)
        ens.set_tuning("small_kernel", small)
        ens.chain_config(20)
        ens.run(14, 1, True)
        ens.run(2, 3, True)
        assert ens.status() == 0
        outs.append((ens.chain_read(0, 0, 16), ens.chain_read(1, 0, 16), ens.accepted_counts(), ens.get_state()[0], ens.get_state()[1]))
        ens.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert outs[0][2].sum() > 0


def test_small_run_chunks_and_resume():
    """more steps than one launch takes (4096), and a second emx_run call continuing the chain"""
    spec = build(32, 5, "iso", 2, 2.0, seed=3)
    fast = run(spec, 1, 5000, 1, False)
    slow = run(spec, 0, 5000, 1, False)
    assert np.array_equal(fast["state"][0], slow["state"][0]) and fast["step"] == 5000
    fast = run(spec, 1, 6, 2, True, chunks=3)
    slow = run(spec, 0, 6, 2, True, chunks=3)
    assert np.array_equal(fast["chain"], slow["chain"]) and np.array_equal(fast["cnt"], slow["cnt"])


def test_small_run_is_faster_and_not_used_when_it_cannot_be():
    spec = build(32, 5, "iso", 2, 2.0, seed=9)
    t = {}
    for small in (1, 0):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(1, 0)
        ens.set_tuning("small_kernel", small)
        ens.run(2000, 1, False)
        ens.sync()
        t0 = time.perf_counter()
        ens.run(20000, 1, False)
        ens.sync()
        t[small] = (time.perf_counter() - t0) / 20000
        ens.close()
    assert t[1] < 0.5 * t[0], t
    # too big for one CU's LDS: the general path must take over silently and still be right
    big = build(8192, 4, "iso", 2, 2.0, seed=5)
    a, b = run(big, 1, 3, 1, True), run(big, 0, 3, 1, True)
    assert np.array_equal(a["chain"], b["chain"])


def _random_small_configs():
    rs = np.random.RandomState(4321)
    out = []
    for _ in range(40):
        kind = ["stretch", "de", "snooker"][rs.randint(3)]
        nsplits = 4 if kind == "snooker" else int(rs.randint(2, 6))
        D = int(rs.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 33, 40]))
        N = int(rs.randint(max(2 * nsplits, 4), 70))
        target = ["iso", "diag", "rosenbrock", "dense"][rs.randint(4)]
        if target == "rosenbrock" and D < 2:
            target = "iso"
        out.append((kind, N, D, target, nsplits, ["mt", "philox"][rs.randint(2)], int(rs.randint(1, 4))))
    return out


@pytest.mark.parametrize("kind,N,D,target,nsplits,rng,thin_by", _random_small_configs())
def test_small_run_random_configs_equal_general_path(kind, N, D, target, nsplits, rng, thin_by):
    spec = build(N, D, target, nsplits, 2.0, seed=N * 41 + D, kind=kind)
    outs = []
    for small in (1, 0):
        ens = make_ens(spec, spec["p0"])
        if rng == "mt":
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(np.random.RandomState(N).get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(N * 7 + D, 2)
        ens.set_tuning("small_kernel", small)
        ens.chain_config(12)
        ens.run(7, thin_by, True)
        ens.run(5, 1, True)
        assert ens.status() == 0
        outs.append((ens.chain_read(0, 0, 12), ens.chain_read(1, 0, 12), ens.accepted_counts(), ens.accepted_mask()))
        ens.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("N,D,target,mode,factor,diag,mix", [
    (32, 5, "iso", "vector", None, False, False), (40, 3, "iso", "random", 2.0, False, False),
    (24, 3, "iso", "sequential", None, False, False), (48, 6, "diag", "vector", 1.5, True, False),
    (32, 5, "dense", "vector", None, False, False), (64, 8, "rosenbrock", "random", None, True, False),
    (32, 3, "iso", "vector", None, False, True), (30, 4, "dense", "sequential", 1.3, True, True),
])
def test_small_run_gaussian_move_equals_general_path(N, D, target, mode, factor, diag, mix):
    """Gaussian Metropolis move (native mode) inside the one-workgroup kernel: same normals (generated in registers
    from the same counters), same factor and sequential cursor as the general path; alone and mixed with the stretch move"""
    rs = np.random.RandomState(D)
    cov = (0.05 + 0.1 * rs.rand(D)) if diag else 0.08
    mvs = [so.MoveSpec("gaussian", cov=cov, mode=mode, factor=factor)]
    weights = None
    if mix:
        mvs = [so.MoveSpec("stretch", live_dangerously=True)] + mvs
        weights = [0.6, 0.4]
    cases.DIGEST_CASES["_sm"] = dict(N=N, D=D, target=target, moves=mvs, weights=weights, nsteps=1, seed=N + D,
                                     p0="rosen" if target == "rosenbrock" else "randn")
    spec = cases.build("_sm")
    del cases.DIGEST_CASES["_sm"]
    outs = []
    for small in (1, 0):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(1618, 0)
        ens.set_tuning("small_kernel", small)
        ens.chain_config(30)
        ens.run(17, 1, True)
        ens.run(4, 2, True)
        assert ens.status() == 0
        cursor = int(ens.get_move(len(mvs) - 1).gammas)
        outs.append((ens.chain_read(0, 0, 21), ens.chain_read(1, 0, 21), ens.accepted_counts(), ens.get_state()[0], np.array([cursor])))
        ens.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert 0 < outs[0][2].sum() < 21 * N
