"""The drop-in Python surface on the GPU: EnsembleSampler / State / Move plugin protocol.

Modelled on the reference's own tests (src/emcee/tests/unit/test_sampler.py, test_state.py,
test_stretch.py, integration/test_proposal.py) plus bit-exact comparisons with the
reference-generated fixtures."""
import pickle

import numpy as np
import pytest

import emcee_amd
from emcee_amd import moves, targets
from emcee_amd.model import Model
from emcee_amd.state import State
from oracle import cases
from oracle import sampler_oracle as so

from helpers import load_golden, rng_from_fixture

pytestmark = pytest.mark.gpu


def target_of(desc):
    k = desc["kind"]
    return {"iso": lambda: targets.IsoGaussian(),
            "diag": lambda: targets.DiagGaussian(desc["mu"], desc["ivar"]),
            "dense": lambda: targets.DenseGaussian(desc["mu"], desc["icov"]),
            "rosenbrock": lambda: targets.Rosenbrock(20.0),
            "box": lambda: targets.UniformBox()}[k]()


def move_of(m):
    kw = dict(nsplits=m.nsplits, randomize_split=m.randomize_split, live_dangerously=m.live_dangerously)
    if m.kind == "gaussian":
        return moves.GaussianMove(m.cov, mode=m.mode, factor=m.factor)
    if m.kind == "walk":
        return moves.WalkMove(s=m.s, **kw)
    if m.kind == "kde":
        return moves.KDEMove(bw_method=m.bw_method, **kw)
    if m.kind == "stretch":
        return moves.StretchMove(a=m.a, **kw)
    if m.kind == "de":
        return moves.DEMove(sigma=m.sigma, gamma0=m.gamma0, **kw)
    kw.pop("nsplits")
    return moves.DESnookerMove(gammas=m.gammas, **kw)


def make_sampler(spec, g, log_prob=None, **kw):
    mv = [move_of(m) for m in spec["moves"]]
    if spec["weights"] is not None:
        mv = list(zip(mv, spec["weights"]))
    s = emcee_amd.EnsembleSampler(spec["N"], spec["D"], log_prob if log_prob is not None else target_of(spec["desc"]),
                                  moves=mv, **kw)
    s._random.set_state(rng_from_fixture(g).get_state())
    return s


EXACT = ["c1_stretch_32x5_iso", "stretch_50x3_iso", "stretch_256x16_dense", "stretch_128x8_rosen",
         "stretch_nsplits5_fixed_40x2", "stretch_thin3_32x2", "de_64x4_iso", "mix_stretch_de_64x5",
         "stretch_box_32x1", "stretch_wide_16x130_live"]


@pytest.mark.parametrize("name", EXACT)
def test_run_mcmc_same_seed_same_chain_as_reference(name):
    g = load_golden(name)
    spec = cases.build(name)
    s = make_sampler(spec, g)
    last = s.run_mcmc(g["p0"], spec["nsteps"], thin_by=spec["thin_by"], skip_initial_state_check=True)
    assert np.array_equal(s.get_chain(), g["chain"])
    np.testing.assert_allclose(s.get_log_prob(), g["log_prob"], rtol=1e-11)
    assert np.array_equal(s.backend.accepted, g["accepted_count"])
    np.testing.assert_array_equal(s.acceptance_fraction, g["accepted_count"] / spec["nsteps"])
    st = s.random_state
    assert np.array_equal(st[1], g["rng_key1"]) and st[2] == int(g["rng_pos1"])
    assert np.array_equal(last.coords, g["chain"][-1])
    assert s.iteration == spec["nsteps"]


@pytest.mark.parametrize("name", list(cases.HOST_MOVE_CASES))
def test_host_proposal_moves_same_chain_as_reference(name):
    """MHMove/GaussianMove (log-probs from the device evaluator), WalkMove/KDEMove (host get_proposal,
    device accept/commit) and the Stretch+Gaussian mixture of the reference's own unit tests."""
    g = load_golden(name)
    spec = cases.build(name)
    s = make_sampler(spec, g)
    s.run_mcmc(g["p0"], spec["nsteps"], skip_initial_state_check=True)
    assert np.array_equal(s.get_chain(), g["chain"])
    np.testing.assert_allclose(s.get_log_prob(), g["log_prob"], rtol=1e-11)
    assert np.array_equal(s.backend.accepted, g["accepted_count"])
    st = s.random_state
    assert np.array_equal(st[1], g["rng_key1"]) and st[2] == int(g["rng_pos1"])


@pytest.mark.parametrize("name", ["c1_stretch_32x5_iso", "mix_stretch_de_64x5", "snooker_64x4_iso"])
def test_sample_generator_and_host_callable_paths(name):
    """(a) sample() generator with a device target, (b) an ordinary per-walker Python log_prob_fn
    (split-phase path), (c) vectorize=True -- all reproduce the reference chain."""
    g = load_golden(name)
    spec = cases.build(name)
    tol = dict(rtol=1e-9, atol=1e-10) if "snooker" in name else None

    def check(chain):
        if tol:
            np.testing.assert_allclose(chain, g["chain"], **tol)
        else:
            assert np.array_equal(chain, g["chain"])

    s = make_sampler(spec, g)
    n = 0
    for st in s.sample(g["p0"], iterations=spec["nsteps"], skip_initial_state_check=True):
        n += 1
        assert isinstance(st, State) and st.coords.shape == (spec["N"], spec["D"])
    assert n == spec["nsteps"]
    check(s.get_chain())

    s = make_sampler(spec, g, log_prob=lambda p: -0.5 * np.sum(p ** 2))
    s.run_mcmc(g["p0"], spec["nsteps"], skip_initial_state_check=True)
    check(s.get_chain())
    if not tol:
        assert np.array_equal(s.get_log_prob(), g["log_prob"])      # same NumPy callable as the reference run

    s = make_sampler(spec, g, log_prob=so.iso_gauss, vectorize=True)
    s.run_mcmc(g["p0"], spec["nsteps"], skip_initial_state_check=True)
    check(s.get_chain())


def _mk(nwalkers=32, ndim=3, **kw):
    np.random.seed(1)
    return emcee_amd.EnsembleSampler(nwalkers, ndim, targets.IsoGaussian(), **kw), np.random.randn(nwalkers, ndim)


@pytest.mark.parametrize("mv", [None, "list", "weighted"])
def test_shapes(mv):
    """reference unit/test_sampler.py:20-84"""
    m = {None: None, "list": [moves.StretchMove(), moves.DEMove()],
         "weighted": [(moves.DEMove(), 0.8), (moves.DESnookerMove(), 0.2)]}[mv]
    s, p0 = _mk(moves=m)
    N = 30
    s.run_mcmc(p0, N, skip_initial_state_check=True)
    assert s.get_chain().shape == (N, 32, 3)
    assert s.get_log_prob().shape == (N, 32)
    assert s.acceptance_fraction.shape == (32,)
    assert s.get_chain(flat=True).shape == (N * 32, 3)
    assert s.get_log_prob(flat=True).shape == (N * 32,)
    assert s.get_chain(thin=3, discard=4).shape == (len(range(4 + 2, N, 3)), 32, 3)
    assert np.array_equal(s.get_chain(thin=3, discard=4), s.get_chain()[6::3])
    assert s.get_blobs() is None
    assert np.all(s.acceptance_fraction > 0)
    s.reset()
    assert s.iteration == 0
    with pytest.raises(AttributeError):
        s.get_chain()


def test_errors():
    """reference unit/test_sampler.py:87-124, ensemble.py error paths"""
    s, p0 = _mk()
    with pytest.raises(ValueError):
        s.run_mcmc(p0[:, :2], 5)
    with pytest.raises(ValueError):
        s.run_mcmc(p0, 5, thin_by=0)
    with pytest.raises(ValueError):
        s.run_mcmc(None, 5)
    with pytest.raises(ValueError):
        next(s.sample(p0, iterations=None, store=True))
    with pytest.raises(ValueError):          # ill-conditioned start
        s.run_mcmc(np.ones((32, 3)), 5)
    with pytest.raises(ValueError):
        s.run_mcmc(p0, 5, thin=0) if False else next(s.sample(p0, iterations=4, thin=0))
    bad = emcee_amd.EnsembleSampler(32, 3, lambda p: np.nan)
    with pytest.raises(ValueError):
        bad.run_mcmc(p0, 2)
    few, q0 = _mk(nwalkers=4, ndim=3)
    with pytest.raises(RuntimeError):
        few.run_mcmc(q0, 2, skip_initial_state_check=True)
    ok, q0 = _mk(nwalkers=4, ndim=3, moves=moves.StretchMove(live_dangerously=True))
    ok.run_mcmc(q0, 2, skip_initial_state_check=True)


def test_thin_by_equals_thinned_full_run_and_restart():
    """reference unit/test_sampler.py:152-209: same seed => thin_by run == strided full run; restart."""
    np.random.seed(7)
    p0 = np.random.randn(32, 3)
    a = emcee_amd.EnsembleSampler(32, 3, targets.IsoGaussian())
    a._random.seed(5)
    a.run_mcmc(p0, 24)
    b = emcee_amd.EnsembleSampler(32, 3, targets.IsoGaussian())
    b._random.seed(5)
    b.run_mcmc(p0, 8, thin_by=3)
    assert np.array_equal(b.get_chain(), a.get_chain()[2::3])
    assert np.array_equal(b.get_log_prob(), a.get_log_prob()[2::3])
    # restart continues the same stream
    c = emcee_amd.EnsembleSampler(32, 3, targets.IsoGaussian())
    c._random.seed(5)
    c.run_mcmc(p0, 10)
    c.run_mcmc(None, 14)
    assert np.array_equal(c.get_chain(), a.get_chain())
    # generator path == fast path
    d = emcee_amd.EnsembleSampler(32, 3, targets.IsoGaussian())
    d._random.seed(5)
    for _ in d.sample(p0, iterations=24):
        pass
    assert np.array_equal(d.get_chain(), a.get_chain())


def test_input_state_not_overwritten_and_pickle():
    """reference unit/test_state.py:35-47, unit/test_sampler.py:225-234"""
    s, p0 = _mk()
    keep = p0.copy()
    st = State(p0)
    s.run_mcmc(st, 10)
    assert np.array_equal(p0, keep) and np.array_equal(st.coords, keep)
    s2 = pickle.loads(pickle.dumps(s))
    assert s2.nwalkers == 32 and s2.pool is None
    x = State(p0, log_prob=np.zeros(32), random_state=s.random_state)
    c, lp, rs = x
    assert c is x.coords and len(x) == 3 and x[-1] is rs


@pytest.mark.parametrize("rng", ["mt19937", "philox"])
def test_state_handed_back_unread_continues_on_the_device(rng):
    """run_mcmc returns a State and takes it back (reference ensemble.py:312, 441-447).  Here the returned object is lazy
    (ResidentState): handed back unread it costs no upload, yet it keeps the values it was returned with whatever runs later,
    and every way of continuing gives the chain of the plain-array continuation."""
    from emcee_amd.state import ResidentState
    mk = lambda: emcee_amd.EnsembleSampler(64, 4, targets.IsoGaussian(), rng=rng)  # noqa: E731
    p0 = np.random.RandomState(11).randn(64, 4)

    def seeded():
        s = mk()
        s._random.seed(5)
        return s

    # reference continuation: arrays go down and up again between the calls
    a = seeded()
    st = a.run_mcmc(p0, 7)
    mid = (st.coords.copy(), st.log_prob.copy())
    st2 = a.run_mcmc(State(st.coords.copy(), log_prob=st.log_prob.copy(), random_state=st.random_state), 9)
    ref_chain, ref_end = a.get_chain(), st2.coords.copy()

    # handed back unread: same chain, and the FIRST returned object still shows the state after 7 steps
    b = seeded()
    r1 = b.run_mcmc(p0, 7)
    assert isinstance(r1, ResidentState) and r1._c is None
    r2 = b.run_mcmc(r1, 9)
    assert r1._c is None and r1._slot is not None          # kept by a device-side snapshot, never crossed PCIe
    assert np.array_equal(b.get_chain(), ref_chain) and np.array_equal(r2.coords, ref_end)
    assert np.array_equal(r1.coords, mid[0]) and np.array_equal(r1.log_prob, mid[1]) and r1._slot is None

    # initial_state=None -> the previous state; and an OLD unread state restarts from its snapshot (no upload either)
    c = seeded()
    q1 = c.run_mcmc(p0, 7)
    c.run_mcmc(None, 9)
    assert np.array_equal(c.get_chain(), ref_chain)
    rs_mid = q1.random_state
    c.reset()
    c.random_state = rs_mid
    if rng == "philox":
        c._philox_step = 7              # the counter of the native stream is sampler state, not part of random_state
    q3 = c.run_mcmc(q1, 9)                                  # q1 is two generations old: restored inside HBM
    assert np.array_equal(c.get_chain(), ref_chain[7:]) and np.array_equal(q3.coords, ref_end)

    # read and edited by the caller: an ordinary State, uploaded like one
    d = seeded()
    e1 = d.run_mcmc(p0, 7)
    e1.coords[:] = mid[0][::-1]
    e1.log_prob[:] = mid[1][::-1]
    e2 = d.run_mcmc(e1, 3, store=False)
    f = mk()
    f.random_state = e1.random_state
    if rng == "philox":
        f._philox_step = 7
    g2 = f.run_mcmc(State(mid[0][::-1].copy(), log_prob=mid[1][::-1].copy(), random_state=e1.random_state), 3, store=False)
    assert np.array_equal(e2.coords, g2.coords)
    # pickling materialises
    h = pickle.loads(pickle.dumps(r2))
    assert type(h) is State and np.array_equal(h.coords, ref_end)
    # a context that goes away brings home what still lives on it: the current state and the ones in snapshot slots
    k = seeded()
    s1 = k.run_mcmc(p0, 7)
    s2 = k.run_mcmc(s1, 9)
    assert s1._slot is not None and s2._c is None
    k._ens.close()
    assert np.array_equal(s1.coords, mid[0]) and np.array_equal(s2.coords, ref_end)


def test_infinite_iterations_without_store():
    s, p0 = _mk()
    for i, st in enumerate(s.sample(p0, iterations=None, store=False)):
        if i > 6:
            break
    assert s.iteration == 0


def test_blobs_and_named_parameters_on_the_host_callable_path():
    np.random.seed(3)
    p0 = np.random.randn(32, 2)

    def lp_blob(p):
        return -0.5 * np.sum(p ** 2), float(p[0]), float(p[1] * 2)

    s = emcee_amd.EnsembleSampler(32, 2, lp_blob)
    s.run_mcmc(p0, 12)
    blobs = s.get_blobs()
    chain = s.get_chain()
    assert blobs.shape == (12, 32, 2)
    np.testing.assert_allclose(blobs[..., 0], chain[..., 0])
    np.testing.assert_allclose(blobs[..., 1], 2 * chain[..., 1])

    def lp_named(params):
        return -0.5 * (params["x"] ** 2 + params["y"] ** 2)

    s = emcee_amd.EnsembleSampler(32, 2, lp_named, parameter_names=["x", "y"])
    s.run_mcmc(p0, 5)
    assert s.get_chain().shape == (5, 32, 2)


def test_move_plugin_protocol_with_a_fake_model():
    """reference unit/test_stretch.py:15-34: propose() needs only the 4-field Model."""
    nwalkers, ndim = 5, 10
    for Mv in (moves.StretchMove, moves.DEMove):
        np.random.seed(11)
        coords = np.random.randn(nwalkers, ndim)
        st = State(coords, log_prob=np.zeros(nwalkers))
        model = Model(None, lambda x: (np.zeros(len(x)), None), map, np.random)
        with pytest.raises(RuntimeError):
            Mv().propose(model, st)
        Mv(live_dangerously=True).propose(model, st)


@pytest.mark.parametrize("kind", ["stretch", "de", "snooker"])
def test_propose_matches_oracle_and_advances_the_callers_rng(kind):
    N, D = 48, 4
    rs = np.random.RandomState(21)
    coords = rs.randn(N, D)
    lp = so.iso_gauss(coords)
    spec = so.MoveSpec(kind, sigma=0.3)
    mv = move_of(spec)
    r1, r2 = np.random.RandomState(99), np.random.RandomState(99)
    xo, lpo = coords.copy(), lp.copy()
    acc_o = so.propose(xo, lpo, so.iso_gauss, r1, spec)
    st = State(coords.copy(), log_prob=lp.copy())
    model = Model(None, lambda x: (so.iso_gauss(x), None), map, r2)
    st2, acc = mv.propose(model, st)
    assert st2 is st and np.array_equal(acc, acc_o)
    if kind == "snooker":
        np.testing.assert_allclose(st.coords, xo, rtol=1e-9, atol=1e-10)
    else:
        assert np.array_equal(st.coords, xo) and np.array_equal(st.log_prob, lpo)
    a, b = r1.get_state(), r2.get_state()
    assert np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and a[4] == b[4]


def test_get_proposal_hook_and_user_subclass():
    """A user RedBlueMove subclass with its own (NumPy) get_proposal runs through the device
    accept/commit; calling the built-in get_proposal directly gives the reference's q, factors."""
    N, D = 40, 3
    rs = np.random.RandomState(2)
    coords = rs.randn(N, D)
    s, c = coords[:20], [coords[20:]]
    r1, r2 = np.random.RandomState(5), np.random.RandomState(5)
    q_ref, f_ref = so._stretch(s, c, r1, 2.0, None)
    q, f = moves.StretchMove().get_proposal(s, c, r2)
    assert np.array_equal(q, q_ref)
    np.testing.assert_allclose(f, f_ref, rtol=1e-14)
    assert r1.get_state()[2] == r2.get_state()[2]

    class MyStretch(moves.RedBlueMove):          # user code: same maths written by hand
        def get_proposal(self, s, c, random):
            c = np.concatenate(c, axis=0)
            zz = ((2.0 - 1.0) * random.rand(len(s)) + 1) ** 2.0 / 2.0
            rint = random.randint(len(c), size=(len(s),))
            return c[rint] - (c[rint] - s) * zz[:, None], (s.shape[1] - 1.0) * np.log(zz)

    np.random.seed(4)
    p0 = np.random.randn(32, 3)
    a = emcee_amd.EnsembleSampler(32, 3, lambda p: -0.5 * np.sum(p ** 2), moves=MyStretch())
    a._random.seed(8)
    a.run_mcmc(p0, 15)
    b = emcee_amd.EnsembleSampler(32, 3, lambda p: -0.5 * np.sum(p ** 2), moves=moves.StretchMove())
    b._random.seed(8)
    b.run_mcmc(p0, 15)
    assert np.array_equal(a.get_chain(), b.get_chain())
    assert np.array_equal(a.backend.accepted, b.backend.accepted)


@pytest.mark.parametrize("mv,ndim,nsteps", [(lambda: moves.StretchMove(), 1, 2000), (lambda: moves.StretchMove(), 3, 2000),
                                            (lambda: moves.DEMove(), 2, 2000), (lambda: moves.DEMove(gamma0=1.0), 1, 2000),
                                            (lambda: moves.DESnookerMove(), 2, 4000),
                                            (lambda: moves.StretchMove(nsplits=5), 2, 2000),
                                            # reference integration/test_gaussian.py: every mode, with and without factor
                                            (lambda: moves.GaussianMove(0.5), 1, 4000),
                                            (lambda: moves.GaussianMove([0.5, 0.3], mode="random"), 2, 6000),
                                            (lambda: moves.GaussianMove(0.5, mode="sequential", factor=2.0), 2, 6000),
                                            (lambda: [(moves.StretchMove(), 0.5), (moves.GaussianMove(0.4, factor=1.5), 0.5)], 2, 3000)])
def test_normal_target_statistics_philox(mv, ndim, nsteps):
    """reference integration/test_proposal.py:31-76 (_test_normal), run in the native RNG mode."""
    from scipy import stats
    np.random.seed(1234)
    nwalkers = 32
    coords = np.random.randn(nwalkers, ndim)
    s = emcee_amd.EnsembleSampler(nwalkers, ndim, targets.IsoGaussian(), moves=mv(), rng="philox")
    s.run_mcmc(coords, nsteps)
    acc = s.acceptance_fraction
    assert np.all((acc < 0.9) * (acc > 0.1)), acc
    samps = s.get_chain(flat=True)
    mu, sig = np.mean(samps, axis=0), np.std(samps, axis=0)
    assert np.all(np.abs(mu) < 0.08), mu
    assert np.all(np.abs(sig - 1) < 0.05), sig
    if ndim == 1:
        ks, _ = stats.kstest(samps[:, 0], "norm")
        assert ks < 0.05


@pytest.mark.parametrize("mv,nsteps", [(lambda: moves.WalkMove(s=3), 1500), (lambda: moves.WalkMove(), 600),
                                       (lambda: moves.KDEMove(), 1500)])
def test_normal_target_statistics_host_proposal_moves(mv, nsteps):
    """reference integration/test_walk.py, test_kde.py (_test_normal): host get_proposal, device accept/commit."""
    np.random.seed(1234)
    nwalkers, ndim = 32, 2
    coords = np.random.randn(nwalkers, ndim)
    s = emcee_amd.EnsembleSampler(nwalkers, ndim, targets.IsoGaussian(), moves=mv())
    s.run_mcmc(coords, nsteps)
    acc = s.acceptance_fraction
    assert np.all((acc < 0.95) * (acc > 0.1)), acc
    samps = s.get_chain(flat=True, discard=100)
    mu, sig = np.mean(samps, axis=0), np.std(samps, axis=0)
    assert np.all(np.abs(mu) < 0.08), mu
    assert np.all(np.abs(sig - 1) < 0.05), sig


def test_gaussian_sequential_cursor_survives_runs():
    """The sequential mode's coordinate cursor is state of the move (gaussian.py:66,97): two runs of 5 + 9
    steps must equal the reference's single 14-step run, and the move object must show the cursor."""
    name = "gauss_iso_sequential_24x3"
    g = load_golden(name)
    spec = cases.build(name)
    s = make_sampler(spec, g)
    st = s.run_mcmc(g["p0"], 5, skip_initial_state_check=True)
    assert s._moves[0].get_proposal.index == 5 % 3
    s.run_mcmc(st, 9, skip_initial_state_check=True)
    assert s._moves[0].get_proposal.index == 14 % 3
    assert np.array_equal(s.get_chain(), g["chain"])
    assert np.array_equal(s.backend.accepted, g["accepted_count"])


@pytest.mark.parametrize("mv,rng", [(lambda: None, "philox"), (lambda: None, "mt19937"), (lambda: moves.DEMove(), "philox"),
                                    (lambda: moves.DESnookerMove(), "philox"), (lambda: moves.GaussianMove(0.5), "philox"),
                                    (lambda: moves.GaussianMove(0.5, mode="random", factor=2.0), "mt19937"),
                                    (lambda: moves.StretchMove(randomize_split=False), "philox")])
def test_uniform_start_leaves_its_initialisation(mv, rng):
    """reference integration/test_proposal.py:79-102 (_test_uniform) for every move family
    (test_stretch.py:22, test_de.py:18, test_de_snooker.py:15, test_gaussian.py:78)."""
    from scipy import stats
    np.random.seed(1234)
    coords = np.random.rand(32, 1)
    s = emcee_amd.EnsembleSampler(32, 1, targets.IsoGaussian(), rng=rng, moves=mv())
    s.run_mcmc(coords, 2000)
    acc = s.acceptance_fraction
    assert np.all((acc < 0.9) * (acc > 0.1))
    samps = s.get_chain(flat=True)
    np.random.shuffle(samps)
    ks, _ = stats.kstest(samps[::100, 0], "uniform")
    assert ks > 0.1


def test_compute_log_prob_and_autocorr_time():
    s, p0 = _mk(nwalkers=64, ndim=2)
    lp, blobs = s.compute_log_prob(p0[:10])
    np.testing.assert_allclose(lp, so.iso_gauss(p0[:10]), rtol=1e-13)
    assert blobs is None
    with pytest.raises(ValueError):
        s.compute_log_prob(np.array([[np.inf, 0.0, 0.0]]))
    np.random.seed(2)
    s = emcee_amd.EnsembleSampler(64, 2, targets.IsoGaussian(), rng="philox")
    s.run_mcmc(np.random.randn(64, 2), 3000)
    tau = s.get_autocorr_time(quiet=True)
    assert tau.shape == (2,) and np.all(tau > 1) and np.all(tau < 60)


def test_device_autocorr_equals_host_estimator():
    """tau from emx_autocorr (batched hipFFT on the HBM-resident chain, behind the C ABI) == the host
    estimator (which tests/test_autocorr_cpu.py holds equal to the reference's) == the torch.fft cross-check."""
    from emcee_amd import autocorr
    from devfft_twin import integrated_time_device, mean_acf
    np.random.seed(5)
    s = emcee_amd.EnsembleSampler(96, 3, targets.IsoGaussian(), rng="philox")
    s.run_mcmc(np.random.randn(96, 3), 1500)
    ens = s.backend._dev
    assert ens is not None
    for discard, thin, c in ((0, 1, 5), (100, 3, 5), (7, 2, 3.5), (1499, 1, 5)):
        host_chain = s.get_chain(discard=discard, thin=thin)
        tau_lib, win, n_t = ens.autocorr(discard=discard, thin=thin, c=c)
        assert n_t == host_chain.shape[0]
        if n_t < 2:
            continue
        tau_h = autocorr.integrated_time(host_chain, c=c, quiet=True)
        np.testing.assert_allclose(tau_lib, tau_h, rtol=1e-8)
        for d in range(3):      # the window itself: auto_window on the walker-averaged ACF of the host chain
            rho = autocorr._batched_acf(np.asarray(host_chain[:, :, d], dtype=float)).mean(axis=1)
            assert win[d] == autocorr.tau_from_mean_acf(rho, c)[0]
        tau_t = integrated_time_device(ens, s.iteration, discard=discard, thin=thin, c=c, quiet=True)    # torch.fft twin
        np.testing.assert_allclose(tau_lib, tau_t, rtol=1e-8)
        f = mean_acf(ens, s.iteration, discard=discard, thin=thin)
        assert f.shape == (host_chain.shape[0], 3) and abs(f[0, 0] - 1.0) < 1e-12
    np.testing.assert_allclose(s.get_autocorr_time(quiet=True), autocorr.integrated_time(s.get_chain(), quiet=True), rtol=1e-8)
    np.testing.assert_allclose(s.get_autocorr_time(quiet=True, discard=50, thin=4),
                               4 * autocorr.integrated_time(s.get_chain(discard=50, thin=4), quiet=True), rtol=1e-8)
    with pytest.raises(autocorr.AutocorrError):
        s.get_autocorr_time(tol=1e6)
    # several chunks of walkers (the chunk size follows from the padded length): a longer chain of a wider ensemble
    np.random.seed(6)
    s = emcee_amd.EnsembleSampler(4096, 8, targets.IsoGaussian(), rng="philox")
    s.run_mcmc(np.random.randn(4096, 8), 600)
    np.testing.assert_allclose(s.get_autocorr_time(quiet=True), autocorr.integrated_time(s.get_chain(), quiet=True), rtol=1e-8)


def test_reference_move_instances_are_recognised_by_duck_typing():
    """A move object that *is* reference emcee's StretchMove / DEMove (same class name, module
    ``emcee.moves...``, un-overridden get_proposal) takes the fused device path."""
    from emcee_amd.ensemble import _native_desc

    def ref_like(name, module, **attrs):
        def get_proposal(self, s, c, random):          # never called on the device path
            raise AssertionError("host get_proposal must not run for a built-in move")
        klass = type(name, (object,), {"get_proposal": get_proposal, "__module__": module})
        obj = klass()
        obj.nsplits, obj.randomize_split, obj.live_dangerously = 2, True, False
        for k, v in attrs.items():
            setattr(obj, k, v)
        return obj

    st = ref_like("StretchMove", "emcee.moves.stretch", a=2.5)
    de = ref_like("DEMove", "emcee.moves.de", sigma=1e-5, gamma0=None)
    d = _native_desc(st, 4)
    assert d is not None and d.kind == 0 and d.a == 2.5
    d = _native_desc(de, 8)
    assert d is not None and d.kind == 1 and abs(d.g0 - 2.38 / 4.0) < 1e-15
    assert _native_desc(ref_like("WalkMove", "emcee.moves.walk"), 4) is None
    assert _native_desc(ref_like("StretchMove", "mypkg.moves", a=2.0), 4) is None

    np.random.seed(3)
    p0 = np.random.randn(32, 4)
    a = emcee_amd.EnsembleSampler(32, 4, targets.IsoGaussian(), moves=st)
    a._random.seed(9)
    a.run_mcmc(p0, 12)
    b = emcee_amd.EnsembleSampler(32, 4, targets.IsoGaussian(), moves=moves.StretchMove(a=2.5))
    b._random.seed(9)
    b.run_mcmc(p0, 12)
    assert np.array_equal(a.get_chain(), b.get_chain())


def test_resume_reset_growth_and_thinned_acceptance():
    np.random.seed(8)
    p0 = np.random.randn(64, 3)
    s = emcee_amd.EnsembleSampler(64, 3, targets.IsoGaussian())
    s._random.seed(4)
    s.run_mcmc(p0, 7)
    s.run_mcmc(None, 5, thin_by=2)                 # chain grows on the device, 10 more proposals
    s.run_mcmc(None, 3, store=False)
    assert s.iteration == 12 and s.get_chain().shape == (12, 64, 3)
    full = emcee_amd.EnsembleSampler(64, 3, targets.IsoGaussian())
    full._random.seed(4)
    full.run_mcmc(p0, 17)
    c = full.get_chain()
    assert np.array_equal(s.get_chain()[:7], c[:7])
    assert np.array_equal(s.get_chain()[7:], c[8:17:2])
    # backend.accepted only counts the stored proposals (reference backend.py:229 behind ensemble.py:416)
    moved_stored = np.concatenate([np.any(np.diff(np.concatenate([p0[None], c[:7]]), axis=0) != 0, axis=2),
                                   np.any(c[8:17:2] != c[7:16:2], axis=2)]).sum(axis=0)
    assert np.array_equal(s.backend.accepted, moved_stored)
    s.reset()
    assert s.iteration == 0
    s.run_mcmc(p0, 4)
    assert s.get_chain().shape == (4, 64, 3)


def test_distributed_front_end_world1():
    """EnsembleSampler(distributed=True): RNG state, inputs and the RCCL id are replicated through
    torch.distributed, the run itself is the library-driven sharded emx_run.  One rank here (one GPU);
    the chain must equal the plain sampler's, for both exchanges."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        name = "mix_stretch_gauss_32x3"
        g = load_golden(name)
        spec = cases.build(name)
        plain = make_sampler(spec, g)
        plain.run_mcmc(g["p0"], spec["nsteps"], skip_initial_state_check=True)
        assert np.array_equal(plain.get_chain(), g["chain"])
        for how in ("run_mcmc", "sample"):
            s = make_sampler(spec, g, distributed=True)
            assert s.device == 0
            if how == "run_mcmc":
                s.run_mcmc(g["p0"], spec["nsteps"], skip_initial_state_check=True)
            else:
                for _ in s.sample(g["p0"], iterations=spec["nsteps"], skip_initial_state_check=True):
                    pass
            assert np.array_equal(s.get_chain(), g["chain"])
            assert np.array_equal(s.backend.accepted, g["accepted_count"])
        s = make_sampler(spec, g, distributed=True, exchange="pull", rng="philox")
        with pytest.raises(RuntimeError, match="partial"):
            s.run_mcmc(g["p0"], 3, skip_initial_state_check=True)
        last = s.run_mcmc(g["p0"], 40, skip_initial_state_check=True, store=False)
        ref = make_sampler(spec, g, rng="philox")
        ref_last = ref.run_mcmc(g["p0"], 40, skip_initial_state_check=True, store=False)
        assert np.array_equal(last.coords, ref_last.coords) and np.array_equal(last.log_prob, ref_last.log_prob)
        s = make_sampler(spec, g, distributed=True, exchange="direct", rng="philox")     # in-place partner reads: block ownership too
        with pytest.raises(RuntimeError, match="partial"):
            s.run_mcmc(g["p0"], 3, skip_initial_state_check=True)
        last = s.run_mcmc(g["p0"], 40, skip_initial_state_check=True, store=False)
        assert np.array_equal(last.coords, ref_last.coords) and np.array_equal(last.log_prob, ref_last.log_prob)
        s = make_sampler(spec, g, distributed=True, exchange="logprob")                  # replicas stay whole: stored chains are fine
        s.run_mcmc(g["p0"], spec["nsteps"], skip_initial_state_check=True)
        assert np.array_equal(s.get_chain(), g["chain"])
        with pytest.raises(RuntimeError, match="DeviceTarget"):
            bad = emcee_amd.EnsembleSampler(32, 3, lambda x: -0.5 * np.sum(x * x), distributed=True)
            bad.run_mcmc(g["p0"], 2)
        # a Python callable: its calls are shared out over the ranks (all of them to the one rank here)
        fn = cases.make_target(spec["desc"])
        np.random.seed(3)
        a = emcee_amd.EnsembleSampler(32, 3, fn, vectorize=True, distributed=True, exchange="logprob")
        a.run_mcmc(g["p0"], 10)
        np.random.seed(3)
        b = emcee_amd.EnsembleSampler(32, 3, fn, vectorize=True)
        b.run_mcmc(g["p0"], 10)
        assert np.array_equal(a.get_chain(), b.get_chain()) and np.array_equal(a.get_log_prob(), b.get_log_prob())
    finally:
        if created:
            dist.destroy_process_group()


def test_generator_rng_state_is_current_whenever_somebody_looks():
    """The MT19937 state stays in libemx between yields and is copied out on demand: sampler.random_state, the
    yielded state's random_state and the backend's must all be what reference emcee would hold at that point."""
    name = "stretch_50x3_iso"
    g = load_golden(name)
    spec = cases.build(name)
    k = 7
    ref = make_sampler(spec, g)
    ref.run_mcmc(g["p0"], k, skip_initial_state_check=True)          # one native call: the state after k steps
    want = ref.random_state
    s = make_sampler(spec, g)
    for n, st in enumerate(s.sample(g["p0"], iterations=spec["nsteps"], skip_initial_state_check=True), 1):
        if n == k:
            for got in (st.random_state, s.random_state, s.backend.random_state, s.get_last_sample().random_state):
                assert np.array_equal(got[1], want[1]) and got[2:] == want[2:]
            assert np.array_equal(st.coords, g["chain"][k - 1])
    end = s.random_state
    assert np.array_equal(end[1], g["rng_key1"]) and end[2] == int(g["rng_pos1"])
    assert np.array_equal(s.backend.random_state[1], g["rng_key1"])
    assert np.array_equal(s.get_chain(), g["chain"])
    # leaving the loop early pins the backend's state at the last stored step
    s = make_sampler(spec, g)
    for n, st in enumerate(s.sample(g["p0"], iterations=spec["nsteps"], skip_initial_state_check=True), 1):
        if n == k:
            break
    s.run_mcmc(s.get_last_sample(), 3, store=False, skip_initial_state_check=True)   # unsaved steps must not leak into it
    got = s.backend.random_state
    assert np.array_equal(got[1], want[1]) and got[2:] == want[2:]
    # setting the state by hand wins over whatever the device holds
    s.random_state = want
    assert np.array_equal(s.random_state[1], want[1])


def test_consecutive_unstored_runs_from_plain_arrays_continue_the_stream():
    """Two runs started from bare ndarrays (whose State carries random_state=None) with nobody reading the generator
    state in between: the second run must continue the MT19937 stream where the first left it in libemx, exactly
    as the reference's silently-failing setter leaves its RandomState alone (ensemble.py:228-238).  (Round-1 advisor
    finding: the setter dropped the device-held state before the failing set_state.)"""
    name = "stretch_50x3_iso"
    g = load_golden(name)
    spec = cases.build(name)
    k = 4
    # the oracle, one generator across both runs
    rs = rng_from_fixture(g)
    fn = cases.make_target(spec["desc"])
    a = so.run(g["p0"], k, fn, rs, moves=spec["moves"], weights=spec["weights"], store=False)
    b = so.run(a["coords"], spec["nsteps"] - k, fn, rs, moves=spec["moves"], weights=spec["weights"], store=False,
               log_prob0=a["lp"])
    assert np.array_equal(b["coords"], g["chain"][-1])          # the oracle itself reproduces the reference run
    for runner in ("sample", "run_mcmc"):
        s = make_sampler(spec, g)
        if runner == "sample":
            for st in s.sample(g["p0"], iterations=k, store=False, skip_initial_state_check=True):
                pass
            mid = np.array(st.coords)
            for st in s.sample(mid, iterations=spec["nsteps"] - k, store=False, skip_initial_state_check=True):
                pass
            end = np.array(st.coords)
        else:
            mid = s.run_mcmc(g["p0"], k, store=False, skip_initial_state_check=True).coords
            end = s.run_mcmc(np.array(mid), spec["nsteps"] - k, store=False, skip_initial_state_check=True).coords
        assert np.array_equal(mid, a["coords"]), runner
        assert np.array_equal(end, g["chain"][-1]), runner
        fin = s.random_state
        assert np.array_equal(fin[1], g["rng_key1"]) and fin[2] == int(g["rng_pos1"]), runner
    # an invalid state is ignored without losing the stream position either
    s = make_sampler(spec, g)
    s.run_mcmc(g["p0"], k, store=False, skip_initial_state_check=True)
    s.random_state = "not a state"
    end = s.run_mcmc(None, spec["nsteps"] - k, store=False, skip_initial_state_check=True).coords
    assert np.array_equal(end, g["chain"][-1])


def test_generator_path_of_a_large_ensemble_runs_on_the_persistent_plan_pipeline():
    """sample() takes one native step per yield.  For ensembles of >= 8192 walkers in MT19937 mode those single steps are
    served by the plan pipeline that keeps running ahead between the calls; reading the generator state in the middle
    retires and restarts it.  Chain and final state must equal the one-call run_mcmc (itself pinned to the oracle at full
    size by test_gpu_full_size.py)."""
    N, D, nst = 8192, 8, 14
    rs = np.random.RandomState(21)
    p0 = rs.randn(N, D)
    mv = [(moves.StretchMove(), 0.6), (moves.DEMove(), 0.4)]
    a = emcee_amd.EnsembleSampler(N, D, targets.IsoGaussian(), moves=mv)
    a._random.seed(77)
    a.run_mcmc(p0, nst, skip_initial_state_check=True)
    b = emcee_amd.EnsembleSampler(N, D, targets.IsoGaussian(), moves=mv)
    b._random.seed(77)
    mid_state = None
    for n, st in enumerate(b.sample(p0, iterations=nst, skip_initial_state_check=True), 1):
        if n == 5:
            mid_state = st.random_state            # copies the MT19937 state out of libemx: the pipeline is retired here
    assert np.array_equal(a.get_chain(), b.get_chain())
    assert np.array_equal(a.get_log_prob(), b.get_log_prob())
    fa, fb = a.random_state, b.random_state
    assert np.array_equal(fa[1], fb[1]) and fa[2:] == fb[2:]
    c = emcee_amd.EnsembleSampler(N, D, targets.IsoGaussian(), moves=mv)
    c._random.seed(77)
    c.run_mcmc(p0, 5, skip_initial_state_check=True)
    want = c.random_state
    assert np.array_equal(mid_state[1], want[1]) and mid_state[2:] == want[2:]


def test_wrong_length_log_prob_vector_is_a_value_error():
    """A vectorised log_prob_fn that returns the wrong number of values must raise, not be read past its end."""
    s, p0 = _mk(32, 3)
    bad = emcee_amd.EnsembleSampler(32, 3, lambda x: -0.5 * np.sum(x ** 2, axis=1)[:-1] if len(x) < 32 else
                                    -0.5 * np.sum(x ** 2, axis=1), vectorize=True)
    with pytest.raises(ValueError):
        bad.run_mcmc(p0, 2, skip_initial_state_check=True)


def test_normal_stretch_with_blobs_on_the_host_callable_path():
    """reference integration/test_stretch.py:17-19 with blobs=True (test_proposal.py:21-23: the blob is a Python
    object per walker)"""
    np.random.seed(1234)
    coords = np.random.randn(32, 2)
    s = emcee_amd.EnsembleSampler(32, 2, lambda x: (-0.5 * np.sum(x ** 2), "blob"))
    s.run_mcmc(coords, 1500)
    acc = s.acceptance_fraction
    assert np.all((acc < 0.9) * (acc > 0.1))
    samps = s.get_chain(flat=True)
    assert np.all(np.abs(np.mean(samps, axis=0)) < 0.08) and np.all(np.abs(np.std(samps, axis=0) - 1) < 0.05)
    assert s.get_blobs().shape == (1500, 32) and s.get_blobs()[3, 4] == "blob"


def test_reference_quickstart_tutorial_outputs():
    """docs/tutorials/quickstart.ipynb:76-327 of the reference prints mean acceptance 0.552 and mean autocorrelation time
    57.112 for this script; with the MT19937 twin the same seed gives the same chain here."""
    np.random.seed(42)
    ndim = 5
    means = np.random.rand(ndim)
    cov = 0.5 - np.random.rand(ndim ** 2).reshape((ndim, ndim))
    cov = np.triu(cov)
    cov += cov.T - np.diag(cov.diagonal())
    cov = np.dot(cov, cov)
    p0 = np.random.rand(32, ndim)
    sampler = emcee_amd.EnsembleSampler(32, ndim, targets.DenseGaussian(means, np.linalg.inv(cov)))
    state = sampler.run_mcmc(p0, 100)
    sampler.reset()
    sampler.run_mcmc(state, 10000)
    assert abs(np.mean(sampler.acceptance_fraction) - 0.552) < 1e-3
    assert abs(np.mean(sampler.get_autocorr_time()) - 57.112) < 0.5
