"""GPU parity tests: the HIP path, called through the C ABI, against the oracle and the
reference-generated golden fixtures.  Run with `-m gpu` on an MI355X box.

Tolerances (BASELINE.json north_star): accept masks / indices bit-exact; float64
coordinates and log-probs within 1e-6 relative.  The tests below are tighter where the
arithmetic allows it: stretch and DE proposals involve no reduction, so coordinates are
required to be BIT-IDENTICAL to the reference; snooker coordinates (norms and dots are
reductions, summed in a different order) and all log-probs are held to 1e-11 relative.
"""
import numpy as np
import pytest

from emcee_amd import _lib
from oracle import cases
from oracle import sampler_oracle as so

from emx_testlib import cdf_of, move_desc, philox_plan
from helpers import digest, load_digests, load_golden, rng_for_case, rng_from_fixture, run_oracle

pytestmark = pytest.mark.gpu

LP_RTOL = 1e-11
# snooker: u = delta/|delta| and two dot products are reductions (different summation order
# than BLAS), amplified for near-zero coordinates; the contract is 1e-6 relative.
SNOOKER_RTOL, SNOOKER_ATOL = 1e-9, 1e-10


def _dev():
    from emcee_amd.device import DeviceEnsemble
    return DeviceEnsemble


def set_target(ens, desc):
    k = desc["kind"]
    if k == "iso":
        ens.set_target(_lib.TARGET_ISO)
    elif k == "diag":
        ens.set_target(_lib.TARGET_DIAG, desc["mu"], desc["ivar"])
    elif k == "dense":
        ens.set_target(_lib.TARGET_DENSE, desc["mu"], desc["icov"])
    elif k == "rosenbrock":
        ens.set_target(_lib.TARGET_ROSENBROCK, scale=20.0)
    elif k == "box":
        ens.set_target(_lib.TARGET_BOX)
    else:
        raise ValueError(k)


def make_ens(spec, p0):
    ens = _dev()(spec["N"], spec["D"])
    set_target(ens, spec["desc"])
    descs = [move_desc(m, spec["D"]) for m in spec["moves"]]
    ens.set_moves(descs, cdf_of(spec["weights"], len(descs)))
    for i, m in enumerate(spec["moves"]):
        if m.kind == "gaussian" and np.ndim(m.cov) == 1:
            ens.set_move_scale(i, np.sqrt(np.asarray(m.cov, dtype=np.float64)))
    ens.set_state(p0)
    ens.eval_state_log_prob()
    return ens


def assert_lp_close(a, b, rtol=LP_RTOL):
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    assert np.array_equal(a[~fin], b[~fin])
    np.testing.assert_allclose(a[fin], b[fin], rtol=rtol, atol=1e-13)


def has_snooker(spec):
    return any(m.kind == "snooker" for m in spec["moves"])


@pytest.mark.parametrize("name", list(cases.CASES))
def test_exact_mode_reproduces_reference_fixture(name):
    """MT19937 mode, free-running from the fixture's RNG state: same chain as reference emcee."""
    g = load_golden(name)
    spec = cases.build(name)
    ens = make_ens(spec, g["p0"])
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(rng_from_fixture(g).get_state())
    ens.chain_config(spec["nsteps"])
    ens.run(spec["nsteps"], thin_by=spec["thin_by"], store=True)
    assert ens.status() == 0
    chain = ens.chain_read(0, 0, spec["nsteps"])
    lps = ens.chain_read(1, 0, spec["nsteps"])
    acc = ens.accepted_counts()
    assert np.array_equal(acc, g["accepted_count"]), "accept decisions differ from the reference"
    if has_snooker(spec):
        np.testing.assert_allclose(chain, g["chain"], rtol=SNOOKER_RTOL, atol=SNOOKER_ATOL)
    else:
        assert np.array_equal(chain, g["chain"]), "coordinates are not bit-identical to the reference"
    assert_lp_close(lps, g["log_prob"], SNOOKER_RTOL if has_snooker(spec) else LP_RTOL)
    st = ens.get_mt19937()
    assert np.array_equal(st[1], g["rng_key1"]) and st[2] == int(g["rng_pos1"])
    assert st[3] == int(g["rng_has_gauss1"]) and st[4] == float(g["rng_cached1"])
    x, lp = ens.get_state()
    assert np.array_equal(x, chain[-1]) and np.array_equal(lp, lps[-1])
    ens.close()


@pytest.mark.parametrize("name", list(cases.DIGEST_CASES))
def test_exact_mode_mid_size_against_oracle(name):
    spec = cases.build(name)
    out = run_oracle(spec, spec["p0"], rng_for_case(spec))
    d = load_digests()[name]
    # always asserted: the digest cases' start state is BLAS-free (oracle/cases.py), the same bits on the GPU box as in the build container
    assert digest(spec["p0"]) == d["p0"], "the start state of a digest case must not depend on the CPU"
    assert digest(out["chain"]) == d["chain"], "oracle no longer matches the reference digest"
    assert digest(out["accepted_count"]) == d["accepted_count"]
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(rng_for_case(spec).get_state())
    ens.chain_config(spec["nsteps"])
    ens.run(spec["nsteps"], thin_by=spec["thin_by"], store=True)
    assert ens.status() == 0
    info = ens.persist_info()
    if info["qualifies"]:          # one move, 512 ... 8 192 walkers: the one-XCD persistent kernels on the host pipeline's plans
        assert info["launches"] >= (spec["nsteps"] + 15) // 16 and info["recovered"] == 0
    if name.endswith(("_40", "_24", "_20", "_35")):
        assert info["qualifies"], "the long mid-size cases are there for the persistent exact path"
    chain = ens.chain_read(0, 0, spec["nsteps"])
    assert np.array_equal(ens.accepted_counts(), out["accepted_count"])
    if has_snooker(spec):
        np.testing.assert_allclose(chain, out["chain"], rtol=SNOOKER_RTOL, atol=SNOOKER_ATOL)
    else:
        assert np.array_equal(chain, out["chain"])
    assert_lp_close(ens.chain_read(1, 0, spec["nsteps"]), out["log_prob"],
                    SNOOKER_RTOL if has_snooker(spec) else LP_RTOL)
    ens.close()


NATIVE_CASES = ["c1_stretch_32x5_iso", "stretch_50x3_iso", "stretch_128x64_dense", "stretch_128x8_rosen",
                "stretch_nsplits3_45x2", "stretch_wide_16x130_live", "de_64x4_iso", "de_g1_s01_30x3",
                "snooker_64x4_iso", "snooker_38x3_diag", "mix_de_snooker_128x8_dense", "stretch_box_32x1",
                "stretch_48x130_dense", "mix_de_snooker_40x113_dense"]


@pytest.mark.parametrize("name", NATIVE_CASES)
def test_native_mode_step_by_step_against_oracle(name):
    """Philox mode: the device's own plan of every step (read back with emx_plan_get) is replayed
    through the oracle on the device's pre-step state; accept masks and new state must agree."""
    spec = cases.build(name)
    fn = cases.make_target(spec["desc"])
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_PHILOX)
    seed = 0xC0FFEE + spec["seed"]
    ens.set_philox(seed, 0)
    cdf = cdf_of(spec["weights"], len(spec["moves"]))
    nacc = 0
    for step in range(8):
        x0, lp0 = ens.get_state()
        k, S = ens.step_begin(store=False)
        assert k == ens.lib.emx_host_move_choice_philox(seed, step, cdf, len(cdf))
        mv = spec["moves"][k]
        plan = ens.plan_get(S)
        host = philox_plan(seed, step, spec["N"], move_desc(mv, spec["D"]))
        for key in ("off", "order", "p0", "p1", "p2"):
            assert np.array_equal(plan[key], host[key]), (step, key)
        np.testing.assert_allclose(plan["s0"], host["s0"], rtol=1e-14)
        assert np.array_equal(plan["uacc"], host["uacc"])
        for s in range(S):
            ens.halfstep(s)
        ens.step_end()
        assert ens.status() == 0
        x1, lp1 = ens.get_state()
        acc_dev = ens.accepted_mask()
        xo, lpo = x0.copy(), lp0.copy()
        acc_or = so.propose_planned(xo, lpo, fn, plan, mv)
        assert np.array_equal(acc_dev, acc_or), "accept mask differs at step %d" % step
        if mv.kind == "snooker":
            np.testing.assert_allclose(x1, xo, rtol=SNOOKER_RTOL, atol=SNOOKER_ATOL)
        else:
            assert np.array_equal(x1, xo), "coordinates differ at step %d" % step
        assert_lp_close(lp1, lpo, SNOOKER_RTOL if mv.kind == "snooker" else LP_RTOL)
        nacc += acc_dev.sum()
    assert nacc > 0
    ens.close()


@pytest.mark.parametrize("name", ["c1_stretch_32x5_iso", "de_64x4_iso", "snooker_64x4_iso", "stretch_128x64_dense"])
def test_host_target_split_phase_equals_fused(name):
    """emx_propose / emx_accept with the log-prob evaluated by NumPy on the host gives the same
    chain as the fused device target (the arbitrary-Python-callable path of the drop-in)."""
    g = load_golden(name)
    spec = cases.build(name)
    fn = cases.make_target(spec["desc"])
    ens = make_ens(spec, g["p0"])
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(rng_from_fixture(g).get_state())
    nst = min(10, spec["nsteps"])
    ens.chain_config(nst)
    for it in range(nst):
        k, S = ens.step_begin(store=True)
        for s in range(S):
            q = ens.propose(s)
            ens.accept(s, fn(q))
        ens.step_end()
    chain = ens.chain_read(0, 0, nst)
    if has_snooker(spec):
        np.testing.assert_allclose(chain, g["chain"][:nst], rtol=SNOOKER_RTOL, atol=SNOOKER_ATOL)
    else:
        assert np.array_equal(chain, g["chain"][:nst])
    ens.close()


@pytest.mark.parametrize("name", ["c1_stretch_32x5_iso", "stretch_128x64_dense", "de_64x4_iso", "mix_de_snooker_128x8_dense"])
@pytest.mark.parametrize("flow", ["device_target_split_phase", "run_then_host_target"])
def test_philox_split_phase_completes_lean_plans(name, flow):
    """Lean native plans (made for the fused kernel: no uacc column) must be completed before a non-fused consumer reads them
    (round-3 advisor finding).  Flow 1: a device target is set, Philox plans, emx_propose / emx_accept.  Flow 2: emx_run leaves
    plans prepared ahead, then the target becomes the host: the next steps' accept decisions must not come from stale columns.
    Either way every step must equal the oracle applied to the plan the host twin computes for (seed, step)."""
    spec = cases.build(name)
    fn = cases.make_target(spec["desc"])
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_PHILOX)
    seed = 0xFACADE + spec["seed"]
    ens.set_philox(seed, 0)
    cdf = cdf_of(spec["weights"], len(spec["moves"]))
    step0 = 0
    if flow == "run_then_host_target":
        ens.run(3, 1, False)             # prepares NATIVE_BATCH_MAX lean plans, takes three
        ens.set_target(_lib.TARGET_HOST, None, None)
        step0 = 3
    for step in range(step0, step0 + 4):
        x0, lp0 = ens.get_state()
        k, S = ens.step_begin(store=False)
        assert k == ens.lib.emx_host_move_choice_philox(seed, step, cdf, len(cdf))
        mv = spec["moves"][k]
        plan = philox_plan(seed, step, spec["N"], move_desc(mv, spec["D"]))
        for s in range(S):
            q = ens.propose(s)
            ens.accept(s, fn(q))
        ens.step_end()
        assert ens.status() == 0
        x1, lp1 = ens.get_state()
        xo, lpo = x0.copy(), lp0.copy()
        acc_or = so.propose_planned(xo, lpo, fn, plan, mv)
        assert np.array_equal(ens.accepted_mask(), acc_or), "accept mask differs at step %d" % step
        if mv.kind == "snooker":
            np.testing.assert_allclose(x1, xo, rtol=SNOOKER_RTOL, atol=SNOOKER_ATOL)
        else:
            assert np.array_equal(x1, xo), "coordinates differ at step %d" % step
    ens.close()


@pytest.mark.parametrize("kind,D", [("iso", 1), ("iso", 5), ("iso", 64), ("iso", 129), ("diag", 7), ("diag", 1024),
                                    ("diag", 2048), ("rosenbrock", 2), ("rosenbrock", 32), ("rosenbrock", 33),
                                    ("rosenbrock", 300), ("dense", 3), ("dense", 16), ("dense", 17), ("dense", 64),
                                    ("dense", 100), ("dense", 112), ("dense", 113), ("dense", 128), ("dense", 129),
                                    ("dense", 200), ("dense", 257), ("dense", 512), ("dense", 1000), ("dense", 2048), ("box", 3)])
def test_batched_log_prob_eval(kind, D):
    rs = np.random.RandomState(D)
    N = 200
    desc = {"kind": kind}
    if kind == "diag":
        desc.update(mu=rs.randn(D), ivar=1.0 / (0.1 + rs.rand(D)))
    if kind == "dense":
        mu, cov, icov = cases._dense_params(D, 5)
        desc.update(mu=mu, icov=icov)
    x = rs.randn(N, D) * (0.6 if kind == "box" else 1.0) + (0.5 if kind == "box" else 0.0)
    ens = _dev()(N, D)
    set_target(ens, desc)
    got = ens.eval_log_prob(x[:77])
    assert_lp_close(got, cases.make_target(desc)(x[:77]))
    ens.set_state(x)
    ens.eval_state_log_prob()
    assert_lp_close(ens.get_state()[1], cases.make_target(desc)(x))
    ens.close()


def test_dense_ndim_limit_is_loud():
    ens = _dev()(8, 2049)
    with pytest.raises(_lib.EmxError):
        ens.set_target(_lib.TARGET_DENSE, np.zeros(2049), np.eye(2049))
    ens.close()


def test_nan_log_prob_is_reported():
    """ensemble.py:550-551: NaN log-prob -> ValueError (sticky status bit here)."""
    N, D = 32, 2
    ens = _dev()(N, D)
    ens.set_target(_lib.TARGET_DIAG, np.zeros(D), np.array([1.0, np.nan]))
    ens.set_state(np.random.RandomState(0).randn(N, D))
    ens.eval_state_log_prob()
    assert ens.status() & 1
    with pytest.raises(ValueError):
        ens.eval_state_log_prob()
        ens.raise_on_status()
    ens.close()


@pytest.mark.parametrize("name", ["stretch_128x64_dense", "stretch_128x8_rosen", "de_64x4_iso", "snooker_64x4_iso",
                                  "stretch_nsplits3_45x2"])
@pytest.mark.parametrize("store", [True, False])
def test_graph_replay_equals_plain_launches(name, store):
    """emx_run replays the native 8-step block as a hipGraph; the chain must be bit-identical to
    the same run issued as plain launches (tuning graph=0), and the graph must really be used."""
    spec = cases.build(name)
    spec["weights"] = None
    spec["moves"] = spec["moves"][:1]
    outs = []
    for graph in (0, 1):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(31337, 0)
        ens.set_tuning("graph", graph)
        ens.set_tuning("small_kernel", 0)     # these fixtures are small enough for k_small_run, which would take over
        ens.chain_config(64)
        ens.run(3, 1, store)          # plain
        ens.run(37, 1, store)         # 1 plain + 4 graph blocks + 4 plain
        ens.run(20, 1, store)         # counters re-synchronised across calls
        assert ens.status() == 0
        disabled, captured = ens.graph_state()
        if graph:
            assert not disabled and captured == (2 if store else 1), (disabled, captured)
        else:
            assert captured == 0
        x, lp = ens.get_state()
        nst = ens.iteration()[0]
        assert nst == (60 if store else 0) and ens.iteration()[1] == 60
        outs.append((x, lp, ens.accepted_mask(), ens.chain_read(0, 0, nst), ens.chain_read(1, 0, nst), ens.accepted_counts(),
                     ens.get_philox()))
        ens.close()
    for a, b in zip(*outs):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_dense_target_keeps_precision_next_to_a_large_mean():
    """(q - mu) is formed before the contraction, so walkers that sit 1e-4 away from a mean of 2.45e6 (Julian dates)
    keep full relative precision in their log-prob -- the reason the faster `Q L - mu^T L` form was not adopted
    (profiles/r01/ab_variants.txt)"""
    from emcee_amd.device import DeviceEnsemble
    rs = np.random.RandomState(8)
    N, D = 64, 6
    mu = 2.45e6 + rs.rand(D)
    A = rs.randn(D, D)
    cov = (A @ A.T / D + 0.1 * np.eye(D)) * 1e-8
    icov = np.linalg.inv(cov)
    icov = 0.5 * (icov + icov.T)
    x = mu + rs.randn(N, D) @ np.linalg.cholesky(cov).T
    ens = DeviceEnsemble(N, D)
    ens.set_target(_lib.TARGET_DENSE, mu, icov)
    got = ens.eval_log_prob(x)
    ens.close()
    r = (x.astype(np.longdouble) - mu.astype(np.longdouble))
    want = np.asarray(-0.5 * np.einsum("ij,jk,ik->i", r, icov.astype(np.longdouble), r), dtype=np.float64)
    np.testing.assert_allclose(got, want, rtol=1e-9)
    # the folded form would be off by ~|mu| / spread * eps ~ 1e-6 here
    L = np.linalg.cholesky(icov)
    folded = -0.5 * np.sum((x @ L - mu @ L) ** 2, axis=1)
    assert np.max(np.abs(folded - want) / np.abs(want)) > 100 * np.max(np.abs(got - want) / np.abs(want))


@pytest.mark.parametrize("name", ["stretch_256x16_dense", "mix_de_snooker_128x8_dense", "stretch_nsplits3_45x2"])
def test_exact_mode_pipeline_inline_and_single_steps_interleave(name):
    """MT19937 mode: emx_run takes its plans from the threaded host pipeline (csrc/emx_mtpipe.cpp), single steps
    (emx_step_begin) and `mt_pipeline = 0` make them inline.  Any interleaving must continue ONE stream: the chain, the
    accept counts and the final generator state equal the reference fixture's."""
    g = load_golden(name)
    spec = cases.build(name)
    if spec["thin_by"] != 1:
        pytest.skip("thinned fixture")
    nst = spec["nsteps"]
    ens = make_ens(spec, g["p0"])
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(rng_from_fixture(g).get_state())
    ens.set_tuning("small_kernel", 0)          # the general path (the one-workgroup kernel has its own bulk plan producer)
    ens.chain_config(nst)
    done = 0
    pattern = [("run", 3), ("step", 1), ("run", 2), ("inline", 2), ("step", 1)]
    k = 0
    while done < nst:
        kind, n = pattern[k % len(pattern)]
        n = min(n, nst - done)
        k += 1
        if kind == "step":
            _, S = ens.step_begin(store=True)
            for s in range(S):
                ens.halfstep(s)
            ens.step_end()
        else:
            ens.set_tuning("mt_pipeline", 0 if kind == "inline" else -1)
            ens.run(n, 1, True)
        done += n
    assert ens.status() == 0
    chain = ens.chain_read(0, 0, nst)
    if has_snooker(spec):
        np.testing.assert_allclose(chain, g["chain"], rtol=SNOOKER_RTOL, atol=SNOOKER_ATOL)
    else:
        assert np.array_equal(chain, g["chain"])
    assert np.array_equal(ens.accepted_counts(), g["accepted_count"])
    st = ens.get_mt19937()
    assert np.array_equal(st[1], g["rng_key1"]) and st[2] == int(g["rng_pos1"])
    assert st[3] == int(g["rng_has_gauss1"]) and st[4] == float(g["rng_cached1"])
    ens.close()


def test_large_device_to_host_copies_through_the_pinned_pipeline():
    """Copies of more than 2 MB into ordinary memory cross PCIe through two pinned 8 MB halves (big_copy_to_host): a
    30 MB chain read in one call (four pieces, the last one partial) equals the same chain read row by row (1 MB rows:
    the direct path); rows of 4 MB read with a stride (one pipeline run per row) equal the contiguous read."""
    for N, D, K in ((4096, 32, 30), (65536, 8, 5)):
        ens = _dev()(N, D)
        ens.set_target(_lib.TARGET_ISO)
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(11, 0)
        ens.set_state(np.random.RandomState(0).randn(N, D))
        ens.eval_state_log_prob()
        ens.chain_config(K)
        ens.run(K, 1, True)
        full = ens.chain_read(0, 0, K)
        for s in range(K):
            assert np.array_equal(full[s], ens.chain_read(0, s, s + 1)[0]), s
        assert np.array_equal(full[1::2], ens.chain_read(0, 1, K, 2))
        assert np.array_equal(full[-1], ens.get_state()[0])
        assert np.array_equal(ens.chain_read(1, 0, K)[-1], ens.get_state()[1])
        ens.close()
