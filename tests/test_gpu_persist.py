"""The persistent half-step kernel (k_persist, include/emx.h emx_persist_info): 16 steps per launch, a device-wide barrier where
the kernel boundaries were.  It must be the path emx_run takes for the headline shape, and give the bits of the launch-per-
half-step kernels -- which tests/test_gpu_full_size.py and tests/test_gpu_parity.py hold equal to the oracle -- and of the oracle
itself (red_blue.py:55-106 with stretch.py:27-34) at the BASELINE size."""
import numpy as np
import pytest

from emcee_amd import _lib
from oracle import cases
from oracle import sampler_oracle as so

from emx_testlib import move_desc, philox_plan
from test_gpu_full_size import FULL, full_spec
from test_gpu_parity import assert_lp_close, make_ens

pytestmark = pytest.mark.gpu

S = so.MoveSpec
SEED = 0x5EED5


def dense_spec(N, D, seed=3):
    return full_spec(N, D, "dense", [S("stretch")], seed=seed)


def native_ens(spec, persist, step=0):
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(SEED, step)
    ens.set_tuning("persist", persist)
    ens.set_tuning("persist_timeout_ms", 100)        # (a test that deadlocks should fail in a moment, not after seconds per barrier)
    return ens


def run_both(spec, nsteps, thin_by=1, store=False, calls=1):
    out = []
    for persist in (1, 0):
        ens = native_ens(spec, persist)
        if store:
            ens.chain_config(nsteps * calls)
        for _ in range(calls):
            ens.run(nsteps, thin_by, store)
        assert ens.status() == 0
        info = ens.persist_info()
        x, lp = ens.get_state()
        rec = dict(x=x, lp=lp, acc=ens.accepted_mask(), info=info)
        if store:
            rec["chain"] = ens.chain_read(0, 0, nsteps * calls)
            rec["chain_lp"] = ens.chain_read(1, 0, nsteps * calls)
            rec["counts"] = ens.accepted_counts()
        ens.close()
        out.append(rec)
    return out


@pytest.mark.parametrize("N,D", [(65536, 64), (49152, 64), (32768, 64), (4096, 64), (1024, 64), (512, 64), (16384, 50), (8192, 62), (2080, 64),
                                 (65536, 32), (8192, 16), (4096, 10), (16384, 24), (8192, 48), (2048, 40)])
def test_persistent_kernel_gives_the_bits_of_the_launch_per_halfstep_path(N, D):
    """37 steps (20 + 17: a full and a partial launch, both across the plan batches of sixteen steps), no chain: coordinates, log-probs and the last accept marks bit-equal;
    the persistent run really was persistent (launch and half-step counters), the control really was not.  The sizes cover
    every workgroup shape of the persistent grid (8, 4, 2 and 1 waves: about one workgroup per CU) and every row layout
    k_persist is instantiated for (even ndim up to 64: padded to 16, 32, 48, 64)."""
    spec = dense_spec(N, D)
    p, c = run_both(spec, 37)
    assert p["info"]["qualifies"] and p["info"]["halfsteps"] == 74 and p["info"]["launches"] == 2          # (40 half-steps a launch since round 6)
    assert c["info"]["launches"] == 0
    assert np.array_equal(p["x"], c["x"])
    assert np.array_equal(p["lp"], c["lp"])
    assert np.array_equal(p["acc"], c["acc"])


@pytest.mark.parametrize("N,D,store,thin_by,move", [(512, 64, False, 1, "stretch"), (1024, 64, True, 1, "stretch"), (2048, 32, False, 1, "stretch"),
                                                    (4096, 48, True, 3, "stretch"), (8192, 64, False, 1, "stretch"), (8192, 16, False, 1, "stretch"),
                                                    (1024, 62, False, 1, "stretch"), (6144, 64, True, 1, "stretch"),
                                                    (1024, 64, False, 1, "de"), (4096, 32, True, 1, "de"), (8192, 64, False, 1, "de"),
                                                    (2048, 64, False, 1, "snooker"), (8192, 48, True, 1, "snooker")])
def test_one_xcd_form_and_device_wide_form_agree_with_the_launch_per_halfstep_path(N, D, store, thin_by, move):
    """Ensembles of up to 8 192 walkers (stretch move) take the ONE-XCD form of the persistent kernel (k_persist<..., LOCAL>: an
    eight times larger grid of which every eighth workgroup works, all on one XCD; plain stores, sc1 loads answered by that XCD's
    L2, a flag barrier of its own -- include/emx.h emx_persist_local_launches); tuning persist_local = 0 keeps the device-wide
    form.  Both must give the bits of the launch-per-half-step path: two calls of 37 steps (full and partial launches)."""
    spec = full_spec(N, D, "dense", [S(move)], seed=3)
    recs = {}
    for name, persist, local in (("local", 1, 1), ("wide", 1, 0), ("plain", 0, 0)):
        ens = native_ens(spec, persist)
        ens.set_tuning("persist_local", local)
        if store:
            ens.chain_config(74)
        for _ in range(2):
            ens.run(37, thin_by, store)
        assert ens.status() == 0
        x, lp = ens.get_state()
        rec = dict(x=x, lp=lp, acc=ens.accepted_mask(), info=ens.persist_info())
        if store:
            rec.update(chain=ens.chain_read(0, 0, 74), chain_lp=ens.chain_read(1, 0, 74), counts=ens.accepted_counts())
        ens.close()
        recs[name] = rec
    nl = recs["local"]["info"]
    assert nl["local_launches"] == nl["launches"] > 0 and nl["recovered"] == 0
    assert recs["wide"]["info"]["local_launches"] == 0 and recs["wide"]["info"]["launches"] > 0
    assert recs["plain"]["info"]["launches"] == 0
    for key in recs["plain"]:
        if key != "info":
            assert np.array_equal(recs["local"][key], recs["plain"][key]), "one-XCD form: " + key
            assert np.array_equal(recs["wide"][key], recs["plain"][key]), "device-wide form: " + key


@pytest.mark.parametrize("N,D,target,moves,weights,store,thin_by", [
    (2048, 10, "iso", [S("stretch")], None, False, 1),
    (1024, 5, "iso", [S("stretch")], None, True, 1),                       # odd ndim: one coordinate per lane and chunk
    (8192, 32, "rosenbrock", [S("stretch")], None, False, 1),
    (4096, 64, "diag", [S("stretch")], None, True, 3),
    (2048, 31, "diag", [S("de")], None, False, 1),
    (4096, 16, "rosenbrock", [S("snooker")], None, True, 1),
    (2048, 7, "box", [S("stretch")], None, False, 1),
    (4096, 24, "iso", [S("stretch"), S("de")], [0.6, 0.4], True, 1),       # a mixture: runs of each move in launches of their own
    (8192, 64, "iso", [S("de"), S("snooker")], [0.8, 0.2], False, 1),
    (1024, 8, "iso", [S("stretch")], None, True, 1),                       # rows of 4 lanes: ndim <= 4, even ndim <= 8
    (2048, 3, "diag", [S("stretch")], None, True, 2),
    (4096, 2, "rosenbrock", [S("stretch")], None, False, 1),
    (2048, 6, "diag", [S("de")], None, True, 1),
    (2048, 4, "iso", [S("snooker")], None, False, 1),
    (512, 1, "box", [S("stretch")], None, True, 1),
])
def test_element_wise_targets_run_persistently_on_one_xcd(N, D, target, moves, weights, store, thin_by):
    """csrc/emx_pvalu.hip: ensembles of up to 8 192 walkers on an element-wise target (isotropic / diagonal Gaussian, Rosenbrock, box)
    take the one-XCD persistent kernel k_persist_valu -- rows, partners and log-probs by sc1 loads behind a flag barrier of that
    XCD, proposal / target / decision / commit from registers, up to 32 half-steps a launch.  Same device functions in the same
    order as k_halfstep: coordinates, log-probs, accept marks, chain rows and accept counters bit-equal to the per-half-step
    launches (tuning persist = 0), over two calls of 37 steps."""
    spec = full_spec(N, D, target, moves, weights=weights, seed=5, p0="rosen" if target == "rosenbrock" else ("uniform" if target == "box" else "randn"))
    recs = []
    for persist in (1, 0):
        ens = native_ens(spec, persist)
        ens.set_tuning("small_kernel", 0)
        if store:
            ens.chain_config(74)
        for _ in range(2):
            ens.run(37, thin_by, store)
        assert ens.status() == 0
        x, lp = ens.get_state()
        rec = dict(x=x, lp=lp, acc=ens.accepted_mask(), info=ens.persist_info())
        if store:
            rec.update(chain=ens.chain_read(0, 0, 74), chain_lp=ens.chain_read(1, 0, 74), counts=ens.accepted_counts())
        ens.close()
        recs.append(rec)
    p, c = recs
    assert p["info"]["qualifies"] and p["info"]["local_launches"] == p["info"]["launches"] > 0 and p["info"]["recovered"] == 0
    assert c["info"]["launches"] == 0
    assert p["acc"].any()
    for key in c:
        if key != "info":
            assert np.array_equal(p[key], c[key]), key


@pytest.mark.parametrize("thin_by", [1, 3])
def test_persistent_kernel_stores_the_chain(thin_by):
    """the chain row, its log-probs and the per-walker accept counters of stored steps (backend.py:229), two calls, thinning"""
    spec = dense_spec(8192, 64, seed=4)
    p, c = run_both(spec, 19, thin_by=thin_by, store=True, calls=2)
    assert p["info"]["halfsteps"] == 2 * 2 * 19 * thin_by and c["info"]["launches"] == 0
    for key in ("x", "lp", "chain", "chain_lp", "counts"):
        assert np.array_equal(p[key], c[key]), key
    assert p["counts"].sum() > 0


@pytest.mark.parametrize("N,store", [(65536, False), (8192, True), (1024, False)])
def test_de_move_runs_persistently_too(N, store):
    """DEMove (de.py:40-64: two partner rows, gamma from the plan): the k_persist<..., MOVE_DE> instantiation, same bits"""
    spec = full_spec(N, 64, "dense", [S("de")], seed=8)
    p, c = run_both(spec, 37, store=store)
    assert p["info"]["qualifies"] and p["info"]["halfsteps"] == 74 and c["info"]["launches"] == 0
    for key in ("x", "lp", "acc") + (("chain", "chain_lp", "counts") if store else ()):
        assert np.array_equal(p[key], c[key]), key


@pytest.mark.parametrize("N,D,store", [(65536, 64, False), (8192, 64, True), (2048, 32, False), (4096, 48, True)])
def test_snooker_move_runs_persistently_too(N, D, store):
    """DESnookerMove (de_snooker.py:31-46: four splits, three partner rows, norms and dots by group reductions): the
    k_persist<..., MOVE_SNOOKER> instantiation loads the whole 16-row tile's rows in ONE round trip where k_halfstep takes two
    -- the same arithmetic per walker, the same bits"""
    spec = full_spec(N, D, "dense", [S("snooker")], seed=9)
    p, c = run_both(spec, 21, store=store)
    assert p["info"]["qualifies"] and p["info"]["halfsteps"] == 4 * 21 and p["info"]["launches"] == 3 and c["info"]["launches"] == 0
    for key in ("x", "lp", "acc") + (("chain", "chain_lp", "counts") if store else ()):
        assert np.array_equal(p[key], c[key]), key


def test_mixture_takes_the_persistent_kernel_for_the_runs_it_can():
    """BASELINE config 4 (DEMove 0.8 + DESnookerMove 0.2): the consecutive steps of one move share a persistent launch (two
    half-steps per DE step, four per snooker step); the chain is the one the per-half-step path alone produces"""
    spec = FULL["c4_65536x64_dense_de_snooker"]()
    nst = 40
    p, c = run_both(spec, nst, store=True)
    lib = _lib.load()
    from emx_testlib import cdf_of
    cdf = cdf_of(spec["weights"], len(spec["moves"]))
    n_de = sum(1 for step in range(nst) if spec["moves"][lib.emx_host_move_choice_philox(SEED, step, cdf, len(cdf))].kind == "de")
    assert 0 < n_de < nst
    assert p["info"]["halfsteps"] == 2 * n_de + 4 * (nst - n_de) and 2 <= p["info"]["launches"] <= nst and c["info"]["launches"] == 0
    for key in ("x", "lp", "acc", "chain", "chain_lp", "counts"):
        assert np.array_equal(p[key], c[key]), key


@pytest.mark.parametrize("N,D,mode,factor,store", [(65536, 64, "vector", None, False), (8192, 64, "random", 1.3, True), (4096, 48, "sequential", 2.0, True),
                                                   (2048, 32, "vector", None, False), (1024, 16, "vector", 1.1, True), (32768, 64, "vector", None, True)])
def test_gaussian_move_keeps_its_walkers_in_registers(N, D, mode, factor, store):
    """GaussianMove / MHMove (gaussian.py:76-101, mh.py:57-77) on the fused dense target: k_persist_gauss runs up to 16 steps per
    launch with every walker in registers and no synchronisation at all -- the chain of the per-step launches, bit for bit, in
    the three proposal modes, with and without the step-size factor"""
    rs = np.random.RandomState(D + N)
    mv = S("gaussian", cov=(0.5 + rs.rand(D)) / D, mode=mode, factor=factor)
    mu, cov_, icov = cases._dense_params(D, 5)
    spec = dict(N=N, D=D, moves=[mv], weights=None, desc=dict(kind="dense", mu=mu, cov=cov_, icov=icov))
    p0 = mu + 0.1 * rs.randn(N, D)
    out = []
    for persist in (1, 0):
        ens = make_ens(spec, p0)
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(SEED, 0)
        ens.set_tuning("persist", persist)
        if store:
            ens.chain_config(37)
        ens.run(37, 1, store)
        assert ens.status() == 0
        x, lp = ens.get_state()
        rec = dict(x=x, lp=lp, acc=ens.accepted_mask(), info=ens.persist_info())
        if store:
            rec.update(chain=ens.chain_read(0, 0, 37), chain_lp=ens.chain_read(1, 0, 37), counts=ens.accepted_counts())
        ens.close()
        out.append(rec)
    p, c = out
    assert p["info"]["qualifies"] and p["info"]["launches"] == 3 and p["info"]["halfsteps"] == 37 and c["info"]["launches"] == 0
    for key in ("x", "lp", "acc") + (("chain", "chain_lp", "counts") if store else ()):
        assert np.array_equal(p[key], c[key]), key
    assert 0.0 < p["acc"].mean() < 1.0


@pytest.mark.parametrize("N,D,weights,store,thin_by", [
    (65536, 64, [0.8, 0.2], False, 1),        # C4: the device-wide form, 8 waves a workgroup (4 of them work in a snooker half-step)
    (32768, 48, [0.5, 0.5], True, 1),         # 4 waves a workgroup
    (16384, 64, [0.3, 0.7], True, 1),
    (8192, 64, [0.6, 0.4], True, 1),          # the one-XCD form from here on
    (4096, 32, [0.7, 0.3], False, 1),
    (2048, 64, [0.5, 0.5], False, 1),         # 2 waves a workgroup: the first of them
    (1024, 16, [0.4, 0.6], False, 1),         # 1 wave a workgroup: every other workgroup
    (1024, 32, [0.4, 0.6], True, 1),
    (4096, 48, [0.5, 0.5], True, 3),
    (512, 64, [0.8, 0.2], False, 1),
])
def test_de_and_snooker_steps_share_launches(N, D, weights, store, thin_by):
    """A schedule of DEMove (two splits) and DESnookerMove (four) on the dense target: k_persist_mix takes the steps of both in ONE
    launch -- the move of a half-step is a field of its descriptor, the grid is the DE move's and a snooker half-step uses half its
    waves.  Bit-equal to launches of one move each (tuning persist_mix = 0) and to the per-half-step path, in far fewer launches."""
    spec = full_spec(N, D, "dense", [S("de", gammas=0.0), S("snooker", gammas=1.3)], weights=weights, seed=31)     # (the snooker scale is per half-step)
    nst = 45
    recs = []
    for persist, mix in ((1, 1), (1, 0), (0, 0)):
        ens = native_ens(spec, persist)
        ens.set_tuning("persist_mix", mix)
        if store:
            ens.chain_config(nst)
        ens.run(nst, thin_by, store)               # (nst stored steps of thin_by proposals each)
        assert ens.status() == 0
        x, lp = ens.get_state()
        rec = dict(x=x, lp=lp, acc=ens.accepted_mask(), info=ens.persist_info())
        if store:
            rec.update(chain=ens.chain_read(0, 0, nst), chain_lp=ens.chain_read(1, 0, nst), counts=ens.accepted_counts())
        ens.close()
        recs.append(rec)
    m, p, c = recs
    assert m["info"]["halfsteps"] == p["info"]["halfsteps"] > 2 * nst and c["info"]["launches"] == 0 and m["info"]["recovered"] == 0
    assert m["info"]["launches"] <= 8 * thin_by and m["info"]["launches"] < p["info"]["launches"]       # (32 half-steps a launch at most)
    for key in c:
        if key != "info":
            assert np.array_equal(m[key], c[key]), "mixed launches: " + key
            assert np.array_equal(p[key], c[key]), "one move per launch: " + key


@pytest.mark.parametrize("N,target", [(65536, "dense"), (2048, "dense"), (2048, "iso")])
def test_two_snooker_moves_of_different_scale(N, target):
    """DESnookerMove(gammas=1.7) and DESnookerMove(gammas=0.9) in one schedule (de_snooker.py:26-29): the scale is a launch-wide
    argument of the one-move kernels, so a run of snooker steps ends where the scale changes; with a DEMove next to them the mixed
    kernel takes the scale of every half-step from its descriptor.  The chain is the per-half-step path's."""
    for moves, weights in (([S("snooker", gammas=1.7), S("snooker", gammas=0.9)], [0.5, 0.5]),
                           ([S("de", gammas=0.0), S("snooker", gammas=1.7), S("snooker", gammas=0.9)], [0.4, 0.3, 0.3])):
        spec = full_spec(N, 64 if target == "dense" else 24, target, moves, weights=weights, seed=17)
        p, c = run_both(spec, 40, store=True)
        assert p["info"]["launches"] >= 2 and c["info"]["launches"] == 0
        for key in ("x", "lp", "acc", "chain", "chain_lp", "counts"):
            assert np.array_equal(p[key], c[key]), key


def test_three_move_mixture_shares_launches_where_it_can():
    """StretchMove + DEMove + DESnookerMove: the stretch steps keep launches of their own (their kernel defers the chain rows), the
    runs of DE / snooker steps between them share theirs; the chain is the per-half-step path's"""
    spec = full_spec(4096, 64, "dense", [S("stretch"), S("de", gammas=0.0), S("snooker", gammas=2.1)], weights=[0.2, 0.5, 0.3], seed=8)
    p, c = run_both(spec, 60, store=True)
    assert p["info"]["launches"] >= 3 and c["info"]["launches"] == 0
    for key in ("x", "lp", "acc", "chain", "chain_lp", "counts"):
        assert np.array_equal(p[key], c[key]), key


def test_mixture_whose_moves_take_grids_of_different_sizes():
    """8 192 walkers: a DE half-step is 256 one-wave workgroups, a snooker half-step 128 -- the barrier's arrival counters start
    afresh when the grid changes (a barrier that waited for the other grid's count would time out: status bit 3)"""
    spec = full_spec(8192, 64, "dense", [S("de"), S("snooker")], weights=[0.6, 0.4], seed=10)
    p, c = run_both(spec, 40, store=True)
    assert p["info"]["launches"] >= 2 and c["info"]["launches"] == 0
    for key in ("x", "lp", "acc", "chain", "chain_lp", "counts"):
        assert np.array_equal(p[key], c[key]), key


def test_persistent_kernel_equals_the_oracle_at_the_headline_size():
    """BASELINE config 2 through emx_run (one launch of 3 steps): the oracle's arithmetic applied with the host twin of the
    Philox plans gives the same chain, bit for bit, and the same accept counters."""
    spec = FULL["c2_65536x64_dense_stretch"]()
    fn = cases.make_target(spec["desc"])
    N, D = spec["N"], spec["D"]
    nst = 3
    ens = native_ens(spec, 1)
    ens.chain_config(nst)
    x, lp = ens.get_state()
    ens.run(nst, 1, True)
    assert ens.status() == 0
    info = ens.persist_info()
    assert info["launches"] == 1 and info["halfsteps"] == 2 * nst
    chain = ens.chain_read(0, 0, nst)
    chain_lp = ens.chain_read(1, 0, nst)
    counts = np.zeros(N, dtype=np.int64)
    mv = spec["moves"][0]
    for step in range(nst):
        plan = philox_plan(SEED, step, N, move_desc(mv, D))
        counts += so.propose_planned(x, lp, fn, plan, mv)
        assert np.array_equal(chain[step], x), step
        assert_lp_close(chain_lp[step], lp, 1e-11)
    assert np.array_equal(ens.accepted_counts(), counts)
    ens.close()


def test_persistent_launches_and_the_step_api_interleave():
    """persistent runs, the step API, snapshots, another move set and back: one state, whichever kernel touches it"""
    spec = dense_spec(4096, 64, seed=5)
    ens = native_ens(spec, 1)
    ref = native_ens(spec, 0)
    for e in (ens, ref):
        e.run(20, 1, False)
    slot = ens.snapshot_save()
    for e in (ens, ref):
        for _ in range(2):
            k, nsplit = e.step_begin(store=False)
            for s in range(nsplit):
                e.halfstep(s)
            e.step_end()
        e.run(7, 1, False)
    a, b = ens.get_state(), ref.get_state()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ens.snapshot_restore(slot)
    ens.set_philox(SEED, 20)
    ref2 = native_ens(spec, 0)
    ref2.run(20, 1, False)
    sn = [move_desc(S("stretch", nsplits=3), 64)]         # three splits: a step the persistent kernel has no instantiation for
    for e in (ens, ref2):
        e.set_moves(sn, np.array([1.0]))
        e.run(5, 1, False)
    assert not ens.persist_info()["qualifies"]
    launches = ens.persist_info()["launches"]
    st = [move_desc(S("stretch"), 64)]
    for e in (ens, ref2):
        e.set_moves(st, np.array([1.0]))
        e.run(16, 1, False)
    assert ens.persist_info()["launches"] == launches + 1
    a, b = ens.get_state(), ref2.get_state()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for e in (ens, ref, ref2):
        assert e.status() == 0
        e.close()


def test_what_does_not_qualify():
    # (round 6: ndim 66 ... 128 even and odd ndim up to 63 do qualify now -- k_persist_slab, emx_podd.hip: tests/test_gpu_persist_slab.py)
    for N, D, why in ((1000, 64, "half an ensemble of whole 16-walker tiles"), (4096, 130, "ndim above 128: the wide path"), (4096, 129, "odd ndim above 128"),
                      (131072, 64, "more tiles than waves"), (480, 64, "below persist_min_walkers"), (65536, 128, "the per-half-step slab kernel is level there")):
        ens = native_ens(dense_spec(N, D), 1)
        assert not ens.persist_info()["qualifies"], why
        ens.run(16, 1, False)
        assert ens.persist_info()["launches"] == 0 and ens.status() == 0
        ens.close()
    # exact (MT19937) mode: only where the one-XCD form runs (the host pipeline's plans, test_exact_mode_* below)
    # (device-wide form: up to persist_exact_max_walkers = 32 768 for plans that travel finished -- 49 152: half an ensemble that is no
    # power of two -- and up to persist_exact_regen_max_walkers where the stretch steps go up as generator states, round 6: 65 536)
    for N, want in ((4096, True), (16384, True), (65536, True), (49152, False)):
        ens = native_ens(dense_spec(N, 64), 1)
        ens.set_rng_mode(_lib.RNG_MT19937)
        assert ens.persist_info()["qualifies"] == want, N
        ens.close()


def test_two_ensembles_of_one_process_take_turns():
    """two persistent grids that each held half of the device would wait for each other: the launches of a process are chained"""
    specs = [dense_spec(65536, 64, seed=6), dense_spec(32768, 64, seed=7)]
    ens = [native_ens(s, 1) for s in specs]
    for _ in range(6):
        for e in ens:
            e.run(16, 1, False)          # asynchronous: both streams hold work
    got = []
    for e in ens:
        assert e.status() == 0
        assert e.persist_info()["launches"] == 6
        got.append(e.get_state())
        e.close()
    for s, g in zip(specs, got):
        ref = native_ens(s, 0)
        ref.run(96, 1, False)
        x, lp = ref.get_state()
        assert np.array_equal(g[0], x) and np.array_equal(g[1], lp)
        ref.close()


def test_a_grid_that_cannot_become_co_resident_is_redone_not_void():
    """Co-residency is checked, not assumed: every launch opens with a handshake barrier BEFORE its first store.  A handshake that is
    not met in time (here: the test skews the count it waits for; in the field: another process's persistent grid holds the CUs)
    leaves the ensemble untouched, sends home every persistent launch queued behind it, and the host takes those launches' steps
    again on the per-half-step path at the next sync / status / read: the same bits as a run that never tried, no status bit, no
    void state.  The wait is bounded by the wall clock (one time-out per run of launches, not one per barrier)."""
    import time
    for store, thin_by, local in ((False, 1, 1), (True, 2, 1), (False, 1, 0), (True, 2, 0)):       # (4 096 walkers: the one-XCD form unless told otherwise)
        spec = dense_spec(4096, 64, seed=11)
        ens = native_ens(spec, 1)
        ens.set_tuning("persist_local", local)
        ref = native_ens(spec, 0)
        nst = 40
        for e in (ens, ref):
            if store:
                e.chain_config(2 * nst)
            e.run(nst, thin_by, store)                           # a first call both ways: the second starts from a moved state
        assert ens.persist_info()["launches"] > 0
        ens.set_tuning("persist_timeout_ms", 20)
        ens.set_tuning("persist_test_skew", 1000)
        t0 = time.perf_counter()
        ens.run(nst, thin_by, store)                             # three launches queued: the first times out, the others see the mark
        ens.sync()                                               # ... and all three are redone here
        assert time.perf_counter() - t0 < 1.0
        ref.run(nst, thin_by, store)
        info = ens.persist_info()
        assert info["recovered"] >= 2 and ens.status() == 0
        a, b = ens.get_state(), ref.get_state()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(ens.accepted_mask(), ref.accepted_mask())
        if store:
            assert np.array_equal(ens.chain_read(0, 0, 2 * nst), ref.chain_read(0, 0, 2 * nst))
            assert np.array_equal(ens.chain_read(1, 0, 2 * nst), ref.chain_read(1, 0, 2 * nst))
            assert np.array_equal(ens.accepted_counts(), ref.accepted_counts())
        # the persistent path is off for this context from here on (whatever held the CUs may still be there) ...
        n0 = ens.persist_info()["launches"]
        ens.set_tuning("persist_test_skew", 0)
        ens.run(16, 1, False)
        ref.run(16, 1, False)
        assert ens.persist_info()["launches"] == n0
        # ... until it is asked for again
        ens.set_tuning("persist", 1)
        ens.run(16, 1, False)
        ref.run(16, 1, False)
        # (one launch per batch of prepared plans: the 16 steps may straddle two batches left over from the earlier calls)
        assert n0 + 1 <= ens.persist_info()["launches"] <= n0 + 2 and ens.status() == 0
        a, b = ens.get_state(), ref.get_state()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        ens.close()
        ref.close()


def test_the_persistent_grid_is_checked_against_the_occupancy_calculator():
    """hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs >= workgroups for every shape the persistent path takes (it would fall
    back to the per-half-step launches otherwise): the shapes of the parity test above all qualify on an MI355X"""
    for N, D in [(65536, 64), (49152, 64), (4096, 64), (512, 64), (8192, 16)]:
        ens = native_ens(dense_spec(N, D), 1)
        assert ens.persist_info()["qualifies"]
        ens.run(16, 1, False)
        assert ens.persist_info()["launches"] == 1 and ens.status() == 0
        ens.close()


def test_sampler_runs_persistently(monkeypatch):
    """EnsembleSampler.run_mcmc with the device target and rng="philox" takes the persistent path by itself; EMX_TUNE turns it off"""
    import emcee_amd
    from emcee_amd import targets
    rs = np.random.RandomState(8)
    A = rs.randn(64, 64)
    icov = A @ A.T / 64 + np.eye(64)
    mu = rs.randn(64)
    p0 = rs.randn(2048, 64)
    chains = []
    for tune in (None, "persist=0"):
        if tune:
            monkeypatch.setenv("EMX_TUNE", tune)
        np.random.seed(9)
        s = emcee_amd.EnsembleSampler(2048, 64, targets.DenseGaussian(mu, icov), rng="philox")
        s.run_mcmc(p0, 40)
        info = s.backend._dev.persist_info()
        assert (info["launches"] > 0) == (tune != "persist=0")
        chains.append((s.get_chain().copy(), s.get_log_prob().copy(), s.acceptance_fraction.copy()))
    for a, b in zip(*chains):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("N,trials,store,thin_by", [(65536, 110, False, 1), (32768, 110, False, 1), (8192, 110, False, 1),
                                                    (8192, 24, True, 1), (32768, 12, True, 3),
                                                    (2048, 150, False, 1), (512, 150, False, 1), (4096, 40, True, 1)])     # (<= 8192: the one-XCD form)
def test_persistent_kernel_coherence_stress(N, trials, store, thin_by):
    """The class of bug profiles/r03/persist_coherence.txt records (a variant of k_persist's barrier / sc1 protocol that was wrong
    once in ~120 runs) does not show in a single 37-step run: many short runs do.  366 trials x 50 steps in all, fresh Philox
    seed each, every trial bit-compared with the launch-per-half-step path started from the same state (both ensembles carry
    their own state from trial to trial: one differing bit fails the trial it appears in).  ~30 s."""
    _stress(N, trials, store, thin_by)


def _stress(N, trials, store, thin_by):
    nsteps = 50
    spec = dense_spec(N, 64, seed=11)
    ens = []
    for persist in (1, 0):
        e = native_ens(spec, persist)
        if store:
            e.chain_config(nsteps)
        ens.append(e)
    for t in range(trials):
        got = []
        for e in ens:
            e.set_philox(0xABC000 + 7919 * t + N, 0)
            if store:
                e.chain_reset()
            e.run(nsteps, thin_by, store)
            x, lp = e.get_state()
            rec = [x, lp, e.accepted_mask()]
            if store:
                rec += [e.chain_read(0, 0, nsteps), e.chain_read(1, 0, nsteps), e.accepted_counts()]
            assert e.status() == 0
            got.append(rec)
        for k, (a, b) in enumerate(zip(*got)):
            assert np.array_equal(a, b), "trial %d of %d: output %d of the persistent kernel differs from the per-half-step path" % (t, trials, k)
    p, c = ens[0].persist_info(), ens[1].persist_info()
    assert p["halfsteps"] == 2 * nsteps * thin_by * trials and c["launches"] == 0
    assert (p["local_launches"] == p["launches"]) == (N <= 8192) and p["recovered"] == 0
    for e in ens:
        e.close()


@pytest.mark.parametrize("N,D,target,trials", [(2048, 10, "iso", 150), (8192, 32, "rosenbrock", 60), (1024, 5, "iso", 150)])
def test_element_wise_persistent_kernel_coherence_stress(N, D, target, trials):
    """the one-XCD protocol of k_persist_valu (plain stores, sc1 loads, the flag barrier) under many short runs: 360 trials x 50 steps,
    fresh Philox seed each, every trial bit-compared with the per-half-step path started from the same state"""
    nsteps = 50
    spec = full_spec(N, D, target, [S("stretch")], seed=17, p0="rosen" if target == "rosenbrock" else "randn")
    ens = []
    for persist in (1, 0):
        e = native_ens(spec, persist)
        e.set_tuning("small_kernel", 0)
        ens.append(e)
    for t in range(trials):
        got = []
        for e in ens:
            e.set_philox(0xBEE000 + 104729 * t + N, 0)
            e.run(nsteps, 1, False)
            x, lp = e.get_state()
            assert e.status() == 0
            got.append([x, lp, e.accepted_mask()])
        for k, (a, b) in enumerate(zip(*got)):
            assert np.array_equal(a, b), "trial %d of %d: output %d of the persistent kernel differs from the per-half-step path" % (t, trials, k)
    p, c = ens[0].persist_info(), ens[1].persist_info()
    assert p["halfsteps"] == 2 * nsteps * trials and p["local_launches"] == p["launches"] and p["recovered"] == 0 and c["launches"] == 0
    for e in ens:
        e.close()


@pytest.mark.parametrize("N,D,target,move,store,thin_by", [
    (1024, 64, "dense", "stretch", True, 1), (4096, 64, "dense", "stretch", False, 1), (8192, 32, "dense", "de", False, 1),
    (2048, 10, "iso", "stretch", True, 3), (1024, 5, "iso", "stretch", False, 1), (4096, 32, "rosenbrock", "stretch", True, 1),
    (2048, 16, "diag", "snooker", False, 1), (512, 64, "dense", "stretch", False, 1),
])
def test_exact_mode_takes_the_one_xcd_persistent_kernels(N, D, target, move, store, thin_by):
    """rng = MT19937 (the Python default: same seed => reference emcee's chain), one move, 512 ... 8 192 walkers: the host pipeline's
    plans are uploaded eight steps ahead and the one-XCD persistent kernels (k_persist<..., LOCAL> / k_persist_valu) run them,
    eight steps a launch.  Coordinates, log-probs, accept marks, chain rows, accept counters and the final generator state equal the
    per-half-step exact path's (tuning persist_exact = 0) bit for bit over three calls of 21 steps."""
    spec = full_spec(N, D, target, [S(move)], seed=9, p0="rosen" if target == "rosenbrock" else "randn")
    state = np.random.RandomState(1234 + N).get_state()
    recs = []
    for pe in (1, 0):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(state)
        ens.set_tuning("persist_exact", pe)
        ens.set_tuning("persist_timeout_ms", 200)
        if store:
            ens.chain_config(63)
        for _ in range(3):
            ens.run(21, thin_by, store)
        assert ens.status() == 0
        x, lp = ens.get_state()
        rec = dict(x=x, lp=lp, acc=ens.accepted_mask(), info=ens.persist_info(), rng=ens.get_mt19937())
        if store:
            rec.update(chain=ens.chain_read(0, 0, 63), chain_lp=ens.chain_read(1, 0, 63), counts=ens.accepted_counts())
        ens.close()
        recs.append(rec)
    p, c = recs
    assert p["info"]["local_launches"] == p["info"]["launches"] > 0 and p["info"]["recovered"] == 0 and c["info"]["launches"] == 0
    assert np.array_equal(p["rng"][1], c["rng"][1]) and p["rng"][2] == c["rng"][2]
    for key in c:
        if key not in ("info", "rng"):
            assert np.array_equal(p[key], c[key]), key


@pytest.mark.parametrize("N,D,target,steps_per_launch", [(1024, 64, "dense", 16), (512, 24, "iso", 16), (4096, 32, "dense", 5), (2048, 10, "iso", 1),
                                                         (16384, 64, "dense", 16), (32768, 48, "dense", 16),       # (the device-wide form)
                                                         (16384, 16, "iso", 16), (32768, 5, "iso", 16), (16384, 32, "rosenbrock", 7), (24576, 8, "diag", 16)])
def test_exact_mode_persistent_long_run(N, D, target, steps_per_launch):
    """700 steps of exact mode on the persistent kernels, the host enqueueing ahead of the device: every plan slot is rewritten
    (k_plan_fetch) some twenty times, each time only after a LATER launch than the slot's last reader is known to have started
    (PersistArgs::started_host).  Final coordinates, log-probs, accept counters and generator state equal the per-half-step path's."""
    spec = full_spec(N, D, target, [S("stretch")], seed=21, p0="rosen" if target == "rosenbrock" else "randn")
    state = np.random.RandomState(99).get_state()
    recs = []
    for pe in (1, 0):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(state)
        ens.set_tuning("persist_exact", pe)
        ens.set_tuning("persist_exact_steps", steps_per_launch)
        ens.run(700, 1, False)
        assert ens.status() == 0
        x, lp = ens.get_state()
        recs.append(dict(x=x, lp=lp, counts=ens.accepted_counts(), rng=ens.get_mt19937(), launches=ens.persist_info()["launches"]))
        ens.close()
    p, c = recs
    assert p["launches"] >= 700 // 16 and c["launches"] == 0
    assert np.array_equal(p["x"], c["x"]) and np.array_equal(p["lp"], c["lp"]) and np.array_equal(p["counts"], c["counts"])
    assert np.array_equal(p["rng"][1], c["rng"][1]) and p["rng"][2] == c["rng"][2]


@pytest.mark.parametrize("N,D,target", [(4096, 64, "dense"), (2048, 10, "iso"), (16384, 64, "dense")])
def test_exact_mode_pipeline_is_restarted_when_its_consumer_changes(N, D, target):
    """Round-5 advisor (medium): the host pipeline is configured once, when it starts -- for step-at-a-time uploads with device finish
    (RAW generator words in the plan columns, finished by k_plan_raw) or for the persistent launches' fetch kernel (finished
    columns) -- while "do the persistent kernels take this call" is asked per emx_run and depends on tuning keys and the target.  A
    pipeline started for one consumer used to go on feeding the other: k_plan_fetch read generator words as doubles.  emx_run now
    retires and restarts it (the generator continues behind the last step taken).  Three calls, the consumer switched between
    them, against one consumer throughout: same chain, same accept counters, same final generator state."""
    spec = full_spec(N, D, target, [S("stretch")], seed=13)
    state = np.random.RandomState(4321).get_state()
    recs = []
    for schedule in ((0, 1, 0), (1, 0, 1), (0, 0, 0)):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(state)
        ens.set_tuning("persist_timeout_ms", 200)
        ens.chain_config(66)
        launches = []
        for pe in schedule:
            ens.set_tuning("persist_exact", pe)
            before = ens.persist_info()["launches"]
            ens.run(22, 1, True)
            launches.append(ens.persist_info()["launches"] - before)
        assert ens.status() == 0
        assert [n > 0 for n in launches] == [bool(pe) for pe in schedule]          # each call really took the consumer asked for
        x, lp = ens.get_state()
        recs.append(dict(x=x, lp=lp, chain=ens.chain_read(0, 0, 66), counts=ens.accepted_counts(), rng=ens.get_mt19937()))
        ens.close()
    ref = recs[-1]
    for r in recs[:-1]:
        assert np.array_equal(r["rng"][1], ref["rng"][1]) and r["rng"][2] == ref["rng"][2]
        for key in ("x", "lp", "chain", "counts"):
            assert np.array_equal(r[key], ref[key]), key


def test_exact_mode_persistent_launch_waits_for_its_plans():
    """The plans of a launch's eight steps are fetched by ONE kernel on the upload stream (k_plan_fetch, straight from the
    pipeline's pinned staging buffers); the launch waits for it, and the fetch waits for the launch that last read the slots it
    rewrites.  With the fetch made to idle 300 us before it reads -- ten times a launch -- the chain is still the per-half-step
    path's bit for bit (a launch that did not wait would read the plans of 32 steps earlier, or nothing)."""
    spec = full_spec(2048, 10, "iso", [S("stretch")], seed=4)
    state = np.random.RandomState(77).get_state()
    recs = []
    for pe, delay in ((1, 300), (0, 0)):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(state)
        ens.set_tuning("persist_exact", pe)
        ens.set_tuning("test_fetch_delay_us", delay)
        ens.chain_config(120)
        ens.run(120, 1, True)
        assert ens.status() == 0
        recs.append(dict(chain=ens.chain_read(0, 0, 120), lp=ens.chain_read(1, 0, 120), launches=ens.persist_info()["launches"],
                         rng=ens.get_mt19937()))
        ens.close()
    p, c = recs
    assert p["launches"] >= 7 and c["launches"] == 0
    assert np.array_equal(p["chain"], c["chain"]) and np.array_equal(p["lp"], c["lp"])
    assert np.array_equal(p["rng"][1], c["rng"][1]) and p["rng"][2] == c["rng"][2]


@pytest.mark.parametrize("N,D,target,moves,weights,store", [
    (4096, 64, "dense", [S("de"), S("snooker")], [0.8, 0.2], True),          # DE + snooker: k_persist_mix (one launch for steps of either)
    (1024, 64, "dense", [S("stretch"), S("de")], [0.5, 0.5], False),         # runs of one move per launch
    (2048, 10, "iso", [S("stretch"), S("de"), S("snooker")], [0.4, 0.4, 0.2], True),      # k_persist_valu
    (16384, 64, "dense", [S("de"), S("snooker")], [0.7, 0.3], False),        # the device-wide form
])
def test_exact_mode_mixtures_take_the_persistent_kernels(N, D, target, moves, weights, store):
    """rng = MT19937 with a MIXTURE of moves (round 5; the reference's own recommended usage, docs/tutorials/moves.ipynb): the move of
    the step that follows is read off the pipeline's plan before it is taken (ensemble.py:406's draw was made ahead anyway), so a
    launch ends where the move changes -- or does not, for DE + snooker (k_persist_mix).  Coordinates, log-probs, accept marks,
    chain rows, accept counters and the final generator state equal the per-half-step exact path's (tuning persist_exact = 0) bit for
    bit over three calls of 21 steps."""
    spec = full_spec(N, D, target, moves, weights=weights, seed=31)
    state = np.random.RandomState(4321 + N).get_state()
    recs = []
    for pe in (1, 0):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(state)
        ens.set_tuning("persist_exact", pe)
        ens.set_tuning("persist_timeout_ms", 200)
        if store:
            ens.chain_config(63)
        for _ in range(3):
            ens.run(21, 1, store)
        assert ens.status() == 0
        x, lp = ens.get_state()
        rec = dict(x=x, lp=lp, acc=ens.accepted_mask(), info=ens.persist_info(), rng=ens.get_mt19937())
        if store:
            rec.update(chain=ens.chain_read(0, 0, 63), chain_lp=ens.chain_read(1, 0, 63), counts=ens.accepted_counts())
        ens.close()
        recs.append(rec)
    p, c = recs
    assert p["info"]["launches"] > 0 and p["info"]["recovered"] == 0 and c["info"]["launches"] == 0
    assert p["info"]["halfsteps"] == sum(1 for _ in range(0)) or p["info"]["halfsteps"] >= 2 * 63        # every step went through a persistent launch
    assert np.array_equal(p["rng"][1], c["rng"][1]) and p["rng"][2] == c["rng"][2]
    for key in c:
        if key not in ("info", "rng"):
            assert np.array_equal(p[key], c[key]), key


@pytest.mark.parametrize("N,D,target,moves,weights", [(4096, 64, "dense", [S("stretch")], None), (2048, 10, "iso", [S("stretch")], None),
                                                      (1024, 64, "dense", [S("de"), S("snooker")], [0.7, 0.3])])
def test_exact_mode_launches_that_cannot_become_resident_are_redone(N, D, target, moves, weights):
    """Round 5: the give-up path in exact mode.  A persistent launch whose handshake is not met (here: the test skews the count it
    waits for; in the field: another process's persistent grid holds the CUs -- two processes per GPU) leaves the ensemble untouched,
    and so do the launches queued behind it; their steps have left the host pipeline, but the pipeline keeps the generator state behind
    each of its last steps: it is taken back to the state in front of the first such launch, and the steps are drawn again on the
    per-half-step path.  Same bits -- coordinates, log-probs, chain, accept counters, the generator state handed back -- as a run
    that never tried; no status bit, no void state (round 4: status bit 3)."""
    spec = full_spec(N, D, target, moves, weights=weights, seed=17)
    state = np.random.RandomState(777 + N).get_state()
    recs = []
    nst = 40
    for pe in (1, 0):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(state)
        ens.set_tuning("persist_exact", pe)
        ens.chain_config(3 * nst)
        ens.run(nst, 1, True)                                # a first call both ways: the second starts from a moved state
        if pe:
            assert ens.persist_info()["launches"] > 0
            ens.set_tuning("persist_timeout_ms", 20)
            ens.set_tuning("persist_test_skew", 1000)
        ens.run(nst, 1, True)                                # the first launch times out, the others see the mark
        ens.sync()                                           # ... and all of them are redone here
        if pe:
            assert ens.persist_info()["recovered"] >= 1
            ens.set_tuning("persist_test_skew", 0)
        ens.run(nst, 1, True)                                # (the persistent path stays off for this context: per-half-step from here on)
        assert ens.status() == 0
        x, lp = ens.get_state()
        recs.append(dict(x=x, lp=lp, chain=ens.chain_read(0, 0, 3 * nst), chain_lp=ens.chain_read(1, 0, 3 * nst), counts=ens.accepted_counts(),
                         rng=ens.get_mt19937()))
        ens.close()
    p, c = recs
    assert np.array_equal(p["rng"][1], c["rng"][1]) and p["rng"][2] == c["rng"][2]
    for key in c:
        if key != "rng":
            assert np.array_equal(p[key], c[key]), key


@pytest.mark.parametrize("N,D,target,moves,weights", [
    (8192, 64, "dense", [S("stretch")], None), (65536, 64, "dense", [S("stretch")], None), (4096, 10, "iso", [S("stretch")], None),
    (4096, 128, "dense", [S("stretch")], None), (4096, 33, "dense", [S("de")], None), (8192, 64, "dense", [S("de"), S("snooker")], [0.8, 0.2]),
])
@pytest.mark.parametrize("calls,nsteps,thin_by,store", [(2, 45, 1, True), (3, 13, 3, False), (2, 19, 3, True), (4, 29, 1, False)])
def test_a_launch_reads_the_plans_of_two_batches_at_most(N, D, target, moves, weights, calls, nsteps, thin_by, store):
    """Round 6: a persistent launch holds up to 40 half-steps (twenty stretch steps), the ring of Philox plans two batches of sixteen
    steps.  A call that starts a few steps before a batch boundary would take plans of THREE batches -- the third written into the half
    of the ring the launch still reads (found by the suite the day the launches grew: every kernel family, the second call of a
    run).  run_persist ends the launch there.  Calls whose starts fall everywhere in a batch, against the per-half-step launches."""
    spec = full_spec(N, D, target, moves, weights=weights, seed=23)
    store = store and N <= 8192                  # (a stored step of 65 536 x 64 is 33.5 MB to read back: the final state says as much there)
    recs = []
    for persist in (1, 0):
        ens = native_ens(spec, persist)
        ens.set_tuning("small_kernel", 0)
        if store:
            ens.chain_config(calls * nsteps)
        for _ in range(calls):
            ens.run(nsteps, thin_by, store)
        assert ens.status() == 0
        x, lp = ens.get_state()
        rec = dict(x=x, lp=lp, acc=ens.accepted_mask(), info=ens.persist_info())
        if store:
            rec.update(chain=ens.chain_read(0, 0, calls * nsteps), counts=ens.accepted_counts())
        ens.close()
        recs.append(rec)
    p, c = recs
    assert p["info"]["launches"] > 0 and c["info"]["launches"] == 0
    for key in c:
        if key != "info":
            assert np.array_equal(p[key], c[key]), key


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,target,moves,weights", [
    (65536, 64, "dense", [S("stretch")], None), (16384, 64, "dense", [S("de"), S("snooker")], [0.8, 0.2]),
])
def test_the_staggered_partner_loads_change_no_bit(N, D, target, moves, weights):
    """Round 6: the waves of SIMDs 2 and 3 ask for their partner rows 256 clocks after the others (persist_stagger_wait; by default in
    device-wide stretch launches without stored rows).  A wait, nothing else: every form of it leaves the same ensemble as none."""
    spec = full_spec(N, D, target, moves, weights=weights, seed=29)
    recs = []
    for stagger in (0, -1, 516, 8, 1027, 300):
        ens = native_ens(spec, 1)
        ens.set_tuning("persist_stagger", stagger)
        for _ in range(2):
            ens.run(21, 1, False)
        assert ens.status() == 0
        x, lp = ens.get_state()
        assert ens.persist_info()["launches"] > 0
        recs.append((x, lp, ens.accepted_mask()))
        ens.close()
    for r in recs[1:]:
        for a, b in zip(recs[0], r):
            assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,thin_by", [(32768, 64, 1), (16384, 48, 2), (65536, 32, 1), (4096, 64, 1), (1024, 32, 3)])
def test_stored_launches_take_the_late_own_rows_and_change_no_bit(N, D, thin_by):
    """Round 6: launches that store chain rows run k_persist<..., ROWS_LATE> (the next half-step's own rows asked for behind the MFMA
    phase; tuning persist_rows_late).  Same chain, log-probs, counts and final state as the other instantiation and as the launches."""
    spec = full_spec(N, D, "dense", [S("stretch")], seed=31)
    nsteps = 6                      # stored steps; thin_by iterations each
    recs = []
    for persist, late in ((1, 2), (1, 0), (0, 1)):
        ens = native_ens(spec, persist)
        ens.set_tuning("persist_rows_late", late)
        ens.chain_config(nsteps)
        ens.run(nsteps, thin_by, True)
        assert ens.status() == 0
        x, lp = ens.get_state()
        assert (ens.persist_info()["launches"] > 0) == bool(persist)
        recs.append(dict(x=x, lp=lp, chain=ens.chain_read(0, 0, nsteps), chain_lp=ens.chain_read(1, 0, nsteps),
                         counts=ens.accepted_counts()))
        ens.close()
    for r in recs[1:]:
        for key, val in recs[0].items():
            if val is not None:
                assert np.array_equal(val, r[key]), key
