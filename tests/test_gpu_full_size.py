"""Parity at the FULL sizes of BASELINE.json's configs (the oracle still finishes in seconds for a
couple of steps), plus size-independent invariants of longer native runs at those sizes."""
import numpy as np
import pytest

from emcee_amd import _lib
from oracle import cases
from oracle import sampler_oracle as so

from emx_testlib import cdf_of, move_desc, philox_plan
from test_gpu_parity import assert_lp_close, make_ens, set_target

pytestmark = pytest.mark.gpu

S = so.MoveSpec


def full_spec(N, D, target, moves, weights=None, seed=0, p0="randn"):
    spec = dict(N=N, D=D, target=target, moves=moves, weights=weights, seed=seed, p0=p0, nsteps=2, thin_by=1)
    cases.DIGEST_CASES["_tmp_full"] = dict(N=N, D=D, target=target, moves=moves, weights=weights, nsteps=2, seed=seed, p0=p0)
    out = cases.build("_tmp_full")
    del cases.DIGEST_CASES["_tmp_full"]
    return out


FULL = {
    "c2_65536x64_dense_stretch": lambda: full_spec(65536, 64, "dense", [S("stretch")], seed=11),
    "c3_262144x32_rosenbrock_stretch": lambda: full_spec(262144, 32, "rosenbrock", [S("stretch")], seed=12, p0="rosen"),
    "c4_65536x64_dense_de_snooker": lambda: full_spec(65536, 64, "dense", [S("de"), S("snooker")], weights=[0.8, 0.2], seed=13),
    "c5_16384x1024_diag_stretch": lambda: full_spec(16384, 1024, "diag", [S("stretch")], seed=14),
    # not a BASELINE configuration: the wide dense path at the size where the role-split log-prob kernel (k_wide_lp_ws) runs
    "wd_65536x130_dense_stretch": lambda: full_spec(65536, 130, "dense", [S("stretch")], seed=15),
}


@pytest.mark.parametrize("name", list(FULL))
def test_exact_mode_equals_oracle_at_full_size(name):
    """MT19937 mode, free running at the BASELINE size -- 24 steps at C2, 20 at C3, 10 at C5 (round-5 verdict: the accept-mask
    evidence at these sizes was 2-3 steps long while `plan_log` replaced the library logarithm in ln u and (D-1) ln zz; the oracle
    costs 0.3-0.8 s a step there), 3 where the oracle's snooker loop is the cost: same accept masks, bit-identical coordinates
    (stretch / DE), same final generator state as the reference-pinned oracle."""
    spec = FULL[name]()
    fn = cases.make_target(spec["desc"])
    rs = np.random.RandomState(spec["rng_seed"])
    nst = {"c2": 24, "c3": 20, "c5": 10}.get(name[:2], 3)
    out = so.run(spec["p0"], nst, fn, rs, moves=spec["moves"], weights=spec["weights"])
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(np.random.RandomState(spec["rng_seed"]).get_state())
    ens.chain_config(nst)
    ens.run(nst, 1, True)
    assert ens.status() == 0
    assert np.array_equal(ens.accepted_counts(), out["accepted_count"])
    chain = ens.chain_read(0, 0, nst)
    used = [spec["moves"][k].kind for k in out["move_choices"]]
    if "snooker" in used:
        np.testing.assert_allclose(chain, out["chain"], rtol=1e-9, atol=1e-10)
    else:
        assert np.array_equal(chain, out["chain"])
    assert_lp_close(ens.chain_read(1, 0, nst), out["log_prob"], 1e-9 if "snooker" in used else 1e-11)
    a, b = ens.get_mt19937(), rs.get_state()
    assert np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and a[4] == b[4]
    ens.close()


@pytest.mark.parametrize("name", list(FULL))
def test_native_mode_invariants_at_full_size(name):
    """Philox mode, 30 steps at the BASELINE size.  Invariants that hold for any correct run:
    (i) stored log-prob == target evaluated at the stored coordinates; (ii) a walker's row changes
    between consecutive stored steps iff it was accepted, and the per-walker accept counters add
    up; (iii) every walker is proposed exactly once per step (the split is a partition);
    (iv) acceptance fraction in the regime the reference shows for this target."""
    spec = FULL[name]()
    fn = cases.make_target(spec["desc"])
    N, D = spec["N"], spec["D"]
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(99, 0)
    nst = 6
    ens.chain_config(nst)
    ens.run(24, 1, False)
    x_prev, _ = ens.get_state()
    changed_total = np.zeros(N)
    for it in range(nst):
        k, nsplits = ens.step_begin(store=True)
        plan = ens.plan_get(nsplits)
        assert np.array_equal(np.sort(plan["order"]), np.arange(N))               # (iii)
        for s in range(nsplits):
            ens.halfstep(s)
        ens.step_end()
        x, lp = ens.get_state()
        acc = ens.accepted_mask()
        moved = np.any(x != x_prev, axis=1)
        assert np.array_equal(moved, acc)                                          # (ii)
        changed_total += acc
        x_prev = x
    assert ens.status() == 0
    chain = ens.chain_read(0, 0, nst)
    lps = ens.chain_read(1, 0, nst)
    assert np.array_equal(chain[-1], x_prev)
    assert np.array_equal(ens.accepted_counts(), changed_total)
    sel = np.random.RandomState(0).choice(N, size=min(N, 4096), replace=False)
    for it in (0, nst - 1):
        assert_lp_close(lps[it][sel], fn(chain[it][sel]), 1e-11)                    # (i)
    frac = changed_total.mean() / nst
    lo, hi = {"c2": (0.10, 0.25), "c3": (0.10, 0.45), "c4": (0.10, 0.45), "c5": (0.005, 0.12), "wd": (0.04, 0.25)}[name[:2]]
    assert lo < frac < hi, frac                                                    # (iv)
    ens.close()


@pytest.mark.parametrize("name", list(FULL))
def test_native_mode_teacher_forced_at_full_size(name):
    """Philox mode -- the mode the bench headline is measured in -- replayed through the oracle AT the BASELINE size.
    For 3 steps: the plan the device evaluated (k_native_plan_batch, read back with emx_plan_get) must equal the host
    twin of the keyed permutation / Philox draws (emx_host_plan_philox) entry for entry, and the oracle's arithmetic
    (stretch.py:33, de.py:53-62, de_snooker.py:41-46, red_blue.py:96-104) applied to the device's pre-step state with
    that plan must give the same accept mask (bit-exact) and the same post-step state (coordinates bit-identical for
    stretch / DE, 1e-9 for snooker whose norms and dots are reductions)."""
    spec = FULL[name]()
    fn = cases.make_target(spec["desc"])
    N, D = spec["N"], spec["D"]
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_PHILOX)
    seed = 0xBEEF03 + spec["seed"]          # c4: steps 0..2 draw DE, snooker, DE
    ens.set_philox(seed, 0)
    cdf = cdf_of(spec["weights"], len(spec["moves"]))
    nst = 3
    used = set()
    nacc = 0
    for step in range(nst):
        x0, lp0 = ens.get_state()
        k, S = ens.step_begin(store=False)
        assert k == ens.lib.emx_host_move_choice_philox(seed, step, cdf, len(cdf))
        mv = spec["moves"][k]
        used.add(mv.kind)
        plan = ens.plan_get(S)
        host = philox_plan(seed, step, N, move_desc(mv, D))
        for key in ("off", "order", "p0", "p1", "p2"):
            assert np.array_equal(plan[key], host[key]), (step, key)
        np.testing.assert_allclose(plan["s0"], host["s0"], rtol=1e-14)
        assert np.array_equal(plan["uacc"], host["uacc"])
        assert np.array_equal(np.sort(plan["order"]), np.arange(N))          # the split is a partition
        for s in range(S):
            ens.halfstep(s)
        ens.step_end()
        assert ens.status() == 0
        x1, lp1 = ens.get_state()
        acc_dev = ens.accepted_mask()
        xo, lpo = x0.copy(), lp0.copy()
        acc_or = so.propose_planned(xo, lpo, fn, plan, mv)
        assert np.array_equal(acc_dev, acc_or), "accept mask differs at step %d (%d walkers)" % (step, int(np.sum(acc_dev != acc_or)))
        if mv.kind == "snooker":
            np.testing.assert_allclose(x1, xo, rtol=1e-9, atol=1e-10)
        else:
            assert np.array_equal(x1, xo), "coordinates differ at step %d" % step
        assert_lp_close(lp1, lpo, 1e-9 if mv.kind == "snooker" else 1e-11)
        nacc += int(acc_dev.sum())
    assert nacc > 0
    if len(spec["moves"]) > 1:
        assert used == {m.kind for m in spec["moves"]}, "the seed must exercise every move of the mixture: %s" % used
    ens.close()


@pytest.mark.parametrize("name,world", [("c2_65536x64_dense_stretch", 4), ("c4_65536x64_dense_de_snooker", 2)])
def test_pull_exchange_at_full_size(name, world):
    """Pull exchange at a BASELINE size with `world` logical ranks on the one GPU: the per-pair record
    capacity (mean + 8 sigma) must hold, every rank's block and the re-synchronised replicas must equal
    the single-rank run bit for bit."""
    import torch
    from emcee_amd.parallel import DeviceEngine, LocalGroup, block_range, pull_capacity
    spec = FULL[name]()
    nst = 3

    def setup(ens):
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(31337, 0)
        ens.chain_config(nst)

    ref = make_ens(spec, spec["p0"])
    setup(ref)
    ref.run(nst, 1, True)
    ref_chain, ref_lp, ref_acc = ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst), ref.accepted_counts()
    ref.close()
    engines = []
    for r in range(world):
        ens = make_ens(spec, spec["p0"])
        setup(ens)
        engines.append(DeviceEngine(ens, r, world, torch.device("cuda", 0), exchange="pull"))
    nd = spec["D"]

    def sync():
        for e in engines:
            e.ens.sync()
        torch.cuda.synchronize()

    for _ in range(nst):
        res = [e.step_begin(True) for e in engines]
        assert all(r == res[0] for r in res)
        npart = {"stretch": 1, "de": 2, "snooker": 3}[spec["moves"][res[0][0]].kind]
        for split in range(res[0][1]):
            caps = [e.pull_prepare(split) for e in engines]
            # one capacity per context: the largest any installed move needs (records keep their address between half-steps)
            assert set(caps) == {max(pull_capacity(spec["N"], world, m.nsplits, {"stretch": 1, "de": 2, "snooker": 3}[m.kind])
                                     for m in spec["moves"])}
            assert caps[0] >= pull_capacity(spec["N"], world, res[0][1], npart)
            sync()
            LocalGroup._all_to_all(engines, caps[0] * (nd + 1))
            sync()
            for e in engines:
                e.pull_apply(split)
        for e in engines:
            e.step_end()
    for r, e in enumerate(engines):
        lo, hi = block_range(spec["N"], r, world)
        assert e.ens.status() == 0, "exchange capacity exceeded"
        assert np.array_equal(e.ens.chain_read(0, 0, nst)[:, lo:hi], ref_chain[:, lo:hi])
        assert np.array_equal(e.ens.chain_read(1, 0, nst)[:, lo:hi], ref_lp[:, lo:hi])
    per = [e.replica_pack() for e in engines]
    sync()
    LocalGroup._all_gather_n(engines, per[0] * (nd + 3))
    sync()
    for e in engines:
        e.replica_unpack()
        x, lp = e.ens.get_state()
        assert np.array_equal(x, ref_chain[-1]) and np.array_equal(lp, ref_lp[-1])
        assert np.array_equal(e.ens.accepted_counts(), ref_acc)
        e.ens.close()


@pytest.mark.parametrize("name,world", [("c2_65536x64_dense_stretch", 4), ("c3_262144x32_rosenbrock_stretch", 2)])
def test_direct_exchange_at_full_size(name, world):
    """Direct exchange at a BASELINE size with `world` logical ranks on the one GPU (host-ordered half-steps): the shapes
    whose sharded runs take the LEAN = 2 instantiation of the half-step kernel.  Every rank's block of the chain must equal
    the single-rank run bit for bit."""
    import torch
    from emcee_amd.parallel import DeviceEngine, attach_direct_peers, block_range
    spec = FULL[name]()
    nst = 3

    def setup(ens):
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(271828, 0)
        ens.chain_config(nst)

    ref = make_ens(spec, spec["p0"])
    setup(ref)
    ref.run(nst, 1, True)
    ref_chain, ref_lp = ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst)
    ref.close()
    engines = []
    for r in range(world):
        ens = make_ens(spec, spec["p0"])
        setup(ens)
        engines.append(DeviceEngine(ens, r, world, torch.device("cuda", 0), exchange="direct"))
    attach_direct_peers([e.ens for e in engines])
    for _ in range(nst):
        res = [e.step_begin(True) for e in engines]
        assert all(r == res[0] for r in res)
        for split in range(res[0][1]):
            for e in engines:
                e.ens.direct_halfstep(split, barrier=False)
            for e in engines:
                e.ens.sync()
        for e in engines:
            e.step_end()
    for r, e in enumerate(engines):
        lo, hi = block_range(spec["N"], r, world)
        assert e.ens.status() == 0
        assert np.array_equal(e.ens.chain_read(0, 0, nst)[:, lo:hi], ref_chain[:, lo:hi])
        assert np.array_equal(e.ens.chain_read(1, 0, nst)[:, lo:hi], ref_lp[:, lo:hi])
        e.ens.close()
