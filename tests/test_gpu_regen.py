"""Exact (MT19937) mode, large ensembles: the fixed-length draws of a stretch step made again ON THE DEVICE from the generator's state
(k_plan_regen, round 6; csrc/emx_mtpipe.hpp PipeStepInfo::regen) instead of crossing from the generator's core to the tokenizer's,
into the staging buffer and over PCIe.  The chain must not notice: same coordinates, log-probs, accept counters and final generator
state as with the words copied (tuning mt_regen_min_walkers = 0) -- and as the reference-pinned oracle (tests/test_gpu_full_size.py
runs BASELINE's C2 through this path for 24 steps)."""
import numpy as np
import pytest

from emcee_amd import _lib
from oracle import cases
from oracle import sampler_oracle as so

from test_gpu_full_size import full_spec
from test_gpu_parity import make_ens

pytestmark = pytest.mark.gpu

S = so.MoveSpec


def _run(spec, state, regen_min, calls, nsteps, tuning=None):
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(state)
    ens.set_tuning("mt_regen_min_walkers", regen_min)
    for k, v in (tuning or {}).items():
        ens.set_tuning(k, v)
    ens.chain_config(calls * nsteps)
    for _ in range(calls):
        ens.run(nsteps, 1, True)
    assert ens.status() == 0
    x, lp = ens.get_state()
    rec = dict(x=x, lp=lp, chain=ens.chain_read(0, 0, calls * nsteps), counts=ens.accepted_counts(), rng=ens.get_mt19937(),
               hand=ens.pipeline_handovers(), info=ens.persist_info())
    ens.close()
    return rec


@pytest.mark.parametrize("N,D,target,pos", [(65536, 64, "dense", None), (65536, 8, "iso", 623), (131072, 4, "iso", 1), (16384, 130, "dense", 300),
                                            (32768, 32, "rosenbrock", 77), (262144 // 2, 6, "diag", None)])
def test_regen_steps_give_the_chain_of_copied_words(N, D, target, pos):
    spec = full_spec(N, D, target, [S("stretch")], seed=31, p0="rosen" if target == "rosenbrock" else "randn")
    st = list(np.random.RandomState(5 + N).get_state())
    if pos is not None:
        st[2] = pos
    st = tuple(st)
    # (persist_exact = 0, mt_device = 0: the host pipeline with an upload per step, whatever the size -- the path regen lives on)
    fixed = {"persist_exact": 0, "mt_device": 0}
    a = _run(spec, st, 16384, 3, 7, fixed)
    b = _run(spec, st, 0, 3, 7, fixed)
    assert a["hand"]["regen_steps"] == 21 and a["hand"]["raw_steps"] == 21
    assert b["hand"]["regen_steps"] == 0 and b["hand"]["raw_steps"] == 21
    assert np.array_equal(a["rng"][1], b["rng"][1]) and a["rng"][2] == b["rng"][2]
    for key in ("x", "lp", "chain", "counts"):
        assert np.array_equal(a[key], b[key]), key
    assert a["counts"].sum() > 0


def test_regen_against_the_oracle_and_in_a_mixture():
    """32 768 x 16 against the oracle (which draws from numpy.random.RandomState itself) over 12 steps; a stretch + DE mixture: regen for
    its stretch steps, the DE steps as ever"""
    spec = full_spec(32768, 16, "dense", [S("stretch")], seed=32)
    fn = cases.make_target(spec["desc"])
    rs = np.random.RandomState(spec["rng_seed"])
    out = so.run(spec["p0"], 12, fn, rs, moves=spec["moves"], weights=spec["weights"])
    a = _run(spec, np.random.RandomState(spec["rng_seed"]).get_state(), 16384, 1, 12, {"persist_exact": 0})
    assert a["hand"]["regen_steps"] == 12
    assert np.array_equal(a["chain"], out["chain"]) and np.array_equal(a["counts"], out["accepted_count"])
    b = rs.get_state()
    assert np.array_equal(a["rng"][1], b[1]) and a["rng"][2] == b[2]
    spec = full_spec(16384, 8, "iso", [S("stretch"), S("de")], weights=[0.5, 0.5], seed=33)
    st = np.random.RandomState(9).get_state()
    a, b = _run(spec, st, 16384, 2, 15, {"persist_exact": 0}), _run(spec, st, 0, 2, 15, {"persist_exact": 0})
    assert 0 < a["hand"]["regen_steps"] < 30 and b["hand"]["regen_steps"] == 0
    assert np.array_equal(a["rng"][1], b["rng"][1]) and a["rng"][2] == b["rng"][2]
    for key in ("x", "lp", "chain", "counts"):
        assert np.array_equal(a[key], b[key]), key


@pytest.mark.parametrize("N,D,target", [(65536, 64, "dense"), (32768, 64, "dense"), (16384, 32, "dense"), (65536, 16, "iso"), (32768, 33, "dense"), (16384, 128, "dense")])
def test_persistent_launches_take_regen_steps(N, D, target):
    """Round 6, second half: the persistent launches' fetch (k_plan_fetch) takes regen steps as they are -- `order` and the generator
    states -- and the batched k_plan_regen / k_plan_raw behind it finish all steps of a launch at once.  A launch's plans are 6.8 MB at
    65 536 walkers where finished plans were 25 MB, so BASELINE's C2 runs the persistent kernel in exact mode now (persist_exact_max_walkers
    = 32 768 stays the bound for plans that travel finished).  Three ways to the same chain: persistent + regen, per-step uploads + regen,
    per-step uploads with the words copied."""
    spec = full_spec(N, D, target, [S("stretch")], seed=41)
    st = np.random.RandomState(77 + N).get_state()
    a = _run(spec, st, 16384, 3, 23, {"persist_timeout_ms": 400})
    b = _run(spec, st, 16384, 3, 23, {"persist_exact": 0})
    c = _run(spec, st, 0, 3, 23, {"persist_exact": 0})
    assert a["info"]["launches"] >= 6 and a["hand"]["regen_steps"] == 69 and a["info"]["recovered"] == 0
    assert b["info"]["launches"] == 0 and b["hand"]["regen_steps"] == 69
    assert c["info"]["launches"] == 0 and c["hand"]["regen_steps"] == 0
    for other in (b, c):
        assert np.array_equal(a["rng"][1], other["rng"][1]) and a["rng"][2] == other["rng"][2]
        for key in ("x", "lp", "chain", "counts"):
            assert np.array_equal(a[key], other[key]), key


def test_long_exact_run_on_the_persistent_kernel_at_the_headline_size():
    """BASELINE configs[1] in the Python default RNG mode: 600 steps, the host enqueueing ahead of the device, every plan slot fetched
    and finished on the device some twenty times -- final state, accept counters and generator state equal the per-step path's"""
    spec = full_spec(65536, 64, "dense", [S("stretch")], seed=42)
    st = np.random.RandomState(4).get_state()
    recs = []
    for pe in (1, 0):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(st)
        ens.set_tuning("persist_exact", pe)
        ens.run(600, 1, False)
        assert ens.status() == 0
        x, lp = ens.get_state()
        recs.append(dict(x=x, lp=lp, counts=ens.accepted_counts(), rng=ens.get_mt19937(), info=ens.persist_info(), hand=ens.pipeline_handovers()))
        ens.close()
    p, c = recs
    assert p["info"]["launches"] >= 600 // 16 and c["info"]["launches"] == 0 and p["hand"]["regen_steps"] == 600
    assert np.array_equal(p["x"], c["x"]) and np.array_equal(p["lp"], c["lp"]) and np.array_equal(p["counts"], c["counts"])
    assert np.array_equal(p["rng"][1], c["rng"][1]) and p["rng"][2] == c["rng"][2]
