"""k_persist_slab (csrc/emx_pslab.hip, round 6): the slab form of the fused dense-Gaussian half-step -- padded ndim 80 ... 128, the
stretch and DE moves -- as a persistent kernel, device-wide and one-XCD.  It must be the path emx_run takes for these shapes and give
the bits of the launch-per-half-step kernels (k_halfstep_slab / k_halfstep, held equal to the oracle by tests/test_gpu_wide_dense.py,
test_gpu_parity.py and test_gpu_full_size.py): coordinates, log-probs, accept marks, chain rows, accept counters -- in both RNG modes,
and the final MT19937 state in exact mode."""
import numpy as np
import pytest

from emcee_amd import _lib
from oracle import sampler_oracle as so

from test_gpu_full_size import full_spec
from test_gpu_parity import make_ens
from test_gpu_persist import SEED, native_ens

pytestmark = pytest.mark.gpu

S = so.MoveSpec


def _run(spec, persist, calls, nsteps, thin_by, store, local=1, tuning=None):
    ens = native_ens(spec, persist)
    ens.set_tuning("small_kernel", 0)          # (an ensemble that fits one workgroup's LDS -- 1 024 x 9 -- would take k_small_run; and padded
                                               # ndim 16 with a stored chain keeps the launches by rule: run_impl, profiles/r03/persist_dims.txt)
    ens.set_tuning("persist_local", local)
    for k, v in (tuning or {}).items():
        ens.set_tuning(k, v)
    if store:
        ens.chain_config(nsteps * calls)
    for _ in range(calls):
        ens.run(nsteps, thin_by, store)
    assert ens.status() == 0
    x, lp = ens.get_state()
    rec = dict(x=x, lp=lp, acc=ens.accepted_mask(), info=ens.persist_info())
    if store:
        rec.update(chain=ens.chain_read(0, 0, nsteps * calls), chain_lp=ens.chain_read(1, 0, nsteps * calls), counts=ens.accepted_counts())
    ens.close()
    return rec


def _same(p, c):
    for key in c:
        if key != "info":
            assert np.array_equal(p[key], c[key]), key


@pytest.mark.parametrize("N,D,move,store,thin_by", [
    (65536, 128, "stretch", False, 1), (65536, 128, "de", False, 1), (65536, 112, "stretch", True, 1), (65536, 100, "de", True, 2),
    (8192, 128, "stretch", True, 1), (8192, 96, "de", False, 1),
    (32768, 128, "stretch", True, 1), (32768, 96, "de", False, 1), (16384, 128, "stretch", False, 1), (16384, 80, "stretch", True, 3),
    (49152, 66, "stretch", False, 1), (24576, 126, "de", False, 1),
])
def test_device_wide_form_gives_the_bits_of_the_per_half_step_path(N, D, move, store, thin_by):
    """two calls of 21 steps (full and partial launches: 16 + 5), every workgroup shape of the persistent grid (8, 4, 2 waves), every
    padded ndim the kernel is instantiated for (80, 96, 112, 128), even ndim that is not a multiple of 16"""
    spec = full_spec(N, D, "dense", [S(move)], seed=7)
    p = _run(spec, 1, 2, 21, thin_by, store, local=0, tuning={"persist_slab": 2})      # (2: also at 65 536 x 112 / 128, where the launches are level)
    c = _run(spec, 0, 2, 21, thin_by, store)
    assert p["info"]["qualifies"] and p["info"]["launches"] >= 4 and p["info"]["halfsteps"] == 84 * thin_by and p["info"]["local_launches"] == 0
    assert c["info"]["launches"] == 0 and p["acc"].any()
    _same(p, c)


@pytest.mark.parametrize("N,D,move,store,thin_by", [
    (4096, 128, "stretch", False, 1), (4096, 112, "de", True, 1), (4096, 128, "stretch", True, 2), (2048, 96, "de", False, 1),
    (1024, 128, "stretch", True, 1), (512, 80, "stretch", False, 1), (3072, 100, "stretch", False, 1), (1024, 66, "de", True, 1),
])
def test_one_xcd_form_gives_the_bits_of_the_per_half_step_path(N, D, move, store, thin_by):
    """ensembles of up to 4 096 walkers (tuning persist_slab_local_max_walkers): every working group on one XCD (plain stores, sc1
    loads, the flag barrier)"""
    spec = full_spec(N, D, "dense", [S(move)], seed=8)
    p = _run(spec, 1, 2, 21, thin_by, store)
    c = _run(spec, 0, 2, 21, thin_by, store)
    assert p["info"]["qualifies"] and p["info"]["local_launches"] == p["info"]["launches"] >= 4 and p["info"]["recovered"] == 0
    assert c["info"]["launches"] == 0 and p["acc"].any()
    _same(p, c)


def test_mixture_of_stretch_and_de_and_what_does_not_qualify():
    """a stretch + DE schedule: runs of each move in launches of their own; a schedule with the snooker move (four rows a walker: no
    slab form), the DE move at an odd ndim and ndim above 128 keep the per-half-step launches -- and all of them agree with the control"""
    spec = full_spec(16384, 128, "dense", [S("stretch"), S("de")], weights=[0.6, 0.4], seed=9)
    p, c = _run(spec, 1, 1, 40, 1, True), _run(spec, 0, 1, 40, 1, True)
    assert p["info"]["launches"] > 0 and c["info"]["launches"] == 0
    _same(p, c)
    for moves, D in (([S("snooker")], 128), ([S("de")], 127), ([S("stretch")], 129)):
        spec = full_spec(4096, D, "dense", moves, seed=10)
        ens = native_ens(spec, 1)
        ens.run(5, 1, False)
        assert ens.persist_info()["launches"] == 0 and ens.status() == 0
        ens.close()
    # tuning persist_slab = 0: the per-half-step launches; and by default 65 536 x 128 keeps them (level there: persist_slab_ok)
    spec = full_spec(4096, 128, "dense", [S("stretch")], seed=10)
    ens = native_ens(spec, 1)
    ens.set_tuning("persist_slab", 0)
    ens.run(5, 1, False)
    assert ens.persist_info()["launches"] == 0
    ens.close()
    spec = full_spec(65536, 128, "dense", [S("stretch")], seed=10)
    ens = native_ens(spec, 1)
    ens.run(5, 1, False)
    assert ens.persist_info()["launches"] == 0 and not ens.persist_info()["qualifies"]
    ens.close()


@pytest.mark.parametrize("N,D,move", [(4096, 128, "stretch"), (16384, 112, "stretch"), (2048, 96, "de"), (32768, 128, "stretch")])
def test_exact_mode_takes_the_persistent_slab_kernel(N, D, move):
    """rng = MT19937 (the Python default): the host pipeline's plans, sixteen steps a launch -- coordinates, log-probs, accept counters
    and the final generator state equal the per-half-step exact path's (tuning persist_exact = 0) over three calls of 21 steps"""
    spec = full_spec(N, D, "dense", [S(move)], seed=11)
    state = np.random.RandomState(4242 + N).get_state()
    recs = []
    for pe in (1, 0):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(state)
        ens.set_tuning("persist_exact", pe)
        ens.set_tuning("persist_timeout_ms", 200)
        ens.chain_config(63)
        for _ in range(3):
            ens.run(21, 1, True)
        assert ens.status() == 0
        x, lp = ens.get_state()
        recs.append(dict(x=x, lp=lp, chain=ens.chain_read(0, 0, 63), counts=ens.accepted_counts(), rng=ens.get_mt19937(), info=ens.persist_info()))
        ens.close()
    p, c = recs
    assert p["info"]["launches"] > 0 and c["info"]["launches"] == 0
    assert np.array_equal(p["rng"][1], c["rng"][1]) and p["rng"][2] == c["rng"][2]
    for key in ("x", "lp", "chain", "counts"):
        assert np.array_equal(p[key], c[key]), key


@pytest.mark.parametrize("N,trials", [(65536, 40), (8192, 60), (4096, 60), (1024, 60)])
def test_coherence_stress(N, trials):
    """many short runs, fresh Philox seed each, every trial bit-compared with the per-half-step path started from the same state
    (the class of bug that shows once in a hundred runs: profiles/r03/persist_coherence.txt)"""
    spec = full_spec(N, 128, "dense", [S("stretch")], seed=12)
    ens = [native_ens(spec, persist) for persist in (1, 0)]
    ens[0].set_tuning("persist_slab", 2)
    for t in range(trials):
        got = []
        for e in ens:
            e.set_philox(0xC0DE00 + 7919 * t + N, 0)
            e.run(40, 1, False)
            x, lp = e.get_state()
            assert e.status() == 0
            got.append([x, lp, e.accepted_mask()])
        for k, (a, b) in enumerate(zip(*got)):
            assert np.array_equal(a, b), "trial %d of %d: output %d differs" % (t, trials, k)
    p, c = ens[0].persist_info(), ens[1].persist_info()
    assert p["halfsteps"] == 80 * trials and p["recovered"] == 0 and c["launches"] == 0
    for e in ens:
        e.close()


@pytest.mark.parametrize("N,D,move,store,thin_by,local", [
    (65536, 63, "stretch", False, 1, 0), (65536, 33, "de", True, 1, 0), (32768, 47, "snooker", False, 1, 0), (16384, 17, "stretch", True, 2, 0),
    (16384, 5, "stretch", False, 1, 0), (8192, 63, "stretch", True, 1, 1), (4096, 33, "snooker", False, 1, 1), (2048, 49, "de", True, 1, 1),
    (1024, 31, "stretch", False, 1, 1), (1024, 9, "de", False, 1, 1), (4096, 33, "stretch", False, 1, 0),
])
def test_odd_ndim_runs_persistently(N, D, move, store, thin_by, local):
    """csrc/emx_podd.hip (round 6): k_persist in the row layouts of an odd ndim -- one coordinate per lane and chunk (rows of an odd
    number of doubles are 8-byte aligned only), rows of 8 lanes up to padded ndim 32 and of 16 lanes at 48 and 64 -- for the stretch,
    DE and snooker moves, device-wide and one-XCD.  Two calls of 21 steps against the per-half-step launches: same bits."""
    spec = full_spec(N, D, "dense", [S(move)], seed=13)
    p = _run(spec, 1, 2, 21, thin_by, store, local=local)
    c = _run(spec, 0, 2, 21, thin_by, store)
    S_ = 4 if move == "snooker" else 2
    assert p["info"]["qualifies"] and p["info"]["launches"] >= 4 and p["info"]["halfsteps"] == 42 * S_ * thin_by
    assert (p["info"]["local_launches"] == p["info"]["launches"]) == bool(local) and p["info"]["recovered"] == 0
    assert c["info"]["launches"] == 0 and p["acc"].any()
    _same(p, c)
    # tuning persist_odd = 0: the per-half-step launches
    if N == 1024:
        ens = native_ens(spec, 1)
        ens.set_tuning("persist_odd", 0)
        ens.run(5, 1, False)
        assert ens.persist_info()["launches"] == 0
        ens.close()


def test_odd_ndim_in_exact_mode_and_in_a_mixture():
    """odd ndim with the reference's own stream (the host pipeline's plans fetched sixteen steps a launch) and a stretch + DE mixture"""
    spec = full_spec(4096, 33, "dense", [S("stretch")], seed=14)
    state = np.random.RandomState(99).get_state()
    recs = []
    for pe in (1, 0):
        ens = make_ens(spec, spec["p0"])
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(state)
        ens.set_tuning("persist_exact", pe)
        ens.chain_config(42)
        for _ in range(2):
            ens.run(21, 1, True)
        assert ens.status() == 0
        x, lp = ens.get_state()
        recs.append(dict(x=x, lp=lp, chain=ens.chain_read(0, 0, 42), counts=ens.accepted_counts(), rng=ens.get_mt19937(), info=ens.persist_info()))
        ens.close()
    p, c = recs
    assert p["info"]["launches"] > 0 and c["info"]["launches"] == 0
    assert np.array_equal(p["rng"][1], c["rng"][1]) and p["rng"][2] == c["rng"][2]
    for key in ("x", "lp", "chain", "counts"):
        assert np.array_equal(p[key], c[key]), key
    spec = full_spec(8192, 61, "dense", [S("stretch"), S("de"), S("snooker")], weights=[0.5, 0.3, 0.2], seed=15)
    p, c = _run(spec, 1, 1, 48, 1, True), _run(spec, 0, 1, 48, 1, True)
    assert p["info"]["launches"] > 0 and c["info"]["launches"] == 0
    _same(p, c)


@pytest.mark.parametrize("N,D,store,thin_by,local", [(32768, 127, False, 1, 0), (16384, 65, True, 1, 0), (8192, 97, False, 1, 0), (4096, 127, True, 2, 1),
                                                     (1024, 65, False, 1, 1), (2048, 111, True, 1, 1), (49152, 81, False, 1, 0)])
def test_odd_ndim_above_64_takes_the_slab_kernel(N, D, store, thin_by, local):
    """odd ndim 65 ... 127, stretch move: k_persist_slab in its 8-byte-granular form (the 16-lane register layout, rows moved 8 bytes at
    a time) against the launch-per-half-step kernels, which run such an ndim in rows of 32 lanes -- the same bits; the DE move at an
    odd ndim keeps the launches"""
    spec = full_spec(N, D, "dense", [S("stretch")], seed=21)
    p = _run(spec, 1, 2, 21, thin_by, store, local=local, tuning={"persist_slab": 2})
    c = _run(spec, 0, 2, 21, thin_by, store)
    assert p["info"]["qualifies"] and p["info"]["launches"] >= 4 and p["info"]["halfsteps"] == 84 * thin_by
    assert (p["info"]["local_launches"] == p["info"]["launches"]) == bool(local) and c["info"]["launches"] == 0 and p["acc"].any()
    _same(p, c)
    if N == 1024:
        spec = full_spec(N, D, "dense", [S("de")], seed=21)
        ens = native_ens(spec, 1)
        ens.run(5, 1, False)
        assert ens.persist_info()["launches"] == 0 and ens.status() == 0
        ens.close()
