"""The product's C++ MT19937/legacy-RandomState re-implementation vs NumPy itself, and the
exact-mode step plan vs the draws the (reference-pinned) oracle makes.  CPU only."""
import numpy as np
import pytest

from oracle import cases

from emx_testlib import HostMT, cdf_of, move_desc, plan_from_trace
from helpers import load_golden, rng_from_fixture, run_oracle


def _pair(seed):
    rs = np.random.RandomState(seed)
    return rs, HostMT(rs.get_state())


def _same_state(a, b):
    assert np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and a[4] == b[4]


def test_random_sample_and_state():
    rs, mt = _pair(3)
    assert np.array_equal(rs.random_sample(5000), mt.random_sample(5000))
    _same_state(rs.get_state(), mt.get_state())


@pytest.mark.parametrize("bound", [1, 2, 3, 16, 25, 1000, 32768, 65537, 2**31, 2**32 - 1, 2**32, 2**32 + 1, 2**40 + 12345, 1056964608])
def test_randint_masked_rejection(bound):
    rs, mt = _pair(bound % 1000)
    assert np.array_equal(rs.randint(bound, size=3000), mt.randint(bound, 3000))
    _same_state(rs.get_state(), mt.get_state())


def test_randn_polar_with_cache():
    rs, mt = _pair(9)
    a = np.concatenate([rs.randn(7), rs.randn(1), rs.randn(4, 1)[:, 0]])
    b = np.concatenate([mt.randn(7), mt.randn(1), mt.randn(4)])
    assert np.array_equal(a, b)
    _same_state(rs.get_state(), mt.get_state())


@pytest.mark.parametrize("n,S", [(2, 2), (50, 2), (64, 4), (65536, 2), (1001, 5)])
def test_shuffle_labels(n, S):
    rs, mt = _pair(n)
    inds = np.arange(n) % S
    rs.shuffle(inds)
    assert np.array_equal(inds, mt.shuffle_labels(n, S))
    _same_state(rs.get_state(), mt.get_state())


def test_choice_cdf():
    rs, mt = _pair(17)
    for w in ([1.0], [0.8, 0.2], [1, 1, 1], [0.1, 0.5, 0.2, 0.2]):
        cdf = cdf_of(w, len(w))
        p = np.asarray(w, dtype=float) / np.sum(w)
        for _ in range(200):
            assert rs.choice(len(w), p=p) == mt.choice_cdf(cdf)
    _same_state(rs.get_state(), mt.get_state())


@pytest.mark.parametrize("name", list(cases.CASES))
def test_exact_plan_matches_oracle_draws(name):
    """emx_host_plan_mt (the producer emx_run uses in MT19937 mode) reproduces every draw the
    oracle -- and therefore reference emcee -- makes for each step of each golden case."""
    g = load_golden(name)
    spec = cases.build(name)
    trace = []
    rs = rng_from_fixture(g)
    out = run_oracle(spec, g["p0"], rs, trace=trace)
    mt = HostMT(rng_from_fixture(g).get_state())
    N, D = spec["N"], spec["D"]
    cdf = cdf_of(spec["weights"], len(spec["moves"]))
    for it, step_trace in enumerate(trace):
        k = mt.choice_cdf(cdf)
        assert k == out["move_choices"][it]
        mv = spec["moves"][k]
        got = mt.plan(N, D, move_desc(mv, D))
        exp = plan_from_trace(step_trace, mv, D)
        for key in ("off", "order", "p0", "uacc"):
            assert np.array_equal(got[key], exp[key]), (name, it, key)
        if mv.kind in ("stretch", "de"):
            assert np.array_equal(got["s0"], exp["s0"]), (name, it, "s0")
        if mv.kind in ("de", "snooker"):
            assert np.array_equal(got["p1"], exp["p1"])
        if mv.kind == "snooker":
            assert np.array_equal(got["p2"], exp["p2"])
    _same_state(mt.get_state(), rs.get_state())
