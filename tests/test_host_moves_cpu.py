"""MHMove / GaussianMove through the product's EnsembleSampler with an ordinary Python log_prob_fn.

These moves propose from the walker's own position: no complement, no red/blue kernel, and with a
host callable nothing touches the GPU, so the whole path is checked here against the
reference-generated fixtures bit for bit (chain, log-prob, accept counts, final MT19937 state)."""
import numpy as np
import pytest

import emcee_amd
from emcee_amd import moves
from oracle import cases

from helpers import load_golden, rng_from_fixture

GAUSS = [n for n, c in cases.HOST_MOVE_CASES.items() if all(m.kind == "gaussian" for m in c["moves"])]


def product_move(m):
    return moves.GaussianMove(m.cov, mode=m.mode, factor=m.factor)


@pytest.mark.parametrize("name", GAUSS)
def test_gaussian_move_same_seed_same_chain_as_reference(name):
    g = load_golden(name)
    spec = cases.build(name)
    fn = cases.make_target(spec["desc"])
    s = emcee_amd.EnsembleSampler(spec["N"], spec["D"], fn, moves=[product_move(m) for m in spec["moves"]], vectorize=True)
    s._random.set_state(rng_from_fixture(g).get_state())
    last = s.run_mcmc(g["p0"], spec["nsteps"], skip_initial_state_check=True)
    assert np.array_equal(s.get_chain(), g["chain"])
    np.testing.assert_allclose(s.get_log_prob(), g["log_prob"], rtol=1e-12)
    assert np.array_equal(s.backend.accepted, g["accepted_count"])
    st = s.random_state
    assert np.array_equal(st[1], g["rng_key1"]) and st[2] == int(g["rng_pos1"])
    assert st[3] == int(g["rng_has_gauss1"]) and st[4] == float(g["rng_cached1"])
    assert np.array_equal(last.coords, g["chain"][-1])


def test_gaussian_move_argument_errors():
    """reference gaussian.py:41-52,69-79 / unit tests of the constructor"""
    with pytest.raises(ValueError, match="Invalid proposal scale dimensions"):
        moves.GaussianMove(np.ones((2, 3)))
    with pytest.raises(ValueError, match="not a recognized mode"):
        moves.GaussianMove(1.0, mode="nope")
    with pytest.raises(ValueError, match="not a recognized mode"):
        moves.GaussianMove(np.eye(2), mode="random")          # matrix proposals are vector-only
    with pytest.raises(ValueError, match="must be >= 1.0"):
        moves.GaussianMove(1.0, factor=0.5)
    mv = moves.GaussianMove([1.0, 2.0])
    assert mv.ndim == 2 and moves.GaussianMove(3.0).ndim is None and moves.GaussianMove(np.eye(3)).ndim == 3


def test_mh_move_dimension_check_and_custom_proposal():
    def shift(coords, rng):
        return coords + 0.1 * rng.randn(*coords.shape), np.zeros(len(coords))

    lp = lambda x: -0.5 * np.sum(x ** 2, axis=1)  # noqa: E731
    s = emcee_amd.EnsembleSampler(10, 2, lp, moves=moves.MHMove(shift, ndim=3), vectorize=True)
    with pytest.raises(ValueError, match="Dimension mismatch in proposal"):
        s.run_mcmc(np.random.RandomState(1).randn(10, 2), 2)
    s = emcee_amd.EnsembleSampler(10, 2, lp, moves=moves.MHMove(shift, ndim=2), vectorize=True)
    s.run_mcmc(np.random.RandomState(1).randn(10, 2), 20)
    assert s.get_chain().shape == (20, 10, 2) and 0.0 < np.mean(s.acceptance_fraction) <= 1.0


def test_sequential_mode_cycles_coordinates():
    mv = moves.GaussianMove(1.0, mode="sequential")
    rs = np.random.RandomState(3)
    x0 = np.zeros((5, 3))
    for step in range(7):
        q, f = mv.get_proposal(x0, rs)
        moved = np.nonzero(np.any(q != x0, axis=0))[0]
        assert list(moved) == [step % 3] and np.all(f == 0)
