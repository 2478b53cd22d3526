"""Pin the NumPy oracle against outputs of reference emcee (tests/golden).

The fixtures were produced by oracle/gen_golden.py from the live reference.
Coordinates, accept counts, split labels, integer draws and the final
MT19937 state must be bit-identical; log-probs that go through BLAS (dense
target) are compared to 1e-12 relative because BLAS kernels are CPU-specific.
"""
import numpy as np
import pytest

from oracle import cases, ref_shim
from oracle import sampler_oracle as so

from helpers import digest, load_digests, load_golden, rng_for_case, rng_from_fixture, run_oracle


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_reproduces_reference_fixture(name):
    g = load_golden(name)
    spec = cases.build(name)
    rs = rng_from_fixture(g)
    trace = []
    out = run_oracle(spec, g["p0"], rs, trace=trace)
    assert np.array_equal(out["chain"], g["chain"]), "coords differ from reference"
    assert np.array_equal(out["accepted_count"], g["accepted_count"])
    if spec["desc"]["kind"] == "dense":
        np.testing.assert_allclose(out["log_prob"], g["log_prob"], rtol=1e-12, atol=0)
    else:
        assert np.array_equal(out["log_prob"], g["log_prob"])
    st = rs.get_state()
    assert np.array_equal(st[1], g["rng_key1"]) and st[2] == int(g["rng_pos1"])
    assert st[3] == int(g["rng_has_gauss1"]) and st[4] == float(g["rng_cached1"])
    # split labels / integer draws as the reference drew them
    if "labels" in g:
        labels = np.stack([step[0]["labels"] for step in trace])
        assert np.array_equal(labels, g["labels"])
    if "ints" in g:
        ints = []
        for step in trace:
            for tr in step:
                if "rint" in tr:
                    ints.append(tr["rint"])
                if "pair_index" in tr:
                    ints.append(tr["pair_index"])
                if "picks" in tr:
                    ints.append(tr["picks"].reshape(-1))
        assert np.array_equal(np.concatenate(ints), g["ints"])


@pytest.mark.parametrize("name", list(cases.HOST_MOVE_CASES))
def test_oracle_reproduces_host_move_fixture(name):
    """MHMove/GaussianMove, WalkMove, KDEMove: same pinning as the hot-path moves."""
    g = load_golden(name)
    spec = cases.build(name)
    rs = rng_from_fixture(g)
    out = run_oracle(spec, g["p0"], rs)
    assert np.array_equal(out["chain"], g["chain"]), "coords differ from reference"
    assert np.array_equal(out["accepted_count"], g["accepted_count"])
    np.testing.assert_allclose(out["log_prob"], g["log_prob"], rtol=1e-12, atol=0)
    st = rs.get_state()
    assert np.array_equal(st[1], g["rng_key1"]) and st[2] == int(g["rng_pos1"])
    assert st[3] == int(g["rng_has_gauss1"]) and st[4] == float(g["rng_cached1"])


@pytest.mark.parametrize("name", list(cases.DIGEST_CASES))
def test_oracle_digest_cases(name):
    d = load_digests()[name]
    spec = cases.build(name)
    # the start state of every digest case is element-wise arithmetic on RandomState draws (oracle/cases.py: no BLAS), hence the
    # same bits on every CPU: the reference's digest is ALWAYS comparable (no skip -- round-5 verdict)
    assert digest(spec["p0"]) == d["p0"], "the start state of a digest case must not depend on the CPU"
    out = run_oracle(spec, spec["p0"], rng_for_case(spec))
    assert digest(out["chain"]) == d["chain"]
    assert digest(out["accepted_count"]) == d["accepted_count"]
    assert float(out["accepted_count"].sum()) == d["accepted_total"]


@pytest.mark.parametrize("n", [2, 3, 5, 16, 33, 100])
def test_de_pair_closed_form_matches_table(n):
    """moves/de.py:67-77 table vs the closed form used by oracle and device."""
    tab = so.de_pair_table(n)
    k = np.arange(n * (n - 1))
    first, second = so.de_pair_decode(k, n)
    assert np.array_equal(first, tab[:, 0]) and np.array_equal(second, tab[:, 1])


def test_de_pair_closed_form_large_boundaries():
    n = 32768
    T = n * (n - 1) // 2
    i = np.array([1, 2, 3, 1000, 32767], dtype=np.int64)
    tri = i * (i - 1) // 2
    for k, ii in zip(np.concatenate([tri, tri + i - 1]), np.concatenate([i, i])):
        f, s = so.de_pair_decode(np.array([k]), n)
        assert f[0] == ii and s[0] == k - ii * (ii - 1) // 2
        f2, s2 = so.de_pair_decode(np.array([k + T]), n)
        assert (f2[0], s2[0]) == (s[0], f[0])


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_autocorr_matches_reference():
    emcee = ref_shim.import_reference()
    rs = np.random.RandomState(3)
    x = np.empty((2000, 6, 2))
    x[0] = rs.randn(6, 2)
    for t in range(1, 2000):
        x[t] = 0.9 * x[t - 1] + rs.randn(6, 2)
    ref = emcee.autocorr.integrated_time(x, quiet=True)
    np.testing.assert_allclose(so.integrated_time(x), ref, rtol=1e-12)
