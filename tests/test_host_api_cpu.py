"""Host-side pieces of the drop-in surface that need no GPU: State (reference unit/test_state.py:14-60),
Backend bookkeeping and errors (unit/test_backends.py:78-101,187-216), the sampler with a Python log_prob_fn and a
whole-ensemble (non split-ensemble) move -- a path that never creates a device context."""
import pickle

import numpy as np
import pytest

import emcee_amd
from emcee_amd import backends, moves
from emcee_amd.state import State


def check_rstate(a, b):
    assert all(np.allclose(a_, b_) for a_, b_ in zip(a[1:], b[1:]))


def test_state_back_compat_and_indexing():
    np.random.seed(1234)
    coords = np.random.randn(16, 3)
    log_prob = np.random.randn(len(coords))
    blobs = np.random.randn(len(coords))
    rstate = np.random.get_state()
    state = State(coords, log_prob, blobs, rstate)
    c, l, r, b = state
    assert np.allclose(coords, c) and np.allclose(log_prob, l) and np.allclose(blobs, b)
    check_rstate(rstate, r)
    c, l, r = State(coords, log_prob, None, rstate)
    assert np.allclose(coords, c) and np.allclose(log_prob, l)
    np.testing.assert_allclose(state[0], state.coords)
    np.testing.assert_allclose(state[1], state.log_prob)
    check_rstate(state[2], state.random_state)
    np.testing.assert_allclose(state[3], state.blobs)
    np.testing.assert_allclose(state[-1], state.blobs)
    with pytest.raises(IndexError):
        state[4]
    copy = State(state, copy=True)
    copy.coords[0, 0] += 1
    assert state.coords[0, 0] != copy.coords[0, 0]


def test_backend_uninitialised_errors():
    be = backends.Backend()
    with pytest.raises(AttributeError):
        be.get_last_sample()
    for k in ("chain", "log_prob", "blobs"):
        with pytest.raises(AttributeError):
            getattr(be, "get_" + k)()


def lp_plain(x):
    return -0.5 * np.sum(x ** 2, axis=1)


def lp_blob(x):          # vectorised form: one row [log_prob, blob] per walker (reference ensemble.py:504-527)
    return np.column_stack([-0.5 * np.sum(x ** 2, axis=1), np.sum(x, axis=1)])


def run_sampler(be, nwalkers=16, ndim=3, nsteps=12, seed=1234, blobs=False, thin_by=1):
    np.random.seed(seed)
    coords = np.random.randn(nwalkers, ndim)
    s = emcee_amd.EnsembleSampler(nwalkers, ndim, lp_blob if blobs else lp_plain, backend=be, vectorize=True,
                                  moves=moves.GaussianMove(0.3))
    s.run_mcmc(coords, nsteps, thin_by=thin_by)
    return s


def test_blob_usage_errors():
    be = backends.Backend()
    run_sampler(be, blobs=True)
    with pytest.raises(ValueError):
        run_sampler(be, blobs=False)
    be = backends.Backend()
    run_sampler(be, blobs=False)
    with pytest.raises(ValueError):
        run_sampler(be, blobs=True)


def test_backend_contents_thinning_and_restart():
    s1 = run_sampler(backends.Backend(), nsteps=12)
    assert s1.get_chain().shape == (12, 16, 3) and s1.get_log_prob().shape == (12, 16) and s1.get_blobs() is None
    assert s1.get_chain(flat=True).shape == (12 * 16, 3)
    assert np.array_equal(s1.get_chain(discard=2, thin=3), s1.get_chain()[2 + 3 - 1::3])
    assert s1.iteration == 12 and s1.backend.accepted.shape == (16,)
    np.testing.assert_allclose(s1.get_log_prob(), lp_plain(s1.get_chain().reshape(-1, 3)).reshape(12, 16))
    last = s1.get_last_sample()
    assert np.array_equal(last.coords, s1.get_chain()[-1])
    # restart: continue from the stored state with the stored RNG state == one uninterrupted run
    s2 = run_sampler(backends.Backend(), nsteps=5)
    s2.run_mcmc(None, 7)
    assert np.array_equal(s2.get_chain(), s1.get_chain())
    check_rstate(s2.get_last_sample().random_state, last.random_state)
    # thin_by keeps every third step and counts acceptances of the kept steps only
    s3 = run_sampler(backends.Backend(), nsteps=4, thin_by=3)
    assert np.array_equal(s3.get_chain(), s1.get_chain()[2::3]) and s3.iteration == 4
    # reset
    s3.reset()
    assert s3.iteration == 0
    with pytest.raises(AttributeError):
        s3.get_chain()


def test_sampler_pickles_and_input_is_not_overwritten():
    np.random.seed(3)
    p0 = np.random.randn(16, 2)
    keep = p0.copy()
    s = emcee_amd.EnsembleSampler(16, 2, lp_plain, vectorize=True, moves=moves.GaussianMove([0.2, 0.4], mode="random"))
    s.run_mcmc(p0, 6)
    assert np.array_equal(p0, keep)
    s2 = pickle.loads(pickle.dumps(s))
    assert np.array_equal(s2.get_chain(), s.get_chain())
    a, b = s.run_mcmc(None, 3), s2.run_mcmc(None, 3)
    assert np.array_equal(a.coords, b.coords)


def test_errors_of_the_generic_path():
    s = emcee_amd.EnsembleSampler(16, 2, lp_plain, vectorize=True, moves=moves.GaussianMove(0.1))
    p0 = np.random.RandomState(0).randn(16, 2)
    with pytest.raises(ValueError, match="incompatible input dimensions"):
        s.run_mcmc(p0[:, :1], 2)
    with pytest.raises(ValueError, match="large condition number"):
        s.run_mcmc(np.ones((16, 2)), 2)
    with pytest.raises(ValueError, match="'store' must be False"):
        next(s.sample(p0, iterations=None, store=True))
    with pytest.raises(ValueError, match="Invalid thinning"):
        s.run_mcmc(p0, 2, thin_by=0)
    bad = emcee_amd.EnsembleSampler(16, 2, lambda x: np.full(len(x), np.nan), vectorize=True, moves=moves.GaussianMove(0.1))
    with pytest.raises(ValueError, match="NaN"):
        bad.run_mcmc(p0, 2)
    with pytest.raises(ValueError, match="rng must be"):
        emcee_amd.EnsembleSampler(16, 2, lp_plain, rng="xorshift")


# ---- named parameters, scalar-likes, progress bar (reference unit/test_ensemble.py, unit/test_pbar.py) -----------
def test_ndarray_to_list_of_dicts():
    import string
    from emcee_amd.ensemble import ndarray_to_list_of_dicts
    for n_keys in (1, 2, 10, 26):
        keys = list(string.ascii_lowercase[:n_keys])
        key_dict = {key: i for i, key in enumerate(keys)}
        for N in (1, 2, 3, 10, 100):
            x = np.random.rand(N, n_keys)
            lod = ndarray_to_list_of_dicts(x, key_dict)
            assert len(lod) == N
            for i, dct in enumerate(lod):
                assert dct.keys() == set(keys)
                for j, key in enumerate(keys):
                    assert dct[key] == x[i, j]


class _Data:
    x = np.random.RandomState(0).randn(100)


def _lnpdf(pars):
    mean, var = pars["mean"], pars["var"]
    if var <= 0:
        return -np.inf
    return -0.5 * ((mean - _Data.x) ** 2 / var + np.log(2 * np.pi * var)).sum()


def _lnpdf_grouped(pars):
    mean1, mean2 = pars["means"]
    var1, var2 = pars["vars"]
    if var1 <= 0 or var2 <= 0:
        return -np.inf
    return -0.5 * ((mean1 - _Data.x) ** 2 / var1 + np.log(2 * np.pi * var1) + (mean2 - _Data.x - 3) ** 2 / var2
                   + np.log(2 * np.pi * var2)).sum() + pars["constant"]


def test_named_parameters_construction_and_asserts():
    names = ["mean", "var"]
    s = emcee_amd.EnsembleSampler(10, 2, _lnpdf, parameter_names=names)
    assert s.params_are_named and list(s.parameter_names.keys()) == names
    with pytest.raises(AssertionError):
        emcee_amd.EnsembleSampler(10, 1, _lnpdf, parameter_names=names)              # ndim / names mismatch
    with pytest.raises(AssertionError):
        emcee_amd.EnsembleSampler(10, 3, _lnpdf, parameter_names=["a", "b", "a"])    # duplicates
    with pytest.raises(AssertionError):
        emcee_amd.EnsembleSampler(10, 2, _lnpdf, parameter_names=names, vectorize=True)


def test_named_parameters_compute_log_prob_and_run():
    for N in (4, 8, 10):
        s = emcee_amd.EnsembleSampler(N, 2, _lnpdf, parameter_names=["mean", "var"])
        lnps, _ = s.compute_log_prob(np.random.rand(N, 2))
        assert len(lnps) == N and lnps.dtype == np.float64
    grouped = {"means": [0, 1], "vars": [2, 3], "constant": 4}
    for N in (8, 10, 20):
        s = emcee_amd.EnsembleSampler(N, 5, _lnpdf_grouped, parameter_names=grouped)
        lnps, _ = s.compute_log_prob(np.random.rand(N, 5))
        assert len(lnps) == N and lnps.dtype == np.float64
    # sort of an integration test, on a move that needs no GPU
    s = emcee_amd.EnsembleSampler(4, 2, _lnpdf, parameter_names=["mean", "var"], moves=moves.GaussianMove(0.01))
    res = s.run_mcmc(np.random.rand(4, 2), 50)
    assert res.coords.shape == (4, 2)
    assert s.chain.shape == (4, 50, 2)            # deprecated (walker, step, dim) spelling


def test_log_prob_fn_may_return_scalar_likes():
    def base(x):
        return float(np.log(np.sqrt(np.pi) * np.exp(-((x / 2.0) ** 2)))[0])

    for fn in (base, lambda x: np.array([base(x)]), lambda x: np.float64(base(x)), lambda x: np.array(base(x))):
        init = np.random.default_rng(1).random((50, 1))
        s = emcee_amd.EnsembleSampler(50, 1, fn, moves=moves.GaussianMove(0.5))
        s.run_mcmc(init, 20)
        assert s.get_log_prob().shape == (20, 50)


def test_progress_bar_modes():
    from emcee_amd.pbar import get_progress_bar
    with get_progress_bar(False, 100) as bar:
        bar.update(3)                                   # the no-op bar swallows updates
    pytest.importorskip("tqdm")
    import tqdm.asyncio
    import tqdm.std
    for mode, cls in ((True, tqdm.asyncio.tqdm_asyncio), ("std", tqdm.std.tqdm), ("auto", tqdm.asyncio.tqdm_asyncio),
                      ("autonotebook", tqdm.std.tqdm)):
        with get_progress_bar(mode, 10, disable=True) as bar:
            bar.update(1)
            assert isinstance(bar._bar, cls), (mode, type(bar._bar))
    with pytest.raises(ImportError):
        get_progress_bar("no_such_flavour", 10)
    s = emcee_amd.EnsembleSampler(8, 1, lambda x: -0.5 * np.sum(x * x), moves=moves.GaussianMove(0.5))
    s.run_mcmc(np.random.RandomState(2).randn(8, 1), 5, progress=True, progress_kwargs={"disable": True})


def test_extended_precision_input_is_refused_not_truncated():
    """reference integration/test_longdouble.py keeps np.longdouble end to end; a float64 device state cannot, and
    says so instead of silently rounding an ensemble that only differs beyond the 16th digit"""
    if np.dtype(np.longdouble).itemsize <= 8:
        pytest.skip("longdouble is float64 on this platform")
    mjd = np.longdouble(58000.0)
    sigma = 100 * np.finfo(np.longdouble).eps * mjd
    p0 = sigma * np.random.RandomState(0).randn(20, 1).astype(np.longdouble) + mjd
    s = emcee_amd.EnsembleSampler(20, 1, lambda x: -0.5 * np.sum(((x - mjd) / sigma) ** 2), moves=moves.GaussianMove(0.5))
    with pytest.raises(TypeError, match="float64"):
        s.run_mcmc(p0, 5, skip_initial_state_check=True)
    s.run_mcmc(np.asarray(p0 - mjd, dtype=np.float64), 5, skip_initial_state_check=True)     # an explicit cast is fine
