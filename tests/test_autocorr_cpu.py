"""autocorr on the host: known answer (reference unit/test_autocorr.py:19-54) and equality with
the reference implementation when it is importable (build container only)."""
import numpy as np
import pytest

from emcee_amd import autocorr
from oracle import ref_shim


def ar1(seed=1234, ndim=3, N=100000, a=0.9):
    rs = np.random.RandomState(seed)
    x = np.empty((N, ndim))
    x[0] = np.zeros(ndim)
    for i in range(1, N):
        x[i] = x[i - 1] * a + rs.rand(ndim)
    return x


def test_known_answer_ar1():
    x = ar1()
    tau = autocorr.integrated_time(x[:, 0])
    assert np.all(np.abs(tau - 19.0) / 19.0 < 0.2)
    t2 = autocorr.integrated_time(x, has_walkers=False)
    for d in range(3):
        np.testing.assert_allclose(autocorr.integrated_time(x[:, d])[0], t2[d], rtol=1e-10)


def test_too_short_raises_and_quiet():
    x = ar1(N=300)
    with pytest.raises(autocorr.AutocorrError):
        autocorr.integrated_time(x[:, 0])
    assert np.isfinite(autocorr.integrated_time(x[:, 0], quiet=True)).all()
    with pytest.raises(ValueError):
        autocorr.function_1d(np.zeros((3, 3)))


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_equals_reference():
    emcee = ref_shim.import_reference()
    rs = np.random.RandomState(3)
    x = np.cumsum(rs.randn(3000, 5, 2), axis=0) * 0.01 + rs.randn(3000, 5, 2)
    np.testing.assert_allclose(autocorr.integrated_time(x, quiet=True),
                               emcee.autocorr.integrated_time(x, quiet=True), rtol=1e-9)
    np.testing.assert_allclose(autocorr.function_1d(x[:, 0, 0]), emcee.autocorr.function_1d(x[:, 0, 0]), atol=1e-12)


def test_nd_layouts_agree():
    """reference unit/test_autocorr.py:31-54: (steps, dims) with has_walkers=False == (steps, 1, dims); a multi-dimension
    estimate equals the per-dimension ones"""
    x = ar1(N=10000)
    np.testing.assert_allclose(autocorr.integrated_time(x[:, np.newaxis]), autocorr.integrated_time(x, has_walkers=False))
    xs = np.random.RandomState(42).randn(16384, 2)
    multi = autocorr.integrated_time(xs[:, np.newaxis])
    single = np.array([autocorr.integrated_time(xs[:, i]) for i in range(2)]).squeeze()
    np.testing.assert_allclose(multi, single)
