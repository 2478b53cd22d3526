"""walkers_independent: the reference's cases (unit/test_sampler.py:237-321) on the host path, and --
on a GPU -- the device path (emx_walkers_independent: Householder QR on the GPU) against the host verdict."""
import numpy as np
import pytest

from emcee_amd import walkers_independent
from emcee_amd import ensemble as ens_mod


def _cases(nw, nd, rs):
    """(name, matrix, expected) scaled-up versions of the reference's independence tests."""
    w = rs.randn(nw, nd)
    p = rs.randn(nd)
    p /= np.sqrt(p @ p)
    proj = (w @ p)[:, None] * p[None, :]
    scales = np.resize(np.array([1, 1e10, 1e100, 1e200, 1e-10, 1e-100, 1e-200]), nd)
    return [
        ("ones", np.ones((nw, nd)), False),
        ("randn", w, True),
        ("offset 1e5", w + 1e5, True),
        ("offset 1e10", w + 1e10, True),
        ("offset 10/eps", w + 10 / np.finfo(float).eps, False),
        ("projected", w - proj, False),
        ("projected shifted", w - proj + p[None, :], False),
        ("squashed 1e-10", w - proj + 1e-10 * proj, False),
        ("squashed 1e-5", w - proj + 1e-5 * proj, True),
        ("scaled", w * scales[None, :], True),
        ("duplicate column", np.column_stack([w[:, :-1], w[:, 0]]), False),
        ("nan", np.where(np.arange(nw * nd).reshape(nw, nd) == 7, np.nan, w), False),
    ]


@pytest.mark.parametrize("nw,nd", [(10, 2), (20, 5), (30, 10)])
def test_reference_cases_host(nw, nd):
    rs = np.random.RandomState(nw)
    for name, m, expect in _cases(nw, nd, rs):
        assert walkers_independent(m) == expect, name
    assert not walkers_independent(rs.randn(nd, nd + 1))          # too few walkers


@pytest.mark.gpu
@pytest.mark.parametrize("nw,nd", [(10, 2), (20, 5), (30, 10), (200, 33)])
def test_reference_cases_through_the_c_abi(nw, nd):
    """emx_walkers_independent itself (Householder QR kernels + extreme singular values of the triangular factor) on the
    reference's own small cases, with the condition number next to NumPy's for the well-posed ones."""
    import ctypes as C
    from emcee_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(nw)
    for name, m, expect in _cases(nw, nd, rs):
        x = np.ascontiguousarray(m, dtype=np.float64)
        verdict, cond = C.c_int32(-1), C.c_double()
        assert lib.emx_walkers_independent(0, x, nw, nd, C.byref(verdict), C.byref(cond)) == 0
        assert bool(verdict.value) == expect, (name, cond.value)
        if expect:
            Cm = x - x.mean(axis=0)
            Cm /= np.abs(Cm).max(axis=0)
            Cm /= np.sqrt((Cm ** 2).sum(axis=0))
            np.testing.assert_allclose(cond.value, np.linalg.cond(Cm), rtol=1e-4)      # (inverse) power iteration: ample for a 1e8 threshold
    verdict = C.c_int32(-1)
    assert lib.emx_walkers_independent(0, np.ascontiguousarray(rs.randn(nd, nd + 1)), nd, nd + 1, C.byref(verdict), None) == 0
    assert verdict.value == 0                                         # too few walkers


@pytest.mark.gpu
@pytest.mark.parametrize("nw,nd", [(65536, 64), (4096, 600), (262144, 8)])
def test_device_path_agrees_with_host(nw, nd):
    rs = np.random.RandomState(nd)
    assert nw * nd >= ens_mod._DEVICE_CHECK_MIN_SIZE
    saved = ens_mod._DEVICE_CHECK_MIN_SIZE
    for name, m, expect in _cases(nw, nd, rs):
        dev = ens_mod._walkers_independent_device(m)
        assert dev is not None, "device path unavailable on a GPU box"
        try:
            ens_mod._DEVICE_CHECK_MIN_SIZE = 1 << 62          # force the reference's host algorithm
            host = walkers_independent(m)
        finally:
            ens_mod._DEVICE_CHECK_MIN_SIZE = saved
        assert dev == host, name
        if name not in ("offset 10/eps",):                   # quantised to multiples of 8: full rank again when nw is large
            assert dev == expect, name
        assert walkers_independent(m) == host, name          # the public entry point takes the device path here


@pytest.mark.gpu
@pytest.mark.parametrize("nw,nd", [(64, 5), (4096, 33)])
def test_resident_state_check_agrees_with_the_c_abi_on_host_coordinates(nw, nd):
    """emx_walkers_independent_resident (round 5): the check on the ensemble a context holds -- what a run continued from the
    State the previous run returned is checked with (reference: every sample() call, ensemble.py:316-323) -- gives the verdict
    and the condition number of emx_walkers_independent on the same coordinates; and the sampler raises the reference's
    ValueError when such a continuation starts from a collapsed ensemble."""
    import ctypes as C
    from emcee_amd import _lib, EnsembleSampler
    from emcee_amd.device import DeviceEnsemble
    lib = _lib.load()
    rs = np.random.RandomState(nd)
    for name, m, expect in _cases(nw, nd, rs):
        if name == "nan":
            continue                                      # (a context refuses a non-finite state at set_state)
        x = np.ascontiguousarray(m, dtype=np.float64)
        ens = DeviceEnsemble(nw, nd)
        ens.set_target(_lib.TARGET_ISO)
        ens.set_state(x)
        v0, c0, v1, c1 = C.c_int32(-1), C.c_double(), C.c_int32(-1), C.c_double()
        assert lib.emx_walkers_independent(0, x, nw, nd, C.byref(v0), C.byref(c0)) == 0
        assert lib.emx_walkers_independent_resident(ens.ctx, C.byref(v1), C.byref(c1)) == 0
        assert v0.value == v1.value == int(expect), name
        if expect:
            assert c0.value == c1.value, name
        assert ens.walkers_independent() == expect
        ens.close()
    # the drop-in path: a continuation from the State the sampler itself returned, after the ensemble has been collapsed behind its back
    from emcee_amd import targets
    sampler = EnsembleSampler(nw, nd, targets.IsoGaussian(), rng="philox")
    st = sampler.run_mcmc(rs.randn(nw, nd), 5)
    sampler.run_mcmc(st, 3)                                # independent: runs
    ones = np.ones((nw, nd))                               # every walker the same point -- through the C ABI directly, so that the
    assert lib.emx_set_state(sampler._ens.ctx, ones.ctypes.data, None) == 0      # State the last run returned still counts as the device state
    with pytest.raises(ValueError, match="large condition number"):
        sampler.run_mcmc(None, 3)
    sampler.run_mcmc(None, 3, skip_initial_state_check=True)
