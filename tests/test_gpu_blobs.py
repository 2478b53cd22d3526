"""Blob handling of the host-callable (split-phase) path: the reference's matrix of blob shapes and the
shape-mismatch error (src/emcee/tests/unit/test_blobs.py:21-99), plus blobs_dtype and flat views."""
import warnings

import numpy as np
import pytest

import emcee_amd

pytestmark = pytest.mark.gpu


class BlobLogProb(object):
    def __init__(self, blob_function):
        self.blob_function = blob_function

    def __call__(self, params):
        return -0.5 * np.sum(params ** 2), self.blob_function(params)


@pytest.mark.parametrize("plain,ragged,blob_shape,func", [
    (True, False, 5, lambda x: np.random.randn(5)),
    (True, False, (5, 3), lambda x: np.random.randn(5, 3)),
    (True, False, (5, 3), lambda x: np.random.randn(1, 5, 1, 3, 1)),
    (True, False, 0, lambda x: np.random.randn()),
    (False, True, 2, lambda x: (1.0, np.random.randn(3))),
    (False, False, 0, lambda x: "face"),
    (False, False, 0, lambda x: object()),
    (False, False, 2, lambda x: ("face", "surface")),
    (False, True, 2, lambda x: (np.random.randn(5), "face")),
])
def test_blob_shape(plain, ragged, blob_shape, func):
    np.random.seed(42)
    coords = np.random.randn(32, 3)
    sampler = emcee_amd.EnsembleSampler(32, 3, BlobLogProb(func))
    nsteps = 10
    with warnings.catch_warnings():
        if ragged:
            warnings.simplefilter("ignore")
        sampler.run_mcmc(coords, nsteps)
    shape = [nsteps, 32]
    if isinstance(blob_shape, tuple):
        shape += blob_shape
    elif blob_shape > 0:
        shape += [blob_shape]
    assert sampler.get_blobs().shape == tuple(shape)
    if not plain:
        assert sampler.get_blobs().dtype == np.dtype("object")
    assert sampler.get_blobs(flat=True).shape[0] == nsteps * 32
    assert sampler.get_chain().shape == (nsteps, 32, 3)


class VariableLogProb:
    def __init__(self):
        self.i = 3

    def __call__(self, *args):
        return 0, np.zeros(self.i)


def test_blob_mismatch():
    np.random.seed(42)
    model = VariableLogProb()
    coords = np.random.randn(32, 3)
    sampler = emcee_amd.EnsembleSampler(32, 3, model)
    model.i += 1
    sampler.run_mcmc(coords, 1)       # blob shapes are taken from the first round of moves
    model.i += 1
    with pytest.raises(ValueError):
        sampler.run_mcmc(coords, 1)


def test_blobs_dtype_and_values_follow_the_accepted_proposals():
    def lp(p):
        return -0.5 * np.sum(p ** 2), p[0] + 1.0, int(p[1] > 0)

    dt = [("shifted", float), ("positive", int)]
    np.random.seed(1)
    coords = np.random.randn(24, 2)
    s = emcee_amd.EnsembleSampler(24, 2, lp, blobs_dtype=dt)
    s.run_mcmc(coords, 15)
    blobs, chain = s.get_blobs(), s.get_chain()
    assert blobs.dtype.names == ("shifted", "positive")
    np.testing.assert_allclose(blobs["shifted"], chain[..., 0] + 1.0)
    assert np.array_equal(blobs["positive"], (chain[..., 1] > 0).astype(int))
    last = s.get_last_sample()
    np.testing.assert_allclose(last.blobs["shifted"], last.coords[:, 0] + 1.0)
