"""Device-resident log-probability targets.

Passing one of these as ``log_prob_fn`` lets :class:`emcee_amd.EnsembleSampler` fuse the
batched log-prob evaluation (reference ``ensemble.py:458-553``) into the half-step kernel.
Any other callable still works: proposals and the Metropolis accept stay on the GPU and only
the callable itself runs on the host (split-phase path).

Calling a target object evaluates it ON THE DEVICE (through ``emx_eval_log_prob``); there is
no NumPy twin in the product.  The formulas are restated for the parity tests in
``oracle/sampler_oracle.py`` only.
"""
import numpy as np

from . import _lib

__all__ = ["DeviceTarget", "IsoGaussian", "DiagGaussian", "DenseGaussian", "Rosenbrock", "UniformBox", "DeviceCallable", "DeviceKernel"]


class DeviceTarget(object):
    """Base class: subclasses provide ``kind`` and the parameter arrays."""

    kind = _lib.TARGET_HOST
    ndim = None

    def emx_params(self):
        """-> (kind, p0, p1, scale) for emx_set_target."""
        return self.kind, None, None, 0.0

    def bind(self, ens):
        kind, p0, p1, scale = self.emx_params()
        ens.set_target(kind, p0, p1, scale)

    def __call__(self, x):
        from .device import DeviceEnsemble
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        rows = np.atleast_2d(x)
        key = rows.shape[1]
        cache = self.__dict__.setdefault("_eval_ctx", {})
        ens = cache.get(key)
        if ens is None:
            ens = DeviceEnsemble(max(1024, 2), key)
            self.bind(ens)
            cache[key] = ens
        out = ens.eval_log_prob(rows)
        return float(out[0]) if single else out

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_eval_ctx", None)
        return d


class IsoGaussian(DeviceTarget):
    """log p = -0.5 sum(x^2)   (reference tests/integration/test_proposal.py:21-22)."""
    kind = _lib.TARGET_ISO


class DiagGaussian(DeviceTarget):
    """log p = -0.5 sum(ivar (x - mean)^2)   (reference docs/index.rst:41-45)."""
    kind = _lib.TARGET_DIAG

    def __init__(self, mean, ivar):
        self.mean = np.ascontiguousarray(mean, dtype=np.float64)
        self.ivar = np.ascontiguousarray(ivar, dtype=np.float64)
        if self.mean.shape != self.ivar.shape or self.mean.ndim != 1:
            raise ValueError("mean and ivar must be 1-d arrays of equal length")
        self.ndim = len(self.mean)

    def emx_params(self):
        return self.kind, self.mean, self.ivar, 0.0


class DenseGaussian(DeviceTarget):
    """log p = -0.5 (x - mean)^T icov (x - mean)   (reference docs/tutorials/quickstart.ipynb:76).

    Evaluated with v_mfma_f64_16x16x4_f64 against the Cholesky factor of ``icov``: LDS-resident and fused into the
    half-step kernel up to ndim 128, streamed through LDS by a log-prob kernel of its own up to ndim 2048."""
    kind = _lib.TARGET_DENSE

    def __init__(self, mean, icov):
        self.mean = np.ascontiguousarray(mean, dtype=np.float64)
        self.icov = np.ascontiguousarray(icov, dtype=np.float64)
        n = len(self.mean)
        if self.icov.shape != (n, n):
            raise ValueError("icov must be (ndim, ndim)")
        self.ndim = n

    def emx_params(self):
        return self.kind, self.mean, self.icov, 0.0


class Rosenbrock(DeviceTarget):
    """log p = -sum_i [100 (x_{i+1} - x_i^2)^2 + (1 - x_i)^2] / scale   (BASELINE config 3)."""
    kind = _lib.TARGET_ROSENBROCK

    def __init__(self, scale=20.0):
        self.scale = float(scale)

    def emx_params(self):
        return self.kind, None, None, self.scale


class UniformBox(DeviceTarget):
    """0 inside [0, 1]^ndim, -inf outside   (reference test_proposal.py:25-28)."""
    kind = _lib.TARGET_BOX


class DeviceCallable(DeviceTarget):
    """A user's vectorised ``log_prob_fn`` that runs on the GPU: ``fn(q)`` receives the ``(n, ndim)`` block of a split's
    proposals as a float64 CUDA tensor -- a zero-copy view of the library's buffer, rows in the order the reference passes them
    (``ensemble.py:486-487``, called at ``red_blue.py:93``) -- and returns their ``n`` log-probabilities as a CUDA tensor.
    Nothing crosses PCIe and nothing synchronises per split: the proposal kernel, ``fn``'s kernels and the accept / commit kernel
    are enqueued on one stream, and ``run_mcmc`` stays one native call.  ``-inf`` is legal; NaN raises the reference's error.
    Blobs are not supported on this path (use an ordinary callable).

        mu_t, icov_t = torch.as_tensor(mu).cuda(), torch.as_tensor(icov).cuda()
        def log_prob(q):                       # q: torch.float64 (n, ndim) on the GPU
            d = q - mu_t
            return -0.5 * ((d @ icov_t) * d).sum(1)
        sampler = EnsembleSampler(nwalkers, ndim, DeviceCallable(log_prob))
    """
    kind = _lib.TARGET_CALLBACK

    def __init__(self, fn, graph=False):
        """``graph=True``: after two eager calls per split shape the kernels ``fn`` launches are captured in a HIP graph and
        replayed (``fn`` must then be a pure function of its argument: same operations, same shapes every call)."""
        if not callable(fn):
            raise TypeError("DeviceCallable needs a callable")
        self.fn = fn
        self.graph = bool(graph)

    def bind(self, ens):
        if getattr(ens, "_cb_owner", None) is not self:
            ens.set_target_callback(self.fn, graph=self.graph)
            ens._cb_owner = self


class DeviceKernel(DeviceTarget):
    """A native log-probability: a C function with the signature ``emx_device_log_prob_fn`` of ``include/emx.h`` (typically one
    that launches the user's own HIP kernel on the stream it is handed) and its opaque ``user`` pointer.  The whole step stays
    three kernel launches per split -- proposal, the user's kernel, accept / commit -- with no Python in between."""
    kind = _lib.TARGET_CALLBACK

    def __init__(self, fn_ptr, user_ptr=None):
        self.fn_ptr, self.user_ptr = fn_ptr, user_ptr

    def bind(self, ens):
        if getattr(ens, "_cb_owner", None) is not self:
            ens.set_target_callback_c(self.fn_ptr, self.user_ptr)
            ens._cb_owner = self
