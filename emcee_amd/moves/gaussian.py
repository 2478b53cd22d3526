"""GaussianMove: Metropolis step with a Gaussian proposal (reference ``moves/gaussian.py:10-118``).

``cov`` may be a scalar (isotropic), a vector (axis-aligned) or a square matrix (general).  Modes:
``"vector"`` moves every coordinate, ``"random"`` one uniformly chosen coordinate per walker,
``"sequential"`` one coordinate per call, cycling.  ``factor`` rescales the step by
``exp(U(-ln factor, ln factor))``, one draw per call.

Draw order per call, which the bit-exact tests rely on (``gaussian.py:86-103``): the factor
uniform (if any), the full ``(nwalkers, ndim)`` normal block (even when only one coordinate is
kept), then the ``randint`` column picks of the ``"random"`` mode.  The matrix form draws ONE
``multivariate_normal`` displacement per call and applies it to every walker, as the reference
does (``gaussian.py:114-117``)."""
import numpy as np

from .mh import MHMove

__all__ = ["GaussianMove"]

_MODES = ("vector", "random", "sequential")


class _GaussianStep(object):
    """The proposal callable handed to MHMove; ``kind`` is 'iso', 'diag' or 'full'."""

    def __init__(self, kind, scale, factor, mode):
        allowed = ["vector"] if kind == "full" else list(_MODES)
        if factor is not None and factor < 1.0:
            raise ValueError("'factor' must be >= 1.0")
        if mode not in allowed:
            raise ValueError("'{0}' is not a recognized mode. Please select from: {1}".format(mode, allowed))
        self.kind = kind
        self.scale = scale              # standard deviation(s); the covariance matrix for 'full'
        self.mode = mode
        self.index = 0                  # next coordinate of the sequential mode
        self._log_factor = None if factor is None else np.log(factor)

    def _stretch(self, rng):
        if self._log_factor is None:
            return 1.0
        return np.exp(rng.uniform(-self._log_factor, self._log_factor))

    def _displaced(self, rng, x0):
        f = self._stretch(rng)                       # drawn before the normals
        if self.kind == "full":
            return x0 + f * rng.multivariate_normal(np.zeros(len(self.scale)), self.scale)
        return x0 + f * self.scale * rng.randn(*x0.shape)

    def __call__(self, x0, rng):
        n, d = x0.shape
        moved = self._displaced(rng, x0)
        zeros = np.zeros(n)
        if self.mode == "vector":
            return moved, zeros
        if self.mode == "random":
            col = rng.randint(d, size=n)
        else:
            col = np.full(n, self.index % d, dtype=int)
            self.index = (self.index + 1) % d
        rows = np.arange(n)
        out = np.array(x0)
        out[rows, col] = moved[rows, col]
        return out, zeros


class GaussianMove(MHMove):
    """:param cov: scalar, vector or square matrix.  :param mode: ``"vector"`` | ``"random"`` |
    ``"sequential"``.  :param factor: optional step-size jitter (>= 1).

    With a device target the scalar and vector forms run fused on the GPU (``EMX_MOVE_GAUSS``:
    displacement rows from ``k_gauss_disp`` / ``k_gauss_scale``, proposal + log-prob + accept + commit
    in ``emx::k_halfstep<..., MOVE_GAUSS, ...>``); the matrix form keeps its host proposal."""

    _fused_only = True       # with a host log_prob_fn the proposal stays on the host as well

    def _is_native(self):
        step = self.get_proposal
        return isinstance(step, _GaussianStep) and step.kind in ("iso", "diag")

    def _desc(self, ndim):
        from .. import _lib
        step = self.get_proposal
        mode = {"vector": _lib.GAUSS_VECTOR, "random": _lib.GAUSS_RANDOM, "sequential": _lib.GAUSS_SEQUENTIAL}[step.mode]
        jitter = step._log_factor is not None
        sigma = float(step.scale) if step.kind == "iso" else 0.0
        return _lib.MoveDesc(_lib.MOVE_GAUSS, 1, 0, mode, 1.0 if jitter else 0.0, sigma,
                             float(step._log_factor) if jitter else 0.0, float(step.index))

    def _scale_vector(self):
        step = self.get_proposal
        return np.asarray(step.scale, dtype=np.float64) if step.kind == "diag" else None

    def __init__(self, cov, mode="vector", factor=None):
        try:
            float(cov)
        except TypeError:
            cov = np.atleast_1d(cov)
            if cov.ndim == 1:
                step, ndim = _GaussianStep("diag", np.sqrt(cov), factor, mode), len(cov)
            elif cov.ndim == 2 and cov.shape[0] == cov.shape[1]:
                step, ndim = _GaussianStep("full", cov, factor, mode), cov.shape[0]
            else:
                raise ValueError("Invalid proposal scale dimensions")
        else:
            step, ndim = _GaussianStep("iso", np.sqrt(cov), factor, mode), None
        super(GaussianMove, self).__init__(step, ndim=ndim)
