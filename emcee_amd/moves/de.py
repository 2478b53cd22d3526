"""DEMove: differential-evolution proposal (reference ``moves/de.py:11-77``).

q = s + gamma (c[j] - c[i]) for an ordered pair (i != j) of complement walkers,
gamma = g0 (1 + sigma N(0,1)), g0 = gamma0 or 2.38 / sqrt(2 ndim); Metropolis factor 0.
The reference materialises all nc (nc-1) ordered pairs (17 GB at nwalkers = 65536); here the
drawn pair index is decoded in closed form (host plan producer / in-kernel for Philox)."""
import numpy as np

from .. import _lib
from .red_blue import RedBlueMove
from .stretch import _device_get_proposal

__all__ = ["DEMove"]


class DEMove(RedBlueMove):
    """Args: ``sigma`` (std-dev of the stretch of the proposal vector, default 1e-5),
    ``gamma0`` (mean stretch factor, default 2.38 / sqrt(2 ndim))."""

    _native_kind = _lib.MOVE_DE

    def __init__(self, sigma=1.0e-5, gamma0=None, **kwargs):
        self.sigma = sigma
        self.gamma0 = gamma0
        super().__init__(**kwargs)

    def setup(self, coords):
        self.g0 = self.gamma0
        if self.g0 is None:
            ndim = coords.shape[1]
            self.g0 = 2.38 / np.sqrt(2 * ndim)      # reference de.py:33-38

    def _desc(self, ndim):
        g0 = self.gamma0 if self.gamma0 is not None else 2.38 / np.sqrt(2 * ndim)
        return _lib.MoveDesc(_lib.MOVE_DE, self.nsplits, int(bool(self.randomize_split)), 0, 2.0,
                             float(self.sigma), float(g0), 0.0)

    def get_proposal(self, s, c, random):
        return _device_get_proposal(self, s, c, random)
