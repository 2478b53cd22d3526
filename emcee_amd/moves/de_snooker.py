"""DESnookerMove: snooker differential-evolution proposal (reference ``moves/de_snooker.py:10-46``).

Four sub-ensembles; for each walker one member z, z1, z2 of each complement set (randomly
ordered), q = s + u gammas (u.z1 - u.z2) with u = (s - z)/|s - z|, Metropolis factor
(ndim-1) (ln|q - z| - ln|s - z|).  The reference loops over walkers in Python; here every walker
is one G-lane group of ``emx::k_halfstep<..., MOVE_SNOOKER, ...>``."""
from .. import _lib
from .red_blue import RedBlueMove
from .stretch import _device_get_proposal

__all__ = ["DESnookerMove"]


class DESnookerMove(RedBlueMove):
    """Args: ``gammas`` (mean stretch factor, default 1.7).  ``nsplits`` is forced to 4."""

    _native_kind = _lib.MOVE_SNOOKER

    def __init__(self, gammas=1.7, **kwargs):
        self.gammas = gammas
        kwargs["nsplits"] = 4
        super(DESnookerMove, self).__init__(**kwargs)

    def _desc(self, ndim):
        return _lib.MoveDesc(_lib.MOVE_SNOOKER, self.nsplits, int(bool(self.randomize_split)), 0, 2.0, 0.0, 0.0,
                             float(self.gammas))

    def get_proposal(self, s, c, random):
        return _device_get_proposal(self, s, c, random)
