"""StretchMove: the Goodman & Weare stretch move (reference ``moves/stretch.py:11-33``).

z ~ g(z) via zz = ((a-1) u + 1)^2 / a, proposal c[rint] - (c[rint] - s) zz, Metropolis factor
(ndim-1) ln zz.  Evaluated in ``emx::k_halfstep<..., MOVE_STRETCH, ...>``."""
from .. import _lib
from .red_blue import RedBlueMove

__all__ = ["StretchMove"]


class StretchMove(RedBlueMove):
    """:param a: the stretch scale parameter (default 2.0)."""

    _native_kind = _lib.MOVE_STRETCH

    def __init__(self, a=2.0, **kwargs):
        self.a = a
        super(StretchMove, self).__init__(**kwargs)

    def _desc(self, ndim):
        return _lib.MoveDesc(_lib.MOVE_STRETCH, self.nsplits, int(bool(self.randomize_split)), 0,
                             float(self.a), 0.0, 0.0, 0.0)

    def get_proposal(self, s, c, random):
        return _device_get_proposal(self, s, c, random)


def _device_get_proposal(move, s, c, random):
    """get_proposal(s, c, random) -> (q, factors) for the built-in moves, on the device.

    Kept so that code calling the reference's ``get_proposal`` hook directly keeps working:
    the sub-ensembles are uploaded as one temporary ensemble [s; c...] with a fixed split and
    the half-step kernel is run in propose-only mode with draws taken from ``random``'s
    MT19937 stream (bit-exact twin), leaving ``random`` where reference emcee would."""
    import numpy as np
    from ..device import DeviceEnsemble
    s = np.ascontiguousarray(s, dtype=np.float64)
    csets = [np.ascontiguousarray(x, dtype=np.float64) for x in c]
    ns, ndim = s.shape
    allc = np.concatenate([s] + csets, axis=0)
    n = len(allc)
    ens = DeviceEnsemble(n, ndim)
    ens.set_target(_lib.TARGET_HOST)
    ens.set_state(allc, np.zeros(n))
    # plan for split 0 = s, complement sets follow in order; draws from `random` in reference order
    from .. import _hostplan
    nsplits = 1 + len(csets)
    plan = _hostplan.single_split_plan(move, ns, [len(x) for x in csets], ndim, random)
    desc = move._desc(ndim)
    desc.nsplits = nsplits
    ens.set_moves([desc], np.array([1.0]))
    ens.set_rng_mode(_lib.RNG_INPUTS)
    ens.step_begin(store=False)
    ens.plan_set(0, plan)
    q, factors = ens.propose(0, with_factors=True)
    q, factors = q.copy(), factors.copy()
    ens.step_end()
    ens.close()
    return q, factors
