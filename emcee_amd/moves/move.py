"""Move: the base of the plugin surface (reference ``moves/move.py:8-45``)."""
import numpy as np

__all__ = ["Move"]


class Move(object):
    """A sampler move.  The driver calls ``propose(model, state) -> (state, accepted)`` once per
    step and ``tune(state, accepted)`` when tuning is on (reference ``ensemble.py:409-413``)."""

    def tune(self, state, accepted):
        pass

    def update(self, old_state, new_state, accepted, subset=None):
        """Commit the accepted rows of ``new_state`` into ``old_state`` in place.

        Host-side helper kept for user-written moves (reference ``moves/move.py:12-45``:
        ``m1 = subset & accepted; m2 = accepted[subset]``).  The built-in moves never call it:
        their commit happens inside the half-step kernel."""
        n = len(old_state.coords)
        subset = np.ones(n, dtype=bool) if subset is None else subset
        dst = subset & accepted
        src = accepted[subset]
        old_state.coords[dst] = new_state.coords[src]
        old_state.log_prob[dst] = new_state.log_prob[src]
        if new_state.blobs is not None:
            if old_state.blobs is None:
                raise ValueError("If you start sampling with a given log_prob, you also need to provide the "
                                 "current list of blobs at that position.")
            old_state.blobs[dst] = new_state.blobs[src]
        return old_state
