"""WalkMove: the Goodman & Weare "walk move" (reference ``moves/walk.py:10-42``).

A split-ensemble move whose proposal is host code (a covariance of ``s`` helper walkers per
updated walker and a ``multivariate_normal`` draw); the Metropolis accept and the commit still run
on the device through :meth:`RedBlueMove._propose_custom` / ``emx_accept_proposals``."""
import numpy as np

from .red_blue import RedBlueMove

__all__ = ["WalkMove"]


class WalkMove(RedBlueMove):
    """:param s: number of helper walkers (default: the whole complement)."""

    def __init__(self, s=None, **kwargs):
        self.s = s
        super(WalkMove, self).__init__(**kwargs)

    def get_proposal(self, s, c, random):
        helpers = np.concatenate(c, axis=0)
        nc = len(helpers)
        take = nc if self.s is None else self.s
        q = np.empty_like(s)
        for k, here in enumerate(s):
            picked = random.choice(nc, take, replace=False)
            spread = np.atleast_2d(np.cov(helpers[picked], rowvar=0))
            q[k] = random.multivariate_normal(here, spread)
        return q, np.zeros(len(s), dtype=np.float64)
