"""The Move plugin surface (reference ``moves/__init__.py``).

StretchMove / DEMove / DESnookerMove are the split-ensemble moves of the hot path: proposal,
log-prob, accept and commit fused in ``emx::k_halfstep``.  MHMove / GaussianMove propose from the
walker's own position (no complement); WalkMove / KDEMove are split-ensemble moves with host-side
proposals that use the device accept/commit (``emx_accept_proposals``)."""
from .de import DEMove
from .de_snooker import DESnookerMove
from .gaussian import GaussianMove
from .kde import KDEMove
from .mh import MHMove
from .move import Move
from .red_blue import RedBlueMove
from .stretch import StretchMove
from .walk import WalkMove

__all__ = ["Move", "MHMove", "GaussianMove", "RedBlueMove", "StretchMove", "WalkMove", "KDEMove", "DEMove",
           "DESnookerMove"]
