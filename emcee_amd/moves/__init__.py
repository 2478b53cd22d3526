"""The Move plugin surface of the hot path (reference ``moves/__init__.py``): the base classes
and the three split-ensemble moves named by the north star.  MHMove / GaussianMove / WalkMove /
KDEMove are out of scope for this path; reference instances of
them can still be passed to the sampler -- they run through their own ``propose``."""
from .de import DEMove
from .de_snooker import DESnookerMove
from .move import Move
from .red_blue import RedBlueMove
from .stretch import StretchMove

__all__ = ["Move", "RedBlueMove", "StretchMove", "DEMove", "DESnookerMove"]
