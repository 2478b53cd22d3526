"""emcee_amd: an MI355X-native affine-invariant ensemble sampler, drop-in for the hot path of
dfm/emcee (``EnsembleSampler`` / ``sample()`` / ``get_chain()`` and the ``Move`` plugin surface).

The red/blue split-ensemble update -- complement gather, stretch / DE / snooker proposal, batched
log-probability, Metropolis accept, commit, chain append -- runs as hand-written HIP kernels for
gfx950 behind a C ABI (``include/emx.h``, ``emcee_amd/libemx.so``).  See DESIGN.md.
"""
__version__ = "0.1.0"

from . import autocorr, backends, moves, targets
from .ensemble import EnsembleSampler, walkers_independent
from .state import State

__all__ = ["EnsembleSampler", "walkers_independent", "State", "moves", "autocorr", "backends", "targets",
           "__version__"]
