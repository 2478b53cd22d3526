"""MHMove: whole-ensemble Metropolis-Hastings with a caller-supplied proposal (reference
``moves/mh.py:11-65``).

Unlike the split-ensemble moves every walker is proposed at once from its own position, so there
is no complement gather and nothing for the red/blue kernels to do: the proposal function is host
code by definition, the log-probabilities come from ``model.compute_log_prob_fn`` (which is the
device evaluator when the sampler was given a :class:`emcee_amd.targets.DeviceTarget`), and the
accept rule ``ln u < ln p(q) - ln p(x) + factors`` is applied with one uniform per walker drawn
after the evaluation -- the reference's order of RNG consumption (``mh.py:50-57``)."""
import numpy as np

from ..state import State
from .move import Move

__all__ = ["MHMove"]


class MHMove(Move):
    """``proposal_function(coords, rng) -> (q, ln q(x;x') - ln q(x';x))``; ``ndim`` optionally pins
    the dimension the proposal is valid for."""

    def __init__(self, proposal_function, ndim=None):
        self.get_proposal = proposal_function
        self.ndim = ndim

    def propose(self, model, state):
        n, d = state.coords.shape
        if self.ndim is not None and d != self.ndim:
            raise ValueError("Dimension mismatch in proposal")
        rng = model.random
        q, log_ratio = self.get_proposal(state.coords, rng)
        lp_q, blobs_q = model.compute_log_prob_fn(q)
        # strictly-less, and the uniforms are drawn only now (mh.py:56-57)
        accepted = np.log(rng.rand(n)) < (lp_q - state.log_prob + log_ratio)
        proposed = State(q, log_prob=lp_q, blobs=blobs_q)
        return self.update(state, proposed, accepted), accepted
