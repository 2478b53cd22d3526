"""RedBlueMove: the split-ensemble update driver (reference ``moves/red_blue.py:12-106``).

``propose(model, state)`` keeps the reference protocol -- it needs only ``model.random`` (a
NumPy ``RandomState``-like object) and ``model.compute_log_prob_fn`` -- but does the work on
the GPU: the split plan and every draw are produced by libemx's bit-exact MT19937 twin from
``model.random``'s state, proposals come from the half-step kernel (``emx_propose``), the
log-prob callback runs wherever the caller put it, and the Metropolis accept + commit run in
``emx_accept``.  ``model.random`` is left in exactly the state reference emcee would leave it.

Subclasses that override ``get_proposal`` (user-defined proposals, host code by nature) still
get the device accept/commit through ``emx_accept_proposals``.
"""
import numpy as np

from .. import _lib
from ..state import State
from .move import Move

__all__ = ["RedBlueMove"]


class RedBlueMove(Move):
    """Abstract red-blue ensemble move.

    Args mirror the reference (``red_blue.py:37-42``): ``nsplits`` sub-ensembles (default 2),
    ``randomize_split`` (default True), ``live_dangerously`` (skip the nwalkers >= 2 ndim guard).
    """

    _native_kind = None      # set by StretchMove / DEMove / DESnookerMove

    def __init__(self, nsplits=2, randomize_split=True, live_dangerously=False):
        self.nsplits = int(nsplits)
        self.live_dangerously = live_dangerously
        self.randomize_split = randomize_split

    def setup(self, coords):
        pass

    def get_proposal(self, sample, complement, random):
        raise NotImplementedError("The proposal must be implemented by subclasses")

    # ---- description for the device ----
    def _is_native(self):
        """True when get_proposal is the library's own (not overridden by a user subclass)."""
        if self._native_kind is None:
            return False
        for klass in type(self).__mro__:
            if "get_proposal" in klass.__dict__:
                return klass.__module__.startswith("emcee_amd.moves")
        return False

    def _desc(self, ndim):
        """-> MoveDesc for emx_set_moves (subclasses fill their parameters)."""
        raise NotImplementedError

    # ---- the plugin entry point ----
    def propose(self, model, state):
        nwalkers, ndim = state.coords.shape
        if nwalkers < 2 * ndim and not self.live_dangerously:
            raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than twice the "
                               "number of dimensions.")
        self.setup(state.coords)
        ens = self._device(nwalkers, ndim)
        ens.set_state(state.coords, state.log_prob)
        if self._is_native():
            accepted = self._propose_native(ens, model, state, ndim)
        else:
            accepted = self._propose_custom(ens, model, state)
        coords, log_prob = ens.get_state()
        state.coords[...] = coords
        state.log_prob[...] = log_prob
        return state, accepted

    def _device(self, nwalkers, ndim):
        from ..device import DeviceEnsemble
        cache = self.__dict__.setdefault("_dev", {})
        ens = cache.get((nwalkers, ndim))
        if ens is None:
            ens = DeviceEnsemble(nwalkers, ndim)
            ens.set_target(_lib.TARGET_HOST)
            cache.clear()
            cache[(nwalkers, ndim)] = ens
        return ens

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_dev", None)
        return d

    def _propose_native(self, ens, model, state, ndim):
        rng = model.random
        ens.set_moves([self._desc(ndim)], np.array([1.0]))
        ens.set_rng_mode(_lib.RNG_MT19937)
        ens.set_mt19937(rng.get_state())
        nsplits = ens.step_begin_with(0, store=False)
        blobs_mask_updates = []
        for split in range(nsplits):
            q = ens.propose(split)                                   # red_blue.py:90
            new_lp, new_blobs = model.compute_log_prob_fn(q)         # red_blue.py:93
            ens.accept(split, np.asarray(new_lp, dtype=np.float64))  # red_blue.py:96-104
            if new_blobs is not None:
                blobs_mask_updates.append((split, new_blobs))
        plan_order = ens.plan_get(nsplits) if blobs_mask_updates else None
        ens.step_end()
        ens.raise_on_status()
        accepted = ens.accepted_mask()
        rng.set_state(ens.get_mt19937())
        for split, new_blobs in blobs_mask_updates:
            if state.blobs is None:
                raise ValueError("If you start sampling with a given log_prob, you also need to provide the "
                                 "current list of blobs at that position.")
            members = plan_order["order"][plan_order["off"][split]:plan_order["off"][split + 1]]
            acc = accepted[members]
            state.blobs[members[acc]] = np.asarray(new_blobs)[acc]
        return accepted

    def _propose_custom(self, ens, model, state):
        """User get_proposal on host arrays; split bookkeeping per red_blue.py:76-87."""
        rng = model.random
        nwalkers, ndim = state.coords.shape
        accepted = np.zeros(nwalkers, dtype=bool)
        inds = np.arange(nwalkers) % self.nsplits
        if self.randomize_split:
            rng.shuffle(inds)
        ens.set_moves([_lib.MoveDesc(_lib.MOVE_STRETCH, self.nsplits, 0, 0, 2.0, 0.0, 0.0, 0.0)], np.array([1.0]))
        ens.set_rng_mode(_lib.RNG_INPUTS)
        sets_idx = [np.nonzero(inds == j)[0] for j in range(self.nsplits)]
        order = np.concatenate(sets_idx).astype(np.int32)
        off = np.concatenate([[0], np.cumsum([len(s) for s in sets_idx])]).astype(np.int32)
        coords = state.coords
        for split in range(self.nsplits):
            members = sets_idx[split]
            sets = [coords[ix] for ix in sets_idx]
            s = sets[split]
            c = sets[:split] + sets[split + 1:]
            q, factors = self.get_proposal(s, c, rng)
            new_lp, new_blobs = model.compute_log_prob_fn(q)
            uacc = np.ones(nwalkers)
            uacc[off[split]:off[split + 1]] = rng.rand(len(members))     # red_blue.py:100, one per walker
            ens.step_begin(store=False)
            ens.plan_set(0, dict(off=off, order=order, p0=order, p1=order, p2=order, s0=np.zeros(nwalkers), uacc=uacc))
            ens.accept_proposals(split, q, factors, np.asarray(new_lp, dtype=np.float64))
            ens.step_end()
            ens.raise_on_status()
            acc = ens.accepted_mask()[members]
            accepted[members] = acc
            coords_dev, _ = ens.get_state(log_prob=False)
            coords[members] = coords_dev[members]              # later splits see the update (red_blue.py:85)
            if new_blobs is not None:
                if state.blobs is None:
                    raise ValueError("If you start sampling with a given log_prob, you also need to provide "
                                     "the current list of blobs at that position.")
                state.blobs[members[acc]] = np.asarray(new_blobs)[acc]
        return accepted
