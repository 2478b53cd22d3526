"""State: the ensemble snapshot handed between sampler, moves and backends.

Mirrors reference ``state.py:10-75`` (same slots, copy semantics, tuple-style unpacking)."""
from copy import deepcopy

import numpy as np

__all__ = ["State", "DeviceState", "ResidentState"]


class State(object):
    """coords (nwalkers, ndim), log_prob (nwalkers,), blobs, random_state.

    Iterating / indexing unpacks to ``coords, log_prob, random_state[, blobs]`` for backwards
    compatibility (reference ``state.py:57-75``)."""

    __slots__ = "coords", "log_prob", "blobs", "random_state"

    def __init__(self, coords, log_prob=None, blobs=None, random_state=None, copy=False):
        dup = deepcopy if copy else (lambda v: v)
        if hasattr(coords, "coords"):          # copy-construct from another State
            src = coords
            self.coords = dup(src.coords)
            self.log_prob = dup(src.log_prob)
            self.blobs = dup(src.blobs)
            self.random_state = dup(src.random_state)
        else:
            self.coords = dup(np.atleast_2d(coords))
            self.log_prob = dup(log_prob)
            self.blobs = dup(blobs)
            self.random_state = dup(random_state)

    def _as_tuple(self):
        base = (self.coords, self.log_prob, self.random_state)
        return base if self.blobs is None else base + (self.blobs,)

    def __len__(self):
        return len(self._as_tuple())

    def __iter__(self):
        return iter(self._as_tuple())

    def __getitem__(self, index):
        t = self._as_tuple()
        if index < 0:
            index += len(t)
        if 0 <= index < len(t):
            return t[index]
        raise IndexError("Invalid index '{0}'".format(index))

    def __repr__(self):
        return "State({0}, log_prob={1}, blobs={2}, random_state={3})".format(
            self.coords, self.log_prob, self.blobs, self.random_state)


class DeviceState(State):
    """A :class:`State` whose ``coords`` / ``log_prob`` live on the GPU and are copied to the host
    only when read.

    ``EnsembleSampler.sample`` yields one of these per iteration when the built-in moves run on the
    device: like the reference -- which mutates and re-yields one ``State`` object
    (``ensemble.py:409-424``) -- the object always reflects the sampler's *current* position, but a
    loop that never looks at the coordinates never pays the (nwalkers x ndim) PCIe copy.
    Assigning to ``coords`` / ``log_prob`` detaches that field from the device."""

    __slots__ = ("_ens", "_c", "_lp", "_rs")

    @property
    def random_state(self):
        """The sampler's current generator state (a provider is resolved on access, like the coordinates)."""
        v = self._rs
        return v() if callable(v) else v

    @random_state.setter
    def random_state(self, value):
        self._rs = value

    def __init__(self, ens, blobs=None, random_state=None):
        self._ens = ens
        self._c = None
        self._lp = None
        self.blobs = blobs
        self.random_state = random_state

    def _invalidate(self):
        """Called by the sampler after every device step."""
        self._c = None
        self._lp = None

    def _fetch(self):
        if self._ens is not None and (self._c is None or self._lp is None):
            c, lp = self._ens.get_state()
            if self._c is None:
                self._c = c
            if self._lp is None:
                self._lp = lp

    @property
    def coords(self):
        self._fetch()
        return self._c

    @coords.setter
    def coords(self, v):
        self._c = v

    @property
    def log_prob(self):
        self._fetch()
        return self._lp

    @log_prob.setter
    def log_prob(self, v):
        self._lp = v


class ResidentState(State):
    """The :class:`State` a device run returns: a snapshot of the ensemble *as of that return* whose arrays cross PCIe
    only if somebody reads them.

    Reference semantics kept (ensemble.py:312, 441-447; unit/test_state.py:35-47): the object never changes after it was
    returned, and handing it back to ``run_mcmc`` (or ``None`` -> the previous state) continues from it.  While it still *is*
    the device state the continuation uploads nothing; when a later call is about to change the ensemble and the object is
    still alive and unread, its values are kept by a device-to-device copy (``emx_snapshot_save``) and read from there on
    demand.  Reading ``coords`` / ``log_prob`` materialises plain NumPy arrays (from then on it behaves like any State: the
    caller may edit them, so the next run uploads them)."""

    __slots__ = ("_ens", "_gen", "_slot", "_c", "_lp", "__weakref__")

    def __init__(self, ens, random_state=None):
        import weakref
        self._ens = ens
        self._gen = ens._gen
        self._slot = None
        self._c = None
        self._lp = None
        self.blobs = None
        self.random_state = random_state
        ens._resident = weakref.ref(self)

    # -- what the sampler asks --
    def _is_device_state(self, ens):
        """True when continuing from this object needs no upload: nobody read (hence nobody could edit) its arrays, and the
        device still holds exactly this state."""
        return self._ens is ens and self._c is None and self._lp is None and self._slot is None and ens._gen == self._gen

    def _restore_on_device(self, ens):
        """-> True when the (unread) snapshot could be made the current device state again, inside HBM"""
        if self._ens is ens and self._c is None and self._lp is None and self._slot is not None:
            ens.snapshot_restore(self._slot)
            return True
        return False

    def _detach_before_change(self):
        """the ensemble is about to change (DeviceEnsemble._touch): keep this object's values"""
        if self._c is not None and self._lp is not None:
            return
        slot = self._ens.snapshot_save()
        if slot is None:
            self._materialise(live=True)         # no slot left: over PCIe after all
        else:
            self._slot = slot
            self._ens._snap_states.add(self)

    def _materialise(self, live=None):
        if self._c is not None and self._lp is not None:
            return
        ens = self._ens
        if self._slot is not None:
            c, lp = ens.snapshot_read(self._slot)
            ens.snapshot_release(self._slot)
            self._slot = None
        elif live or ens._gen == self._gen:
            c, lp = ens.get_state()
        else:
            raise RuntimeError("ResidentState lost its device copy")      # cannot happen: _touch snapshots first
        if self._c is None:
            self._c = c
        if self._lp is None:
            self._lp = lp

    def _peek_coords(self):
        """coordinates for a read-only internal check (the initial-state conditioning): a private copy, the object stays
        resident"""
        if self._c is not None:
            return self._c
        if self._slot is not None:
            return self._ens.snapshot_read(self._slot)[0]
        return self._ens.get_state(log_prob=False)[0]

    def __del__(self):
        try:
            if self._slot is not None:
                self._ens.snapshot_release(self._slot)
        except Exception:  # noqa: BLE001
            pass

    def __reduce__(self):
        return (State, (self.coords, self.log_prob, self.blobs, self.random_state))

    @property
    def coords(self):
        self._materialise()
        return self._c

    @coords.setter
    def coords(self, v):
        self._materialise()
        self._c = v

    @property
    def log_prob(self):
        self._materialise()
        return self._lp

    @log_prob.setter
    def log_prob(self, v):
        self._materialise()
        self._lp = v
