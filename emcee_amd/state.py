"""State: the ensemble snapshot handed between sampler, moves and backends.

Mirrors reference ``state.py:10-75`` (same slots, copy semantics, tuple-style unpacking)."""
from copy import deepcopy

import numpy as np

__all__ = ["State"]


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
