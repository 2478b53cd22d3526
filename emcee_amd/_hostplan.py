"""Host-side access to libemx's bit-exact MT19937 twin (draw bookkeeping only, no sampler math)."""
import ctypes as C

import numpy as np

from . import _lib


class HostMT(object):
    """A C++ MT19937/legacy-RandomState generator seeded from a NumPy state tuple."""

    def __init__(self, state):
        self.lib = _lib.load()
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        self.h = self.lib.emx_mt_create(key, int(state[2]), int(state[3]), float(state[4]))

    def __del__(self):
        try:
            self.lib.emx_mt_destroy(self.h)
        except Exception:  # noqa: BLE001
            pass

    def get_state(self):
        key = np.empty(624, dtype=np.uint32)
        pos, hg, cached = C.c_int32(), C.c_int32(), C.c_double()
        self.lib.emx_mt_get_state(self.h, key, C.byref(pos), C.byref(hg), C.byref(cached))
        return ("MT19937", key, pos.value, hg.value, cached.value)


def single_split_plan(move, ns, nc_sizes, ndim, random):
    """Draws of one ``get_proposal(s, c, random)`` call for a temporary ensemble laid out as
    [s; c[0]; c[1]; ...]; consumes ``random`` exactly as the reference move would."""
    lib = _lib.load()
    n = ns + int(sum(nc_sizes))
    off = np.concatenate([[0], np.cumsum([ns] + list(nc_sizes))]).astype(np.int32)
    order = np.arange(n, dtype=np.int32)
    desc = move._desc(ndim)
    desc.nsplits = len(off) - 1
    mt = HostMT(random.get_state())
    p0, p1, p2 = order.copy(), order.copy(), order.copy()
    s0 = np.zeros(n)
    rc = lib.emx_host_split_draws(mt.h, n, C.byref(desc), off, order, 0, p0, p1, p2, s0)
    if rc != 0:
        raise ValueError("cannot draw a proposal plan for this move / split layout")
    random.set_state(mt.get_state())
    return dict(off=off, order=order, p0=p0, p1=p1, p2=p2, s0=s0, uacc=np.full(n, 0.5))
