"""Thin test-side wrappers over the C ABI (no oracle imports here)."""
import ctypes as C

import numpy as np

from emcee_amd import _lib


def move_desc(m, ndim):
    """oracle MoveSpec -> C struct (g0 resolved like moves/de.py:33-38)."""
    if m.kind == "gaussian":
        mode = {"vector": 0, "random": 1, "sequential": 2}[m.mode]
        iso = np.ndim(m.cov) == 0
        lf = 0.0 if m.factor is None else float(np.log(m.factor))
        return _lib.MoveDesc(_lib.MOVE_GAUSS, 1, 0, mode, 0.0 if m.factor is None else 1.0,
                             float(np.sqrt(m.cov)) if iso else 0.0, lf, float(m.index))
    kind = {"stretch": 0, "de": 1, "snooker": 2}[m.kind]
    g0 = m.gamma0 if m.gamma0 is not None else 2.38 / np.sqrt(2 * ndim)
    return _lib.MoveDesc(kind, m.nsplits, int(bool(m.randomize_split)), 0, float(m.a), float(m.sigma), float(g0),
                         float(m.gammas))


class HostMT:
    def __init__(self, state):
        self.lib = _lib.load()
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        self.h = self.lib.emx_mt_create(key, int(state[2]), int(state[3]), float(state[4]))

    def __del__(self):
        try:
            self.lib.emx_mt_destroy(self.h)
        except Exception:
            pass

    def get_state(self):
        key = np.empty(624, dtype=np.uint32)
        pos, hg, cached = C.c_int32(), C.c_int32(), C.c_double()
        self.lib.emx_mt_get_state(self.h, key, C.byref(pos), C.byref(hg), C.byref(cached))
        return ("MT19937", key, pos.value, hg.value, cached.value)

    def random_sample(self, n):
        out = np.empty(n)
        self.lib.emx_mt_random_sample(self.h, n, out)
        return out

    def randint(self, bound, n):
        out = np.empty(n, dtype=np.int64)
        self.lib.emx_mt_randint(self.h, bound, n, out)
        return out

    def randn(self, n):
        out = np.empty(n)
        self.lib.emx_mt_randn(self.h, n, out)
        return out

    def shuffle_labels(self, n, S):
        out = np.empty(n, dtype=np.int32)
        self.lib.emx_mt_shuffle_labels(self.h, n, S, out)
        return out

    def choice_cdf(self, cdf):
        cdf = np.ascontiguousarray(cdf, dtype=np.float64)
        return self.lib.emx_mt_choice_cdf(self.h, cdf, len(cdf))

    def plan(self, N, D, md):
        S = md.nsplits
        off = np.zeros(S + 1, dtype=np.int32)
        order, p0, p1, p2 = (np.empty(N, dtype=np.int32) for _ in range(4))
        s0, uacc = np.empty(N), np.empty(N)
        rc = self.lib.emx_host_plan_mt(self.h, N, D, C.byref(md), off, order, p0, p1, p2, s0, uacc)
        assert rc == 0
        return dict(off=off, order=order, p0=p0, p1=p1, p2=p2, s0=s0, uacc=uacc)


def philox_plan(seed, step, N, md):
    lib = _lib.load()
    S = md.nsplits
    off = np.zeros(S + 1, dtype=np.int32)
    order, p0, p1, p2 = (np.empty(N, dtype=np.int32) for _ in range(4))
    s0, uacc = np.empty(N), np.empty(N)
    rc = lib.emx_host_plan_philox(seed, step, N, C.byref(md), off, order, p0, p1, p2, s0, uacc)
    assert rc == 0
    return dict(off=off, order=order, p0=p0, p1=p1, p2=p2, s0=s0, uacc=uacc)


def cdf_of(weights, n):
    """ensemble.py:128-129 + RandomState.choice's cdf."""
    w = np.ones(n) if weights is None else np.atleast_1d(weights).astype(float)
    w = w / np.sum(w)
    cdf = w.cumsum()
    cdf /= cdf[-1]
    return cdf


def plan_from_trace(step_trace, move, ndim):
    """Expected plan (walker-resolved) from the oracle's per-split trace of one step."""
    labels = step_trace[0]["labels"]
    S = move.nsplits
    sets = [np.nonzero(labels == j)[0] for j in range(S)]
    order = np.concatenate(sets).astype(np.int32)
    off = np.concatenate([[0], np.cumsum([len(s) for s in sets])]).astype(np.int32)
    N = len(labels)
    p0, p1, p2 = order.copy(), order.copy(), order.copy()
    s0, uacc = np.zeros(N), np.zeros(N)
    for tr in step_trace:
        sp = tr["split"]
        sl = slice(off[sp], off[sp + 1])
        csets = sets[:sp] + sets[sp + 1:]
        comp = np.concatenate(csets)
        if move.kind == "stretch":
            p0[sl] = comp[tr["rint"]]
            s0[sl] = tr["zz"]
        elif move.kind == "de":
            g0 = move.gamma0 if move.gamma0 is not None else 2.38 / np.sqrt(2 * ndim)
            p0[sl] = comp[tr["first"]]
            p1[sl] = comp[tr["second"]]
            s0[sl] = g0 * (1 + move.sigma * tr["gauss"])
        else:
            w = np.stack([csets[j][tr["picks"][:, j]] for j in range(3)], axis=1)
            wp = np.take_along_axis(w, tr["perm"], axis=1)
            p0[sl], p1[sl], p2[sl] = wp[:, 0], wp[:, 1], wp[:, 2]
        uacc[sl] = tr["u_acc"]
    return dict(off=off, order=order, p0=p0, p1=p1, p2=p2, s0=s0, uacc=uacc)
