"""NumPy stand-in for the libemx sharded engine (TEST DOUBLE, CPU only).

It follows the same protocol as emcee_amd.parallel.DeviceEngine -- plan from the product's
host plan producer, slot ranges from emcee_amd.parallel.shard_range, [row | log_prob |
accepted] records -- and does the arithmetic with the oracle's formulas, so the world_size>1
exchange logic can be checked against a single-rank oracle run without a GPU.
"""
import numpy as np

from emcee_amd.parallel import rows_per_rank, shard_range
from oracle import sampler_oracle as so

from emx_testlib import HostMT, cdf_of, move_desc


class FakeEngine:
    def __init__(self, p0, lp_fn, moves, weights, rng_state, rank, world, make_buffer=np.zeros):
        self.X = np.array(p0, dtype=np.float64, copy=True)
        self.N, self.D = self.X.shape
        self.lp_fn = lp_fn
        self.lp = np.asarray(lp_fn(self.X), dtype=np.float64)
        self.acc = np.zeros(self.N, dtype=bool)
        self.moves = moves
        self.cdf = cdf_of(weights, len(moves))
        self.mt = HostMT(rng_state)
        self.rank, self.world = rank, world
        self.rows = rows_per_rank(self.N, world)
        self.rec = self.D + 2
        self.sendbuf = make_buffer(self.rows * self.rec)
        self.gathered = make_buffer(world * self.rows * self.rec)
        self.chain, self.chain_lp = [], []
        self.acc_count = np.zeros(self.N)
        self.plan = None

    def step_begin(self, store):
        k = self.mt.choice_cdf(self.cdf)
        self.move = self.moves[k]
        self.plan = self.mt.plan(self.N, self.D, move_desc(self.move, self.D))
        self.store = store
        return k, self.move.nsplits

    def _np(self, buf):
        return buf.numpy() if hasattr(buf, "numpy") else buf

    def halfstep(self, split):
        off = self.plan["off"]
        ns = off[split + 1] - off[split]
        lo, hi = shard_range(int(ns), self.rank, self.world)
        sub = {k: (v[off[split] + lo: off[split] + hi] if k != "off" else np.array([0, hi - lo])) for k, v in self.plan.items()}
        idx = sub["order"]
        # propose_planned mutates in place; run it on the owned slots only
        acc = so.propose_planned(self.X, self.lp, self.lp_fn, sub, self.move)
        self.acc[idx] = acc[idx]
        sb = self._np(self.sendbuf).reshape(self.rows, self.rec)
        sb[: hi - lo, : self.D] = self.X[idx]
        sb[: hi - lo, self.D] = self.lp[idx]
        sb[: hi - lo, self.D + 1] = acc[idx]

    def scatter_gathered(self, split):
        off = self.plan["off"]
        ns = int(off[split + 1] - off[split])
        ga = self._np(self.gathered).reshape(self.world, self.rows, self.rec)
        for r in range(self.world):
            if r == self.rank:
                continue
            lo, hi = shard_range(ns, r, self.world)
            idx = self.plan["order"][off[split] + lo: off[split] + hi]
            self.X[idx] = ga[r, : hi - lo, : self.D]
            self.lp[idx] = ga[r, : hi - lo, self.D]
            self.acc[idx] = ga[r, : hi - lo, self.D + 1] != 0

    def step_end(self):
        if self.store:
            self.chain.append(self.X.copy())
            self.chain_lp.append(self.lp.copy())
            self.acc_count += self.acc
