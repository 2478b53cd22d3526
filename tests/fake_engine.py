"""NumPy stand-in for the libemx sharded engine (TEST DOUBLE, CPU only).

It follows the same protocol as emcee_amd.parallel.DeviceEngine -- plan from the product's
host plan producer, slot ranges from emcee_amd.parallel.shard_range, [row | log_prob |
accepted] records -- and does the arithmetic with the oracle's formulas, so the world_size>1
exchange logic can be checked against a single-rank oracle run without a GPU.
"""
import numpy as np

from emcee_amd.parallel import block_owner, block_range, pull_capacity, rows_per_rank, shard_range
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


class FakePullEngine(FakeEngine):
    """The pull exchange (emcee_amd.parallel.PullStepper) on the same NumPy double: walker-block
    ownership, [index | row] records for the peers, block all-gather at the end."""

    NPART = {"stretch": 1, "de": 2, "snooker": 3}

    def __init__(self, *a, **kw):
        make_buffer = kw.get("make_buffer", np.zeros)
        super().__init__(*a, **kw)
        self.ndim = self.D
        self.bmax = -(-self.N // self.world)
        capmax = max(pull_capacity(self.N, self.world, m.nsplits, self.NPART[m.kind]) for m in self.moves)
        pairs = self.world * capmax * (self.D + 1)
        self.sendbuf = make_buffer(max(pairs, self.bmax * (self.D + 3)))
        self.gathered = make_buffer(max(pairs, self.world * self.bmax * (self.D + 3)))
        self.lo, self.hi = block_range(self.N, self.rank, self.world)

    def pull_prepare(self, split):
        off = self.plan["off"]
        sl = slice(off[split], off[split + 1])
        npart = self.NPART[self.move.kind]
        cap = pull_capacity(self.N, self.world, self.move.nsplits, npart)
        oi = block_owner(self.plan["order"][sl], self.N, self.world)
        self._mine = np.nonzero(oi == self.rank)[0] + off[split]
        sb = self._np(self.sendbuf)[: self.world * cap * (self.D + 1)].reshape(self.world, cap, self.D + 1)
        sb[:, :, 0] = -1.0
        fill = np.zeros(self.world, dtype=int)
        for j in range(npart):
            pj = self.plan["p%d" % j][sl]
            here = (oi != self.rank) & (block_owner(pj, self.N, self.world) == self.rank)
            for q, row in zip(oi[here], pj[here]):
                assert fill[q] < cap, "pull capacity exceeded"
                sb[q, fill[q], 0] = row
                sb[q, fill[q], 1:] = self.X[row]
                fill[q] += 1
        self._cap = cap
        return cap

    def pull_apply(self, split):
        ga = self._np(self.gathered)[: self.world * self._cap * (self.D + 1)].reshape(self.world, self._cap, self.D + 1)
        for r in range(self.world):
            if r == self.rank:
                continue
            ok = ga[r, :, 0] >= 0
            self.X[ga[r, ok, 0].astype(np.int64)] = ga[r, ok, 1:]
        sub = {k: v[self._mine] for k, v in self.plan.items() if k != "off"}
        sub["off"] = np.array([0, len(self._mine)])
        idx = sub["order"]
        acc = so.propose_planned(self.X, self.lp, self.lp_fn, sub, self.move)
        self.acc[idx] = acc[idx]

    def replica_pack(self):
        sb = self._np(self.sendbuf)[: self.bmax * (self.D + 3)].reshape(self.bmax, self.D + 3)
        n = self.hi - self.lo
        sb[:n, : self.D] = self.X[self.lo: self.hi]
        sb[:n, self.D] = self.lp[self.lo: self.hi]
        sb[:n, self.D + 1] = self.acc[self.lo: self.hi]
        sb[:n, self.D + 2] = self.acc_count[self.lo: self.hi]
        return self.bmax

    def replica_unpack(self):
        ga = self._np(self.gathered)[: self.world * self.bmax * (self.D + 3)].reshape(self.world, self.bmax, self.D + 3)
        for r in range(self.world):
            if r == self.rank:
                continue
            lo, hi = block_range(self.N, r, self.world)
            self.X[lo:hi] = ga[r, : hi - lo, : self.D]
            self.lp[lo:hi] = ga[r, : hi - lo, self.D]
            self.acc[lo:hi] = ga[r, : hi - lo, self.D + 1] != 0
            self.acc_count[lo:hi] = ga[r, : hi - lo, self.D + 2]

    def step_end(self):
        # only the own block of a stored step is meaningful before the replicas are synchronised
        if self.store:
            self.chain.append(self.X.copy())
            self.chain_lp.append(self.lp.copy())
            self.acc_count[self.lo: self.hi] += self.acc[self.lo: self.hi]


class FakeDirectEngine(FakePullEngine):
    """The direct exchange on the NumPy double: every rank's coordinate array and barrier flags live in POSIX shared
    memory (the CPU stand-in for hipIpc-mapped HBM); a half-step reads partner rows from the segment of the rank that
    owns them, and a flag barrier (store the epoch into every peer's flag array, spin on the own one) is the only
    synchronisation between the processes inside a step -- the protocol of emx_direct_halfstep / k_peer_barrier."""

    def __init__(self, *a, **kw):
        from multiprocessing import shared_memory
        tag = kw.pop("tag")
        super().__init__(*a, **kw)
        self._shm_x = shared_memory.SharedMemory(create=True, size=self.N * self.D * 8, name="emx_%s_x%d" % (tag, self.rank))
        self._shm_f = shared_memory.SharedMemory(create=True, size=8 * 8, name="emx_%s_f%d" % (tag, self.rank))
        x = np.ndarray((self.N, self.D), dtype=np.float64, buffer=self._shm_x.buf)
        x[:] = self.X
        self.X = x                                            # the replica itself is what the peers map
        self.flags = np.ndarray(8, dtype=np.int64, buffer=self._shm_f.buf)
        self.flags[:] = 0
        self.epoch = 0
        self.peers, self._attached = {}, []

    # the two calls emcee_amd.parallel.import_direct_peers makes on a DeviceEnsemble
    def direct_export(self):
        h = np.zeros(128, dtype=np.uint8)
        for o, name in ((0, self._shm_x.name), (64, self._shm_f.name)):
            b = name.encode()
            h[o:o + len(b)] = np.frombuffer(b, dtype=np.uint8)
        return h

    def direct_import(self, handles):
        from multiprocessing import shared_memory
        h = np.asarray(handles, dtype=np.uint8).reshape(self.world, 128)
        for q in range(self.world):
            if q == self.rank:
                self.peers[q] = (self.X, self.flags)
                continue
            names = [bytes(h[q, o:o + 64]).rstrip(b"\0").decode() for o in (0, 64)]
            sx, sf = shared_memory.SharedMemory(name=names[0]), shared_memory.SharedMemory(name=names[1])
            self._attached += [sx, sf]
            self.peers[q] = (np.ndarray((self.N, self.D), dtype=np.float64, buffer=sx.buf),
                             np.ndarray(8, dtype=np.int64, buffer=sf.buf))

    def _barrier(self, timeout=60.0):
        import time
        self.epoch += 1
        for q in range(self.world):
            if q != self.rank:
                self.peers[q][1][self.rank] = self.epoch       # "my previous half-step is complete"
        t0 = time.time()
        while any(self.flags[q] < self.epoch for q in range(self.world) if q != self.rank):
            if time.time() - t0 > timeout:
                raise RuntimeError("direct exchange: a peer never reached the barrier")
            time.sleep(0)

    def direct_halfstep(self, split, barrier=True):
        if barrier:
            self._barrier()
        off = self.plan["off"]
        sl = slice(off[split], off[split + 1])
        mine = np.nonzero(block_owner(self.plan["order"][sl], self.N, self.world) == self.rank)[0] + off[split]
        sub = {k: v[mine] for k, v in self.plan.items() if k != "off"}
        sub["off"] = np.array([0, len(mine)])
        # partner rows come from the replica of the rank that owns them (own rows from the own replica)
        view = np.array(self.X, copy=True)
        for j in range(self.NPART[self.move.kind]):
            pj = sub["p%d" % j]
            own = block_owner(pj, self.N, self.world)
            for q in range(self.world):
                if q != self.rank:
                    rows = pj[own == q]
                    view[rows] = self.peers[q][0][rows]
        idx = sub["order"]
        acc = so.propose_planned(view, self.lp, self.lp_fn, sub, self.move)
        self.X[idx] = view[idx]
        self.acc[idx] = acc[idx]

    def close(self):
        self.peers = {}
        x = np.array(self.X, copy=True)
        self.X, self.flags = x, None
        for s in self._attached:
            s.close()
        for s in (self._shm_x, self._shm_f):
            s.close()
            s.unlink()


class FakeLogProbEngine(FakeEngine):
    """The log-prob exchange (emcee_amd.parallel.LogProbStepper) on the NumPy double: proposal, decision and commit
    replicated on every rank, the target evaluated on a share of the proposals, 8 bytes per walker gathered in place."""

    def __init__(self, *a, **kw):
        make_buffer = kw.get("make_buffer", np.zeros)
        super().__init__(*a, **kw)
        self.ndim = self.D
        self.sendbuf = None
        self.gathered = make_buffer(self.N + self.world)
        self.evaluated = 0          # proposals this rank evaluated (the point of the protocol: ~1/world of them)

    def _sub(self, split):
        off = self.plan["off"]
        sub = {k: v[off[split]: off[split + 1]] for k, v in self.plan.items() if k != "off"}
        sub["off"] = np.array([0, off[split + 1] - off[split]])
        return sub

    def logprob_begin(self, split):
        sub = self._sub(split)
        ns = int(sub["off"][1])
        per = -(-ns // self.world)
        box = {}

        def capture(q):               # the oracle's own proposal arithmetic, on copies; nothing is evaluated here
            box["q"] = np.array(q, copy=True)
            return np.zeros(len(q))

        so.propose_planned(self.X.copy(), self.lp.copy(), capture, sub, self.move)
        lo, hi = min(self.rank * per, ns), min((self.rank + 1) * per, ns)
        if hi > lo:
            self._np(self.gathered)[lo:hi] = self.lp_fn(box["q"][lo:hi])
            self.evaluated += hi - lo
        self._ns = ns
        return per

    def logprob_finish(self, split):
        sub = self._sub(split)
        new_lp = np.array(self._np(self.gathered)[: self._ns], copy=True)
        acc = so.propose_planned(self.X, self.lp, lambda q: new_lp, sub, self.move)
        idx = sub["order"]
        self.acc[idx] = acc[idx]


class FakeReplayEngine(FakeEngine):
    """The replay exchange (emcee_amd.parallel.ReplayStepper) on the NumPy double: a rank updates its share of the split and
    publishes its decisions -- the new log-prob of an accepted proposal, NaN otherwise: 8 bytes per walker-update -- and after
    the all-gather recomputes the accepted updates of the others on its own replica (same rows, same plan, same formulas)."""

    def __init__(self, *a, **kw):
        make_buffer = kw.get("make_buffer", np.zeros)
        super().__init__(*a, **kw)
        self.ndim = self.D
        per = rows_per_rank(self.N, self.world, min([2] + [m.nsplits for m in self.moves]))
        self.sendbuf = make_buffer(per)
        self.gathered = make_buffer(per * self.world)
        self.evaluated = 0            # target evaluations on this rank: ~1/world of them, and never for a replayed update
        self.doubles_sent = 0

    def _sub(self, split, lo, hi):
        off = self.plan["off"]
        sub = {k: v[off[split] + lo: off[split] + hi] for k, v in self.plan.items() if k != "off"}
        sub["off"] = np.array([0, hi - lo])
        return sub

    def replay_begin(self, split):
        off = self.plan["off"]
        ns = int(off[split + 1] - off[split])
        rows = -(-ns // self.world)
        lo, hi = shard_range(ns, self.rank, self.world)
        if hi > lo:
            sub = self._sub(split, lo, hi)
            box = {}

            def lp_counted(q):
                box["lp"] = np.asarray(self.lp_fn(q), dtype=np.float64)
                self.evaluated += len(q)
                return box["lp"]

            idx = sub["order"]
            acc = so.propose_planned(self.X, self.lp, lp_counted, sub, self.move)
            self.acc[idx] = acc[idx]
            self._np(self.sendbuf)[: hi - lo] = np.where(acc[idx], box["lp"], np.nan)
        self._ns = ns
        self.doubles_sent += rows
        return rows

    def replay_finish(self, split):
        ns, rows = self._ns, -(-self._ns // self.world)
        ga = self._np(self.gathered)
        for r in range(self.world):
            if r == self.rank:
                continue
            lo, hi = shard_range(ns, r, self.world)
            if hi <= lo:
                continue
            dec = np.array(ga[r * rows: r * rows + hi - lo], copy=True)
            ok = ~np.isnan(dec)
            sub = self._sub(split, lo, hi)
            idx = sub["order"]
            self.acc[idx] = ok
            if not ok.any():
                continue
            hit = {k: (v[ok] if k != "off" else np.array([0, int(ok.sum())])) for k, v in sub.items()}
            # the proposal arithmetic again, on this replica; the decision is the owner's (forced), the log-prob the owner's value
            forced = dict(hit)
            forced["uacc"] = np.zeros(int(ok.sum()))                 # ln 0 = -inf: every replayed proposal is "accepted"
            new_lp = dec[ok]
            with np.errstate(divide="ignore"):
                so.propose_planned(self.X, self.lp, lambda q: new_lp, forced, self.move)
