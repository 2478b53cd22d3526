"""Walker-sharded stepping across the GPUs of one node (one process per GPU).

Design (DESIGN.md "Multi-GPU"): every rank holds a full replica of the ensemble
(it must: a walker's partner is drawn uniformly from the whole complement) and makes the same
random decisions (same RNG stream / same Philox key).  The slots of each sub-ensemble are
partitioned contiguously over ranks; a rank proposes + evaluates + accepts only its slots,
writes the resulting [row | log_prob | accepted] records into a packed send buffer, and ONE
all-gather per half-step (RCCL over xGMI via torch.distributed) brings every rank's records to
every rank, where a scatter kernel folds them into the local replica before the next split --
the ordering red_blue.py:85,104 requires.  No other collective sits on the data path.

A third protocol, the *direct* exchange (include/emx.h "direct exchange"), keeps the pull exchange's walker-block
ownership but sends nothing: the peers' coordinate arrays are mapped (IPC) and the half-step kernel reads each partner
row from its owner's HBM over xGMI, with a one-wave device-side barrier between half-steps (`attach_direct_peers`,
`import_direct_peers`; `emx_run` drives it).

A second protocol, the *pull* exchange (`PullStepper`, include/emx.h "pull exchange"), moves only
the rows a half-step reads: rank r owns the walker block [N r / G, N (r+1) / G); because the RNG
plan is replicated each rank knows which of its rows its peers' walkers picked as partners, packs
them as [index | row] records, and ONE all-to-all per half-step delivers them (1/G of the
all-gather's volume per rank).  Replicas are re-synchronised (one all-gather of the blocks) only
when somebody needs the whole ensemble.

The engine is abstract (`DeviceEngine` drives libemx; the CPU test-suite supplies a NumPy
double) so the protocols themselves are covered by world_size-2 gloo tests without a GPU.
"""
import numpy as np

__all__ = ["shard_range", "rows_per_rank", "ShardedStepper", "PullStepper", "ReplayStepper", "DeviceEngine", "LocalGroup",
           "block_range", "block_owner", "pull_capacity", "attach_direct_peers", "import_direct_peers"]


def shard_range(ns, rank, world):
    """Contiguous slot range of `rank` among `ns` slots (mirrors shard_range in emx.hip)."""
    return ns * rank // world, ns * (rank + 1) // world


def rows_per_rank(nwalkers, world, min_nsplits=2):
    """Records per rank in the exchange buffers (mirrors shard_rows_per_rank in emx.hip):
    a rank's share of the largest sub-ensemble; ``min_nsplits=1`` when a Gaussian move is installed
    (its single split is the whole ensemble)."""
    maxns = -(-nwalkers // min_nsplits) + 1
    return (maxns + world - 1) // world + 1


class ShardedStepper:
    """Drives one engine per rank through sharded steps.

    engine API: step_begin(store) -> (move, nsplits); halfstep(split); scatter_gathered(split);
    step_end(); attributes sendbuf / gathered (buffers the collective understands).
    all_gather(gathered, sendbuf): every rank's sendbuf, concatenated in rank order.
    """

    def __init__(self, engine, all_gather):
        self.engine = engine
        self.all_gather = all_gather

    def step(self, store=False):
        e = self.engine
        move, nsplits = e.step_begin(store)
        for split in range(nsplits):
            e.halfstep(split)                         # own slots only; fills sendbuf
            self.all_gather(e.gathered, e.sendbuf)    # the one exchange per half-step
            e.scatter_gathered(split)                 # other ranks' records -> local replica
        e.step_end()
        return move

    def run(self, nsteps, thin_by=1, store=False):
        i = 0
        total = nsteps * thin_by
        hint = getattr(self.engine, "set_prep_hint", None)
        for _ in range(nsteps):
            for _ in range(thin_by):
                if hint is not None:
                    hint(total - i)            # lets the native plan kernel batch the remaining steps
                self.step(store and (i + 1) % thin_by == 0)
                i += 1
        if hint is not None:
            hint(1)


def block_range(nwalkers, rank, world):
    """Walkers owned by `rank` under the pull exchange (mirrors emx_own_walkers)."""
    return nwalkers * rank // world, nwalkers * (rank + 1) // world


def block_owner(w, nwalkers, world):
    """Rank owning walker(s) `w` (mirrors block_owner in emx_kernels.hpp)."""
    return ((np.asarray(w, dtype=np.int64) + 1) * world - 1) // nwalkers


def pull_capacity(nwalkers, world, nsplits, npart):
    """Records per (source, destination) pair of one half-step (mirrors pull_capacity in emx.hip):
    mean + 8 sigma + 64 of the partner rows one rank's walkers pick inside another rank's block."""
    import math
    if world <= 1:
        return 1
    bmax = -(-nwalkers // world)
    nsmax = -(-nwalkers // nsplits)
    hard = npart * min(bmax, nsmax)
    mean = float(npart) * float(nwalkers) / nsplits / world / world
    cap = int(math.ceil(mean + 8.0 * math.sqrt(mean) + 64.0))
    return max(1, min(cap, hard))


class PullStepper:
    """Drives one engine per rank through sharded steps of the pull exchange.

    engine API: step_begin(store) -> (move, nsplits); pull_prepare(split) -> records per peer;
    pull_apply(split); step_end(); replica_pack() -> records per rank; replica_unpack();
    attributes sendbuf / gathered (flat float64 buffers), world, ndim.
    all_to_all(out, inp): equal blocks, block q of `inp` to rank q, landing in block `rank` of its `out`.
    all_gather(out, inp): every rank's `inp`, concatenated in rank order.
    """

    def __init__(self, engine, all_to_all, all_gather):
        self.engine = engine
        self.all_to_all = all_to_all
        self.all_gather = all_gather

    def step(self, store=False):
        e = self.engine
        move, nsplits = e.step_begin(store)
        for split in range(nsplits):
            n = e.world * e.pull_prepare(split) * (e.ndim + 1)     # own rows the peers will read
            self.all_to_all(e.gathered[:n], e.sendbuf[:n])         # the one exchange per half-step
            e.pull_apply(split)                                    # fold in, update the walkers owned here
        e.step_end()
        return move

    def sync_replicas(self):
        e = self.engine
        n = e.replica_pack() * (e.ndim + 3)
        self.all_gather(e.gathered[:e.world * n], e.sendbuf[:n])
        e.replica_unpack()

    def run(self, nsteps, thin_by=1, store=False):
        i = 0
        total = nsteps * thin_by
        hint = getattr(self.engine, "set_prep_hint", None)
        for _ in range(nsteps):
            for _ in range(thin_by):
                if hint is not None:
                    hint(total - i)
                self.step(store and (i + 1) % thin_by == 0)
                i += 1
        if hint is not None:
            hint(1)
        self.sync_replicas()


class LogProbStepper:
    """Drives one engine per rank through steps of the log-prob exchange (the reference's pool.map model,
    ensemble.py:486-496): proposal, decision and commit replicated, the log-prob evaluations shared out.

    engine API: step_begin(store) -> (move, nsplits); logprob_begin(split) -> doubles per rank; logprob_finish(split);
    step_end(); attributes gathered (flat float64 buffer, gathered in place), rank, world.
    all_gather(out, inp): every rank's `inp`, concatenated in rank order (`inp` is block `rank` of `out`).
    """

    def __init__(self, engine, all_gather):
        self.engine = engine
        self.all_gather = all_gather

    def step(self, store=False):
        e = self.engine
        move, nsplits = e.step_begin(store)
        for split in range(nsplits):
            per = e.logprob_begin(split)                      # 8 bytes per walker-update travel, never a coordinate
            if per > 0:
                self.all_gather(e.gathered[:e.world * per], e.gathered[e.rank * per:(e.rank + 1) * per])
            e.logprob_finish(split)
        e.step_end()
        return move

    def run(self, nsteps, thin_by=1, store=False):
        i = 0
        total = nsteps * thin_by
        hint = getattr(self.engine, "set_prep_hint", None)
        for _ in range(nsteps):
            for _ in range(thin_by):
                if hint is not None:
                    hint(total - i)
                self.step(store and (i + 1) % thin_by == 0)
                i += 1
        if hint is not None:
            hint(1)


class ReplayStepper:
    """Drives one engine per rank through steps of the replay exchange (include/emx.h "replay exchange"): every rank updates its
    share of each split and publishes its DECISIONS (the new log-prob of an accepted proposal, NaN otherwise: 8 bytes per
    walker-update); after the all-gather every rank recomputes the accepted updates of the others on its own replica.

    engine API: step_begin(store) -> (move, nsplits); replay_begin(split) -> doubles per rank; replay_finish(split); step_end();
    attributes sendbuf / gathered (flat float64 buffers), world.
    all_gather(out, inp): every rank's `inp`, concatenated in rank order."""

    def __init__(self, engine, all_gather):
        self.engine = engine
        self.all_gather = all_gather

    def step(self, store=False):
        e = self.engine
        move, nsplits = e.step_begin(store)
        for split in range(nsplits):
            rows = e.replay_begin(split)
            if rows > 0 and e.world > 1:
                self.all_gather(e.gathered[:e.world * rows], e.sendbuf[:rows])
            e.replay_finish(split)
        e.step_end()
        return move

    def run(self, nsteps, thin_by=1, store=False):
        i = 0
        total = nsteps * thin_by
        hint = getattr(self.engine, "set_prep_hint", None)
        for _ in range(nsteps):
            for _ in range(thin_by):
                if hint is not None:
                    hint(total - i)
                self.step(store and (i + 1) % thin_by == 0)
                i += 1
        if hint is not None:
            hint(1)


class _DevView(object):
    """__cuda_array_interface__ carrier for a library-owned device buffer of float64"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2,
                                         "strides": None}


def _wrap_device_buffer(ens, which, dev):
    import torch
    ptr, nbytes = ens.device_ptr(which)
    return torch.as_tensor(_DevView(ptr, nbytes // 8), device=dev)


def attach_direct_peers(ensembles, which=0):
    """Logical ranks of ONE process (tests, several contexts on one GPU): hand every context the others' coordinate
    arrays (``which=0``: direct exchange) or receive buffers (``which=3``: device-side replay exchange) and barrier flags as
    plain device pointers.  Between processes the same is done with IPC handles:
    ``handles = all_gather(ens.direct_export()); ens.direct_import(handles)``."""
    xs = [e.device_ptr(which)[0] for e in ensembles]
    fs = [e.device_ptr(8)[0] for e in ensembles]
    for e in ensembles:
        e.direct_attach(xs, fs)


def import_direct_peers(ens, dist):
    """One process per GPU: all-gather the 128-byte IPC handles over the (host) process group and map the peers."""
    mine = ens.direct_export()
    every = [None] * dist.get_world_size()
    dist.all_gather_object(every, mine.tobytes())
    ens.direct_import(np.frombuffer(b"".join(every), dtype=np.uint8))


class DeviceEngine:
    """libemx context as a sharded engine; exchange buffers are torch tensors (RCCL-ready)."""

    def __init__(self, ens, rank, world, torch_device=None, exchange="allgather"):
        import torch
        self.ens = ens
        self.rank, self.world = rank, world
        self.ndim = ens.ndim
        dev = torch_device if torch_device is not None else torch.device("cuda", torch.cuda.current_device())
        if exchange == "direct":
            # partner rows are read in place from the peers' HBM; the only buffers are the library's own for the replica
            # re-synchronisation, wrapped here (zero copy) so that a torch collective can move them
            ens.set_exchange("direct")
            ens.set_shard(rank, world)
            self.sendbuf = _wrap_device_buffer(ens, 2, dev)
            self.gathered = _wrap_device_buffer(ens, 3, dev)
            return
        if exchange == "logprob":
            ens.set_exchange("logprob")
            ens.set_shard(rank, world)
            self.sendbuf = None
            self.gathered = _wrap_device_buffer(ens, 3, dev)       # the library's buffer, gathered in place
            return
        if exchange == "replay":
            ens.set_exchange("replay")
            ens.set_shard(rank, world)
            self.sendbuf = _wrap_device_buffer(ens, 2, dev)        # decisions of the own slots
            self.gathered = _wrap_device_buffer(ens, 3, dev)       # ... of every rank
            return
        if exchange == "pull":
            ens.set_exchange("pull")
            ens.set_shard(rank, world)
            ns, nr = ens.exchange_layout()
            self.sendbuf = torch.zeros(ns, dtype=torch.float64, device=dev)
            self.gathered = torch.zeros(nr, dtype=torch.float64, device=dev)
            ens.set_exchange_buffers(self.sendbuf.data_ptr(), ns, self.gathered.data_ptr(), nr)
            return
        ens.set_shard(rank, world)
        rows = rows_per_rank(ens.nwalkers, world, min([2] + [int(m.nsplits) for m in getattr(ens, "_moves", [])]))
        rec = ens.ndim + 2
        self.sendbuf = torch.zeros(rows * rec, dtype=torch.float64, device=dev)
        self.gathered = torch.zeros(world * rows * rec, dtype=torch.float64, device=dev)
        ens.set_shard_buffers(self.sendbuf.data_ptr(), self.gathered.data_ptr(), rows)

    def pull_prepare(self, split):
        return self.ens.pull_prepare(split)

    def pull_apply(self, split):
        self.ens.pull_apply(split)

    def logprob_begin(self, split):
        return self.ens.logprob_begin(split)

    def logprob_finish(self, split):
        self.ens.logprob_finish(split)

    def replay_begin(self, split):
        return self.ens.replay_begin(split)

    def replay_finish(self, split):
        self.ens.replay_finish(split)

    def replica_pack(self):
        return self.ens.replica_pack()

    def replica_unpack(self):
        self.ens.replica_unpack()

    def set_prep_hint(self, n):
        self.ens.set_tuning("prep_hint", n)

    def step_begin(self, store):
        return self.ens.step_begin(store)

    def halfstep(self, split):
        self.ens.halfstep(split)

    def scatter_gathered(self, split):
        self.ens.scatter_gathered(split)

    def step_end(self):
        self.ens.step_end()


class LocalGroup:
    """In-process stand-in for the collective: `world` logical ranks living in one process
    (used to exercise the sharded kernels on a single GPU, and by the CPU tests)."""

    def __init__(self, world):
        self.world = world
        self.pending = {}

    def run_step(self, steppers, store=False):
        """Advance all logical ranks one step in lock-step with a concatenating all-gather."""
        engines = [s.engine for s in steppers]
        res = [e.step_begin(store) for e in engines]
        assert all(r == res[0] for r in res), "ranks disagree on the move: RNG streams diverged"
        nsplits = res[0][1]
        for split in range(nsplits):
            for e in engines:
                e.halfstep(split)
            self._all_gather(engines)
            for e in engines:
                e.scatter_gathered(split)
        for e in engines:
            e.step_end()
        return res[0][0]

    @staticmethod
    def _all_to_all(engines, n):
        """block q (n doubles) of rank r's sendbuf -> block r of rank q's gathered"""
        for r, src in enumerate(engines):
            for q, dst in enumerate(engines):
                dst.gathered[r * n:(r + 1) * n] = src.sendbuf[q * n:(q + 1) * n]

    @staticmethod
    def _all_gather_n(engines, n):
        for e in engines:
            for r, src in enumerate(engines):
                e.gathered[r * n:(r + 1) * n] = src.sendbuf[:n]

    @staticmethod
    def _all_gather(engines):
        n = engines[0].sendbuf.shape[0]
        for e in engines:
            for r, src in enumerate(engines):
                e.gathered[r * n:(r + 1) * n] = src.sendbuf
