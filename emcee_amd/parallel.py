"""Walker-sharded stepping across the GPUs of one node (one process per GPU).

Design (DESIGN.md "Multi-GPU"): every rank holds a full replica of the ensemble
(it must: a walker's partner is drawn uniformly from the whole complement) and makes the same
random decisions (same RNG stream / same Philox key).  The slots of each sub-ensemble are
partitioned contiguously over ranks; a rank proposes + evaluates + accepts only its slots,
writes the resulting [row | log_prob | accepted] records into a packed send buffer, and ONE
all-gather per half-step (RCCL over xGMI via torch.distributed) brings every rank's records to
every rank, where a scatter kernel folds them into the local replica before the next split --
the ordering red_blue.py:85,104 requires.  No other collective sits on the data path.

The engine is abstract (`DeviceEngine` drives libemx; the CPU test-suite supplies a NumPy
double) so the protocol itself is covered by world_size-2 gloo tests without a GPU.
"""
import numpy as np

__all__ = ["shard_range", "rows_per_rank", "ShardedStepper", "DeviceEngine", "LocalGroup"]


def shard_range(ns, rank, world):
    """Contiguous slot range of `rank` among `ns` slots (mirrors shard_range in emx.hip)."""
    return ns * rank // world, ns * (rank + 1) // world


def rows_per_rank(nwalkers, world):
    """Records per rank in the exchange buffers (mirrors shard_rows_per_rank in emx.hip)."""
    maxns = (nwalkers + 1) // 2 + 1
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


class DeviceEngine:
    """libemx context as a sharded engine; exchange buffers are torch tensors (RCCL-ready)."""

    def __init__(self, ens, rank, world, torch_device=None):
        import torch
        self.ens = ens
        self.rank, self.world = rank, world
        ens.set_shard(rank, world)
        rows = rows_per_rank(ens.nwalkers, world)
        dev = torch_device if torch_device is not None else torch.device("cuda", torch.cuda.current_device())
        rec = ens.ndim + 2
        self.sendbuf = torch.zeros(rows * rec, dtype=torch.float64, device=dev)
        self.gathered = torch.zeros(world * rows * rec, dtype=torch.float64, device=dev)
        ens.set_shard_buffers(self.sendbuf.data_ptr(), self.gathered.data_ptr(), rows)

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
    def _all_gather(engines):
        n = engines[0].sendbuf.shape[0]
        for e in engines:
            for r, src in enumerate(engines):
                e.gathered[r * n:(r + 1) * n] = src.sendbuf
