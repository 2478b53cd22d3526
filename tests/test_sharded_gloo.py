"""The N>1 exchange protocol over torch.distributed (gloo, world_size 2 and 3) on CPU.

Each process drives emcee_amd.parallel.ShardedStepper with the NumPy engine double; every
rank must end with the single-rank oracle chain, bit for bit."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, name, nst, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    from emcee_amd.parallel import ShardedStepper
    from fake_engine import FakeEngine
    from helpers import load_golden, rng_from_fixture
    from oracle import cases

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden(name)
    spec = cases.build(name)
    eng = FakeEngine(g["p0"], cases.make_target(spec["desc"]), spec["moves"], spec["weights"],
                     rng_from_fixture(g).get_state(), rank, world,
                     make_buffer=lambda n: torch.zeros(n, dtype=torch.float64))
    st = ShardedStepper(eng, lambda out, inp: dist.all_gather_into_tensor(out, inp))
    st.run(nst, 1, True)
    q.put((rank, np.stack(eng.chain), np.stack(eng.chain_lp), eng.acc_count))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("stretch_50x3_iso", 2), ("mix_de_snooker_128x8_dense", 2),
                                        ("stretch_nsplits3_45x2", 3)])
def test_sharded_protocol_over_gloo(name, world):
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    from helpers import load_golden
    g = load_golden(name)
    nst = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, nst, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, chain, lp, acc in res:
        if "snooker" in name:
            np.testing.assert_allclose(chain, g["chain"][:nst], rtol=1e-12, atol=1e-14)
        else:
            assert np.array_equal(chain, g["chain"][:nst]), "rank %d diverged" % rank
        np.testing.assert_allclose(lp, g["log_prob"][:nst], rtol=1e-12)


def test_shard_ranges_tile_the_slots():
    from emcee_amd.parallel import rows_per_rank, shard_range
    for ns in (1, 2, 7, 16, 32768, 131073):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(ns, r, world)
                cover += list(range(lo, hi)) if ns < 100 else [lo, hi]
                assert hi - lo <= rows_per_rank(2 * ns, world)
            if ns < 100:
                assert cover == list(range(ns))
