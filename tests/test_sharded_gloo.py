"""The N>1 exchange protocol over torch.distributed (gloo, world_size 2 and 3) on CPU.

Each process drives emcee_amd.parallel.ShardedStepper with the NumPy engine double; every
rank must end with the single-rank oracle chain, bit for bit."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(mp, target, world, args, timeout=180):
    """`world` spawned processes of target(rank, world, port, *args, q) on a port that is free now; their results.  The
    processes are daemons and are ended if one of them fails or stalls, so a failed rendezvous cannot hold the test
    session at exit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=timeout) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
                if p.is_alive():
                    p.kill()
    return res


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
    res = _run_world(mp, _worker, world, (name, nst,))
    for rank, chain, lp, acc in res:
        if "snooker" in name:
            np.testing.assert_allclose(chain, g["chain"][:nst], rtol=1e-12, atol=1e-14)
        else:
            assert np.array_equal(chain, g["chain"][:nst]), "rank %d diverged" % rank
        np.testing.assert_allclose(lp, g["log_prob"][:nst], rtol=1e-12)


def _pull_worker(rank, world, port, name, nst, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    from emcee_amd.parallel import PullStepper
    from fake_engine import FakePullEngine
    from helpers import load_golden, rng_from_fixture
    from oracle import cases

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden(name)
    spec = cases.build(name)
    eng = FakePullEngine(g["p0"], cases.make_target(spec["desc"]), spec["moves"], spec["weights"],
                         rng_from_fixture(g).get_state(), rank, world,
                         make_buffer=lambda n: torch.zeros(n, dtype=torch.float64))
    st = PullStepper(eng, lambda out, inp: dist.all_to_all_single(out, inp),
                     lambda out, inp: dist.all_gather_into_tensor(out, inp))
    st.run(nst, 1, True)      # ends with the block all-gather
    q.put((rank, eng.lo, eng.hi, np.stack(eng.chain), np.stack(eng.chain_lp), eng.X.copy(), eng.lp.copy(), eng.acc_count.copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("stretch_50x3_iso", 2), ("mix_de_snooker_128x8_dense", 2),
                                        ("stretch_nsplits3_45x2", 3)])
def test_pull_protocol_over_gloo(name, world):
    """Pull exchange over a real all-to-all: every rank's block of the chain, and every replica after the
    final block all-gather, equal the single-rank oracle chain."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    from helpers import load_golden
    g = load_golden(name)
    nst = 6
    res = _run_world(mp, _pull_worker, world, (name, nst,))
    exact = "snooker" not in name
    acc_total = (np.diff(g["chain"][: nst], axis=0, prepend=g["p0"][None]) != 0).any(axis=2).sum(axis=0)
    for rank, lo, hi, chain, lp, x, lpf, acc in res:
        if exact:
            assert np.array_equal(chain[:, lo:hi], g["chain"][:nst, lo:hi]), "rank %d diverged" % rank
            assert np.array_equal(x, g["chain"][nst - 1])
        else:
            np.testing.assert_allclose(chain[:, lo:hi], g["chain"][:nst, lo:hi], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(x, g["chain"][nst - 1], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(lp[:, lo:hi], g["log_prob"][:nst, lo:hi], rtol=1e-12)
        np.testing.assert_allclose(lpf, g["log_prob"][nst - 1], rtol=1e-12)
        assert np.array_equal(acc, acc_total)


def _logprob_worker(rank, world, port, name, nst, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    from emcee_amd.parallel import LogProbStepper
    from fake_engine import FakeLogProbEngine
    from helpers import load_golden, rng_from_fixture
    from oracle import cases

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden(name)
    spec = cases.build(name)
    eng = FakeLogProbEngine(g["p0"], cases.make_target(spec["desc"]), spec["moves"], spec["weights"],
                            rng_from_fixture(g).get_state(), rank, world,
                            make_buffer=lambda n: torch.zeros(n, dtype=torch.float64))
    st = LogProbStepper(eng, lambda out, inp: dist.all_gather_into_tensor(out, inp.clone()))
    st.run(nst, 1, True)
    q.put((rank, np.stack(eng.chain), np.stack(eng.chain_lp), eng.acc_count.copy(), eng.evaluated))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("stretch_50x3_iso", 2), ("mix_de_snooker_128x8_dense", 3),
                                        ("stretch_nsplits3_45x2", 2)])
def test_logprob_protocol_over_gloo(name, world):
    """Log-prob exchange (the reference's pool.map model) over a real in-place all-gather: every rank's replica of the
    chain equals the single-rank oracle chain, and every rank evaluated only its share of the proposals."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    from helpers import load_golden
    g = load_golden(name)
    nst = 6
    res = _run_world(mp, _logprob_worker, world, (name, nst,))
    exact = "snooker" not in name
    N = g["p0"].shape[0]
    total_eval = 0
    for rank, chain, lp, acc, evaluated in res:
        if exact:
            assert np.array_equal(chain, g["chain"][:nst]), "rank %d diverged" % rank
        else:
            np.testing.assert_allclose(chain, g["chain"][:nst], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(lp, g["log_prob"][:nst], rtol=1e-12)
        total_eval += evaluated
        assert evaluated <= nst * (N // world + 4 * 5)          # a share, not everything
    assert total_eval == nst * N                                # every proposal evaluated exactly once


def _replay_worker(rank, world, port, name, nst, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    from emcee_amd.parallel import ReplayStepper
    from fake_engine import FakeReplayEngine
    from helpers import load_golden, rng_from_fixture
    from oracle import cases

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden(name)
    spec = cases.build(name)
    eng = FakeReplayEngine(g["p0"], cases.make_target(spec["desc"]), spec["moves"], spec["weights"],
                           rng_from_fixture(g).get_state(), rank, world,
                           make_buffer=lambda n: torch.zeros(n, dtype=torch.float64))
    st = ReplayStepper(eng, lambda out, inp: dist.all_gather_into_tensor(out, inp.clone()))
    st.run(nst, 1, True)
    q.put((rank, np.stack(eng.chain), np.stack(eng.chain_lp), eng.acc_count.copy(), eng.evaluated, eng.doubles_sent))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("stretch_50x3_iso", 2), ("mix_de_snooker_128x8_dense", 3),
                                        ("stretch_nsplits3_45x2", 2), ("stretch_128x64_dense", 3)])
def test_replay_protocol_over_gloo(name, world):
    """Replay exchange over a real all-gather between processes: what travels is one double per walker-update (the decision),
    every rank's replica of the chain equals the single-rank oracle chain, every proposal is evaluated exactly once across
    the ranks and a replayed update never evaluates the target."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    from helpers import load_golden
    g = load_golden(name)
    nst = 6
    res = _run_world(mp, _replay_worker, world, (name, nst,))
    exact = "snooker" not in name
    N, D = g["p0"].shape
    acc_total = (np.diff(g["chain"][: nst], axis=0, prepend=g["p0"][None]) != 0).any(axis=2).sum(axis=0)
    total_eval = 0
    for rank, chain, lp, acc, evaluated, sent in res:
        if exact:
            assert np.array_equal(chain, g["chain"][:nst]), "rank %d diverged" % rank
        else:
            np.testing.assert_allclose(chain, g["chain"][:nst], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(lp, g["log_prob"][:nst], rtol=1e-12)
        assert np.array_equal(acc, acc_total)
        total_eval += evaluated
        assert sent <= nst * (N // world + 6)                   # 8 bytes per own walker-update, never a row
    assert total_eval == nst * N


def _shared_lp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import emcee_amd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"rows": 0}

    def fn(x):                                   # vectorised, with a blob column
        calls["rows"] += len(x)
        return np.column_stack([-0.5 * np.sum(x * x, axis=1), x[:, 0] + 1.0])

    def fn1(x):                                  # per walker, no blobs
        calls["rows"] += 1
        return -0.5 * float(np.sum(x * x))

    X = np.random.RandomState(4).randn(37, 3)    # 37 rows over 3 ranks: shares of 13, 13, 11
    s = emcee_amd.EnsembleSampler(37, 3, fn, vectorize=True, distributed=True, exchange="logprob")
    lp, blobs = s.compute_log_prob(X)
    rows_vec = calls["rows"]
    calls["rows"] = 0
    s1 = emcee_amd.EnsembleSampler(37, 3, fn1, distributed=True, exchange="logprob")
    lp1, blobs1 = s1.compute_log_prob(X)
    lp_few, _ = s1.compute_log_prob(X[:2])       # fewer rows than ranks: somebody's share is empty
    q.put((rank, lp, blobs, rows_vec, lp1, blobs1, calls["rows"], lp_few))
    dist.barrier()
    dist.destroy_process_group()


def test_python_log_prob_calls_are_shared_out_over_gloo():
    """exchange='logprob' with a Python callable (ensemble.py:_shared_log_prob; the reference's pool.map, ensemble.py:486-496,
    with the ranks of the process group for workers): every rank gets the full vector and blobs, in row order, having called
    the function on its share only.  No GPU involved: compute_log_prob is host code."""
    import torch.multiprocessing as mp
    world = 3
    res = _run_world(mp, _shared_lp_worker, world, ())
    X = np.random.RandomState(4).randn(37, 3)
    want = -0.5 * np.sum(X * X, axis=1)
    shares = {0: 13, 1: 13, 2: 11}
    for rank, lp, blobs, rows_vec, lp1, blobs1, rows_one, lp_few in res:
        assert np.array_equal(lp, want) and np.array_equal(np.ravel(blobs), X[:, 0] + 1.0)
        assert rows_vec == shares[rank]
        assert np.array_equal(lp1, want) and blobs1 is None
        assert rows_one == shares[rank] + (1 if rank < 2 else 0)
        assert np.array_equal(lp_few, want[:2])


def _direct_worker(rank, world, port, name, nst, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from emcee_amd.parallel import import_direct_peers
    from fake_engine import FakeDirectEngine
    from helpers import load_golden, rng_from_fixture
    from oracle import cases

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden(name)
    spec = cases.build(name)
    eng = FakeDirectEngine(g["p0"], cases.make_target(spec["desc"]), spec["moves"], spec["weights"],
                           rng_from_fixture(g).get_state(), rank, world, tag="%d" % port)
    import_direct_peers(eng, dist)             # the product's handle exchange: 128 bytes per rank over the host group
    dist.barrier()
    for _ in range(nst):
        k, S = eng.step_begin(True)
        for split in range(S):
            eng.direct_halfstep(split, barrier=True)      # no collective inside the loop: flags in shared memory only
        eng.step_end()
    eng._barrier()                             # everybody has finished reading before anybody unmaps
    q.put((rank, eng.lo, eng.hi, np.stack(eng.chain), np.stack(eng.chain_lp)))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("stretch_50x3_iso", 2), ("mix_de_snooker_128x8_dense", 2),
                                        ("stretch_nsplits3_45x2", 3)])
def test_direct_protocol_between_processes(name, world):
    """Direct exchange, control plane and hazards, without a GPU: the ranks' replicas and barrier flags are shared-memory
    segments mapped through emcee_amd.parallel.import_direct_peers; inside the step loop the processes meet only at the
    flag barrier.  Every rank's block of the chain equals the single-rank oracle chain."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    from helpers import load_golden
    g = load_golden(name)
    nst = 6
    res = _run_world(mp, _direct_worker, world, (name, nst,))
    for rank, lo, hi, chain, lp in res:
        if "snooker" not in name:
            assert np.array_equal(chain[:, lo:hi], g["chain"][:nst, lo:hi]), "rank %d diverged" % rank
        else:
            np.testing.assert_allclose(chain[:, lo:hi], g["chain"][:nst, lo:hi], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(lp[:, lo:hi], g["log_prob"][:nst, lo:hi], rtol=1e-12)


def test_pull_capacity_mirror_and_bounds():
    """The Python mirror equals the library's capacity; the capacity never exceeds what a pair can need and
    stays within ~15 % of the mean at bench scale."""
    from emcee_amd import _lib
    from emcee_amd.parallel import pull_capacity
    lib = _lib.load()
    for N in (32, 45, 128, 4096, 65536, 8 * 65536):
        for world in (1, 2, 3, 4, 8):
            for S, npart in ((2, 1), (2, 2), (4, 3), (3, 1)):
                c = pull_capacity(N, world, S, npart)
                assert c == lib.emx_host_pull_capacity(N, world, S, npart)
                assert 1 <= c <= max(1, npart * min(-(-N // world), -(-N // S)))
    assert pull_capacity(8 * 65536, 8, 2, 1) < 1.2 * (8 * 65536 / 2 / 64)


def test_pull_requests_fit_the_capacity():
    """Native plans at a realistic size: the per-pair request counts stay far below the capacity."""
    from emcee_amd.parallel import block_owner, pull_capacity
    from emx_testlib import philox_plan, move_desc
    from oracle import sampler_oracle as so
    N, world = 16384, 8
    mv = so.MoveSpec("stretch")
    cap = pull_capacity(N, world, 2, 1)
    worst = 0
    for step in range(6):
        plan = philox_plan(99, step, N, move_desc(mv, 8))
        for split in range(2):
            sl = slice(plan["off"][split], plan["off"][split + 1])
            oi = block_owner(plan["order"][sl], N, world)
            oj = block_owner(plan["p0"][sl], N, world)
            cnt = np.zeros((world, world), dtype=int)
            np.add.at(cnt, (oj, oi), 1)
            np.fill_diagonal(cnt, 0)
            worst = max(worst, cnt.max())
    mean = N / 2 / world / world
    assert worst < mean + 5 * np.sqrt(mean) < cap


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
