"""Sharded path on ONE GPU: `world` logical ranks (one libemx context each) in one process,
exchanging through an in-process all-gather.  Every rank's replica and chain must equal the
single-rank run bit for bit (the kernels make identical decisions; only the work is split)."""
import numpy as np
import pytest

from emcee_amd import _lib
from emcee_amd.parallel import DeviceEngine, LocalGroup, ShardedStepper
from oracle import cases

from emx_testlib import cdf_of, move_desc
from helpers import load_golden, rng_from_fixture
from test_gpu_parity import make_ens

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,world,rng", [
    ("c1_stretch_32x5_iso", 2, "mt"), ("stretch_50x3_iso", 3, "mt"), ("stretch_128x64_dense", 2, "mt"),
    ("stretch_128x64_dense", 4, "philox"), ("mix_de_snooker_128x8_dense", 2, "mt"),
    ("mix_de_snooker_128x8_dense", 3, "philox"), ("stretch_128x8_rosen", 8, "philox"),
    ("stretch_nsplits3_45x2", 2, "mt"), ("mix_stretch_gauss_32x3", 2, "mt"), ("gauss_diag_random_factor_30x4", 3, "philox"),
    ("stretch_48x130_dense", 2, "mt"), ("mix_de_snooker_40x113_dense", 3, "philox"),
])
def test_logical_ranks_equal_single_rank(name, world, rng):
    import torch
    g = load_golden(name)
    spec = cases.build(name)
    nst = min(8, spec["nsteps"])

    def setup(ens):
        if rng == "mt":
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(rng_from_fixture(g).get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(777, 0)
        ens.chain_config(nst)

    ref = make_ens(spec, g["p0"])
    setup(ref)
    ref.run(nst, 1, True)
    ref_chain, ref_lp, ref_acc = ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst), ref.accepted_counts()
    if rng == "mt" and not any(m.kind == "snooker" for m in spec["moves"]):
        assert np.array_equal(ref_chain, g["chain"][:nst])

    group = LocalGroup(world)
    steppers = []
    for r in range(world):
        ens = make_ens(spec, g["p0"])
        setup(ens)
        eng = DeviceEngine(ens, r, world, torch.device("cuda", 0))
        steppers.append(ShardedStepper(eng, None))
    for _ in range(nst):
        _run_step(group, steppers)
    for s in steppers:
        ens = s.engine.ens
        assert ens.status() == 0
        assert np.array_equal(ens.chain_read(0, 0, nst), ref_chain)
        assert np.array_equal(ens.chain_read(1, 0, nst), ref_lp)
        assert np.array_equal(ens.accepted_counts(), ref_acc)
        x, lp = ens.get_state()
        assert np.array_equal(x, ref_chain[-1]) and np.array_equal(lp, ref_lp[-1])
        ens.close()
    ref.close()


def _run_step(group, steppers):
    """LocalGroup.run_step with explicit stream syncs: the contexts run on their own streams
    while the in-process all-gather runs on torch's."""
    import torch
    engines = [s.engine for s in steppers]
    res = [e.step_begin(True) for e in engines]
    assert all(r == res[0] for r in res)
    for split in range(res[0][1]):
        for e in engines:
            e.halfstep(split)
        for e in engines:
            e.ens.sync()
        group._all_gather(engines)
        torch.cuda.synchronize()
        for e in engines:
            e.scatter_gathered(split)
    for e in engines:
        e.step_end()


@pytest.mark.parametrize("name,world,rng", [
    ("c1_stretch_32x5_iso", 2, "mt"), ("stretch_50x3_iso", 3, "mt"), ("stretch_128x64_dense", 2, "mt"),
    ("stretch_128x64_dense", 4, "philox"), ("mix_de_snooker_128x8_dense", 2, "mt"),
    ("mix_de_snooker_128x8_dense", 3, "philox"), ("stretch_128x8_rosen", 8, "philox"),
    ("stretch_nsplits3_45x2", 2, "mt"), ("stretch_50x3_iso", 7, "philox"),
    ("mix_stretch_gauss_32x3", 2, "mt"), ("gauss_diag_random_factor_30x4", 3, "philox"),
    ("stretch_48x130_dense", 3, "philox"), ("mix_de_snooker_40x113_dense", 2, "mt"),
])
def test_pull_exchange_equals_single_rank(name, world, rng):
    """Pull exchange (walker-block ownership, partner rows only): each logical rank's block of the chain,
    and every replica after the block all-gather, must equal the single-rank run bit for bit."""
    import torch
    from emcee_amd.parallel import block_range
    g = load_golden(name)
    spec = cases.build(name)
    nst = min(8, spec["nsteps"])

    def setup(ens):
        if rng == "mt":
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(rng_from_fixture(g).get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(777, 0)
        ens.chain_config(nst)

    ref = make_ens(spec, g["p0"])
    setup(ref)
    ref.run(nst, 1, True)
    ref_chain, ref_lp, ref_acc = ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst), ref.accepted_counts()
    ref.close()

    engines = []
    for r in range(world):
        ens = make_ens(spec, g["p0"])
        setup(ens)
        engines.append(DeviceEngine(ens, r, world, torch.device("cuda", 0), exchange="pull"))
    nd = engines[0].ndim

    def sync():
        for e in engines:
            e.ens.sync()
        torch.cuda.synchronize()

    for _ in range(nst):
        res = [e.step_begin(True) for e in engines]
        assert all(r == res[0] for r in res)
        for split in range(res[0][1]):
            caps = [e.pull_prepare(split) for e in engines]
            assert len(set(caps)) == 1
            sync()
            LocalGroup._all_to_all(engines, caps[0] * (nd + 1))
            sync()
            for e in engines:
                e.pull_apply(split)
        for e in engines:
            e.step_end()
    # blocks first (replicas are not synchronised yet) ...
    for r, e in enumerate(engines):
        lo, hi = block_range(ref_chain.shape[1], r, world)
        assert e.ens.own_walkers() == (lo, hi)
        assert e.ens.status() == 0
        assert np.array_equal(e.ens.chain_read(0, 0, nst)[:, lo:hi], ref_chain[:, lo:hi])
        assert np.array_equal(e.ens.chain_read(1, 0, nst)[:, lo:hi], ref_lp[:, lo:hi])
        assert np.array_equal(e.ens.accepted_counts()[lo:hi], ref_acc[lo:hi])
    # ... then the block all-gather
    per = [e.replica_pack() for e in engines]
    assert len(set(per)) == 1
    sync()
    LocalGroup._all_gather_n(engines, per[0] * (nd + 3))
    sync()
    for e in engines:
        e.replica_unpack()
        x, lp = e.ens.get_state()
        assert np.array_equal(x, ref_chain[-1]) and np.array_equal(lp, ref_lp[-1])
        assert np.array_equal(e.ens.accepted_counts(), ref_acc)
        e.ens.close()


DIRECT_CASES = [
    ("c1_stretch_32x5_iso", 2, "mt"), ("stretch_50x3_iso", 3, "mt"), ("stretch_128x64_dense", 2, "mt"),
    ("stretch_128x64_dense", 4, "philox"), ("mix_de_snooker_128x8_dense", 2, "mt"),
    ("mix_de_snooker_128x8_dense", 3, "philox"), ("stretch_128x8_rosen", 8, "philox"),
    ("stretch_nsplits3_45x2", 2, "mt"), ("stretch_50x3_iso", 7, "philox"),
    ("mix_stretch_gauss_32x3", 2, "mt"), ("gauss_diag_random_factor_30x4", 3, "philox"),
    ("stretch_48x130_dense", 2, "philox"), ("mix_de_snooker_40x113_dense", 2, "mt"),
]


def _direct_setup(name, world, rng, nst):
    import torch
    from emcee_amd.parallel import attach_direct_peers
    g = load_golden(name)
    spec = cases.build(name)

    def setup(ens):
        if rng == "mt":
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(rng_from_fixture(g).get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(777, 0)
        ens.chain_config(nst)

    ref = make_ens(spec, g["p0"])
    setup(ref)
    ref.run(nst, 1, True)
    out = (ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst), ref.accepted_counts())
    ref.close()
    engines = []
    for r in range(world):
        ens = make_ens(spec, g["p0"])
        setup(ens)
        engines.append(DeviceEngine(ens, r, world, torch.device("cuda", 0), exchange="direct"))
    attach_direct_peers([e.ens for e in engines])
    return engines, out


def _direct_check(engines, out, nst, world):
    import torch
    from emcee_amd.parallel import block_range
    ref_chain, ref_lp, ref_acc = out
    nd = engines[0].ndim
    statuses = [e.ens.status() for e in engines]
    for r, e in enumerate(engines):
        lo, hi = block_range(ref_chain.shape[1], r, world)
        assert statuses[r] == 0
        assert np.array_equal(e.ens.chain_read(0, 0, nst)[:, lo:hi], ref_chain[:, lo:hi])
        assert np.array_equal(e.ens.chain_read(1, 0, nst)[:, lo:hi], ref_lp[:, lo:hi])
        assert np.array_equal(e.ens.accepted_counts()[lo:hi], ref_acc[lo:hi])
    per = [e.replica_pack() for e in engines]
    assert len(set(per)) == 1
    for e in engines:
        e.ens.sync()
    torch.cuda.synchronize()
    LocalGroup._all_gather_n(engines, per[0] * (nd + 3))
    torch.cuda.synchronize()
    for e in engines:
        e.replica_unpack()
        x, lp = e.ens.get_state()
        assert np.array_equal(x, ref_chain[-1]) and np.array_equal(lp, ref_lp[-1])
        assert np.array_equal(e.ens.accepted_counts(), ref_acc)
        e.ens.close()


@pytest.mark.parametrize("name,world,rng", DIRECT_CASES)
def test_direct_exchange_equals_single_rank(name, world, rng):
    """Direct exchange: every logical rank owns a walker block and its half-step kernel reads partner rows straight from
    the other contexts' coordinate arrays (between GPUs: the peers' HBM over xGMI).  Host-ordered here (stream syncs
    between half-steps stand in for the device-side barrier).  Each rank's block of the chain, and every replica after the
    block all-gather, must equal the single-rank run bit for bit."""
    nst = min(8, cases.build(name)["nsteps"])
    engines, out = _direct_setup(name, world, rng, nst)
    for _ in range(nst):
        res = [e.step_begin(True) for e in engines]
        assert all(r == res[0] for r in res)
        for split in range(res[0][1]):
            for e in engines:
                e.ens.direct_halfstep(split, barrier=False)
            for e in engines:
                e.ens.sync()
        for e in engines:
            e.step_end()
    _direct_check(engines, out, nst, world)


def _device_side_processes(world, todo, port):
    """`world` processes, one rank each (tests/workers/device_side_worker.py), all on cuda:0; every case of `todo` must report OK
    on every rank.  A barrier that is not met is a failure: ranks in processes of their own cannot share a hardware queue."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["EMX_DS_CASES"] = ",".join("%s:%s:%s" % c for c in todo)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "workers", "device_side_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-6000:]
    assert "MISMATCH" not in out, out[-6000:]
    import re
    for c in todo:           # (the ranks' lines may run into each other on the shared pipe: count matches, not lines)
        tag = re.escape("DEVICE_SIDE %s %s %s world %d" % (c + (world,))) + r" rank \d+ status 0 rows \[\d+, \d+\) OK"
        assert len(re.findall(tag, out)) == world, out[-6000:]


def test_direct_exchange_device_side_barrier():
    """The direct exchange with the ranks ordered by the one-wave barrier kernel alone: no host synchronisation between
    half-steps; the barrier kernels of the ranks meet on the device (flag store into the peer's array, spin on the own one;
    bounded: a barrier that is never met raises status bit 3 instead of hanging).  One PROCESS per rank, as deployed -- until
    round 4 these ran as contexts of one process, where the runtime may put two ranks' streams into one hardware queue (a
    spinning barrier kernel then sits in front of the kernels it waits for) and a time-out had to be forgiven as a skip; which
    queue a stream gets depends on every stream the process ever created, so the whole suite's history decided.  Between
    processes the time-out is a failure.  Split k + 1 sees split k's commits: reference moves/red_blue.py:85,104."""
    _device_side_processes(2, [("direct", "stretch_128x64_dense", "philox"), ("direct", "mix_de_snooker_128x8_dense", "mt")], 29681)


@pytest.mark.parametrize("name,world,rng", [
    ("c1_stretch_32x5_iso", 2, "mt"), ("stretch_50x3_iso", 3, "mt"), ("stretch_128x64_dense", 2, "mt"),
    ("stretch_128x64_dense", 5, "philox"), ("mix_de_snooker_128x8_dense", 3, "philox"), ("mix_de_snooker_128x8_dense", 2, "mt"),
    ("stretch_128x8_rosen", 8, "philox"), ("stretch_nsplits3_45x2", 2, "mt"), ("stretch_50x3_iso", 7, "philox"),
    ("stretch_box_32x1", 3, "mt"), ("mix_stretch_gauss_32x3", 2, "mt"), ("gauss_diag_random_factor_30x4", 3, "philox"),
    ("stretch_48x130_dense", 3, "philox"), ("mix_de_snooker_40x113_dense", 2, "mt"), ("snooker_38x3_diag", 4, "mt"),
])
def test_replay_exchange_equals_single_rank(name, world, rng):
    """Replay exchange: every logical rank updates its share of each split (the fused kernel), publishes its decisions (new
    log-prob or NaN, 8 bytes per walker-update -- the all-gather is done here with device copies between the contexts'
    buffers) and recomputes the accepted updates of the others on its own replica.  Every replica's chain, log-probs and
    accept counts equal the single-rank run BIT FOR BIT -- the snooker move included: the replay runs in the row layout of
    the kernel that took the decision, so its group reductions associate identically."""
    import torch
    from emcee_amd.parallel import ReplayStepper
    g = load_golden(name)
    spec = cases.build(name)
    nst = min(8, spec["nsteps"])

    def setup(ens):
        if rng == "mt":
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(rng_from_fixture(g).get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(777, 0)
        ens.set_tuning("small_kernel", 0)
        ens.chain_config(nst)

    ref = make_ens(spec, g["p0"])
    setup(ref)
    ref.run(nst, 1, True)
    ref_chain, ref_lp, ref_acc = ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst), ref.accepted_counts()
    engines = []
    for r in range(world):
        ens = make_ens(spec, g["p0"])
        setup(ens)
        engines.append(DeviceEngine(ens, r, world, torch.device("cuda", 0), exchange="replay"))
    for _ in range(nst):
        res = [e.step_begin(True) for e in engines]
        assert all(x == res[0] for x in res)
        for split in range(res[0][1]):
            rows = [e.replay_begin(split) for e in engines]
            assert len(set(rows)) == 1
            n = rows[0]
            for e in engines:
                e.ens.sync()
            for dst in engines:                       # the all-gather of the decisions
                for r, src in enumerate(engines):
                    dst.gathered[r * n:(r + 1) * n] = src.sendbuf[:n]
            torch.cuda.synchronize()
            for e in engines:
                e.replay_finish(split)
        for e in engines:
            e.step_end()
    for e in engines:
        ens = e.ens
        assert ens.status() == 0
        assert np.array_equal(ens.chain_read(0, 0, nst), ref_chain)
        assert np.array_equal(ens.chain_read(1, 0, nst), ref_lp)
        assert np.array_equal(ens.accepted_counts(), ref_acc)
        x, lp = ens.get_state()
        assert np.array_equal(x, ref_chain[-1]) and np.array_equal(lp, ref_lp[-1])
        assert np.array_equal(ens.accepted_mask(), ref.accepted_mask())
        ens.close()
    ref.close()
    assert ReplayStepper is not None


@pytest.mark.parametrize("name,world,rng", [("stretch_128x64_dense", 3, "philox"), ("stretch_50x3_iso", 4, "mt"),
                                            ("stretch_48x130_dense", 2, "mt"), ("stretch_128x8_rosen", 5, "philox")])
def test_replay_exchange_two_pass_form_of_the_stretch_move(name, world, rng, monkeypatch):
    """The stretch move replays in one launch (k_replay_stretch); its two-pass form -- compact plan, then k_halfstep with
    TGT_REPLAY, what the other moves use -- must give the same bits (tuning key replay_two_pass, through EMX_TUNE)."""
    monkeypatch.setenv("EMX_TUNE", "replay_two_pass=1")
    test_replay_exchange_equals_single_rank(name, world, rng)


def test_replay_exchange_device_side():
    """The replay exchange with no collective at all: every rank stores its decisions into the other ranks' receive buffers
    (between GPUs: over xGMI into the peers' HBM) and the ranks meet at the one-wave barrier kernel -- no host synchronisation
    inside a step.  Every rank's whole replica (chain, log-probs, accept counts) is bit-identical to the single-rank run.
    One process per rank (see test_direct_exchange_device_side_barrier)."""
    _device_side_processes(2, [("replay", "stretch_128x64_dense", "philox"), ("replay", "mix_de_snooker_128x8_dense", "mt"),
                               ("replay", "stretch_50x3_iso", "philox"), ("replay", "stretch_48x130_dense", "mt")], 29683)


def test_device_side_exchanges_between_four_processes():
    """Both device-side protocols with four ranks in four processes (and three for the move mixture): barrier flags and receive
    buffers of three peers each, mapped through hipIpc."""
    _device_side_processes(4, [("direct", "stretch_128x64_dense", "philox"), ("replay", "stretch_128x64_dense", "mt"),
                               ("replay", "stretch_128x8_rosen", "philox")], 29685)
    _device_side_processes(3, [("direct", "mix_de_snooker_128x8_dense", "philox"), ("replay", "mix_de_snooker_128x8_dense", "mt")], 29687)


@pytest.mark.parametrize("name,world,rng", [
    ("c1_stretch_32x5_iso", 2, "mt"), ("stretch_50x3_iso", 3, "mt"), ("stretch_128x64_dense", 2, "mt"),
    ("stretch_128x64_dense", 5, "philox"), ("mix_de_snooker_128x8_dense", 3, "philox"), ("stretch_128x8_rosen", 8, "philox"),
    ("stretch_nsplits3_45x2", 2, "mt"), ("stretch_50x3_iso", 7, "philox"), ("stretch_box_32x1", 3, "mt"),
    ("mix_stretch_gauss_32x3", 2, "mt"), ("gauss_diag_random_factor_30x4", 3, "philox"),
    ("stretch_48x130_dense", 3, "philox"), ("mix_de_snooker_40x113_dense", 2, "mt"),
])
def test_logprob_exchange_equals_single_rank(name, world, rng):
    """Log-prob exchange (the reference's pool.map model): every logical rank proposes and commits the whole split and
    evaluates the target on its share of the proposals; the in-place all-gather of 8 bytes per walker is done here with
    device copies between the contexts' buffers.  Every replica's chain, log-probs and accept counts equal the
    single-rank run bit for bit (the fused kernel and propose / evaluate / commit share their arithmetic)."""
    import torch
    from emcee_amd.parallel import LogProbStepper
    g = load_golden(name)
    spec = cases.build(name)
    nst = min(8, spec["nsteps"])

    def setup(ens):
        if rng == "mt":
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(rng_from_fixture(g).get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(777, 0)
        ens.set_tuning("small_kernel", 0)
        ens.chain_config(nst)

    ref = make_ens(spec, g["p0"])
    setup(ref)
    ref.run(nst, 1, True)
    ref_chain, ref_lp, ref_acc = ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst), ref.accepted_counts()
    engines = []
    for r in range(world):
        ens = make_ens(spec, g["p0"])
        setup(ens)
        engines.append(DeviceEngine(ens, r, world, torch.device("cuda", 0), exchange="logprob"))
    for _ in range(nst):
        res = [e.step_begin(True) for e in engines]
        assert all(x == res[0] for x in res)
        for split in range(res[0][1]):
            pers = [e.logprob_begin(split) for e in engines]
            assert len(set(pers)) == 1
            per = pers[0]
            for e in engines:
                e.ens.sync()
            for dst in engines:                       # the in-place all-gather
                for r, src in enumerate(engines):
                    if src is not dst:
                        dst.gathered[r * per:(r + 1) * per] = src.gathered[r * per:(r + 1) * per]
            torch.cuda.synchronize()
            for e in engines:
                e.logprob_finish(split)
        for e in engines:
            e.step_end()
    for e in engines:
        ens = e.ens
        assert ens.status() == 0
        assert np.array_equal(ens.chain_read(0, 0, nst), ref_chain)
        assert np.array_equal(ens.chain_read(1, 0, nst), ref_lp)
        assert np.array_equal(ens.accepted_counts(), ref_acc)
        x, lp = ens.get_state()
        assert np.array_equal(x, ref_chain[-1]) and np.array_equal(lp, ref_lp[-1])
        ens.close()
    ref.close()
    assert LogProbStepper is not None


def test_library_driven_rccl_world1():
    """emx_comm_init + sharded emx_run (ncclAllGather enqueued by libemx) at world size 1:
    exercises the RCCL linkage and the exchange buffers; the chain must equal the plain run."""
    from emcee_amd.device import DeviceEnsemble
    name = "stretch_128x64_dense"
    g = load_golden(name)
    spec = cases.build(name)
    chains = []
    for use_comm in (False, "allgather", "pull", "direct", "logprob"):
        ens = make_ens(spec, g["p0"])
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(4242, 0)
        ens.chain_config(10)
        if use_comm:
            ens.set_exchange(use_comm)
            ens.comm_init(0, 1, DeviceEnsemble.rccl_unique_id())
        ens.run(10, 1, True)
        assert ens.status() == 0
        chains.append((ens.chain_read(0, 0, 10), ens.chain_read(1, 0, 10), ens.accepted_counts()))
        if use_comm:
            ens.comm_destroy()
        ens.close()
    for other in chains[1:]:
        for a, b in zip(chains[0], other):
            assert np.array_equal(a, b)
