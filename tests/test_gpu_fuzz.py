"""Randomised shape sweep: exact mode vs the oracle over many (nwalkers, ndim, nsplits, move,
target) combinations -- exercises every row-layout instantiation (G, V, CH), odd dimensions,
non-power-of-two ensembles and uneven splits."""
import numpy as np
import pytest

from emcee_amd import _lib
from oracle import cases
from oracle import sampler_oracle as so

from test_gpu_parity import assert_lp_close, make_ens

pytestmark = pytest.mark.gpu

DIMS = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 100, 112, 127, 128, 129, 200, 255, 256, 257,
        500, 513, 1024, 1030, 2048]


def configs():
    rs = np.random.RandomState(2026)
    out = []
    for D in DIMS:
        for rep in range(2):
            kind = ["stretch", "de", "snooker"][rs.randint(3)] if D <= 256 else "stretch"
            if kind == "snooker" and D > 128:
                kind = "de"
            targets = ["iso", "diag", "rosenbrock"] + (["dense"] if D <= 1030 else [])
            target = targets[rs.randint(len(targets))]
            nsplits = 4 if kind == "snooker" else int(rs.randint(2, 5))
            N = int(2 * D + rs.randint(nsplits * 2 + 2, 40)) if D <= 300 else int(nsplits * 3 + rs.randint(0, 9))
            out.append((D, N, kind, target, nsplits, int(rs.randint(1 << 30))))
    return out


@pytest.mark.parametrize("D,N,kind,target,nsplits,seed", configs())
def test_exact_mode_random_shapes(D, N, kind, target, nsplits, seed):
    mv = so.MoveSpec(kind, nsplits=nsplits, live_dangerously=True, sigma=0.05)
    cases.DIGEST_CASES["_fz"] = dict(N=N, D=D, target=target, moves=[mv], nsteps=3, seed=seed % 100000,
                                     p0="rosen" if target == "rosenbrock" else "randn")
    spec = cases.build("_fz")
    del cases.DIGEST_CASES["_fz"]
    fn = cases.make_target(spec["desc"])
    rs = np.random.RandomState(seed)
    out = so.run(spec["p0"], 3, fn, rs, moves=[mv])
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(np.random.RandomState(seed).get_state())
    ens.chain_config(3)
    ens.run(3, 1, True)
    assert ens.status() == 0
    chain = ens.chain_read(0, 0, 3)
    assert np.array_equal(ens.accepted_counts(), out["accepted_count"]), "accept decisions differ"
    if kind == "snooker":
        np.testing.assert_allclose(chain, out["chain"], rtol=1e-9, atol=1e-10)
    else:
        assert np.array_equal(chain, out["chain"])
    assert_lp_close(ens.chain_read(1, 0, 3), out["log_prob"], 1e-9)
    ens.close()


def sharded_configs():
    rs = np.random.RandomState(777)
    out = []
    for rep in range(24):
        D = int([1, 2, 3, 5, 8, 16, 17, 33, 64, 100, 130, 300][rs.randint(12)])
        kind = ["stretch", "de", "snooker", "gaussian"][rs.randint(4)]
        if D > 128 and kind == "snooker":
            kind = "stretch"
        targets = ["iso", "diag", "rosenbrock"] + (["dense"] if D <= 1030 else [])
        target = targets[rs.randint(len(targets))]
        nsplits = 4 if kind == "snooker" else int(rs.randint(2, 4))
        world = int(rs.randint(2, 9))
        N = int(max(2 * D, world * nsplits * 2) + rs.randint(2, 40))
        out.append((D, N, kind, target, nsplits, world, ["allgather", "pull"][rep % 2], ["mt", "philox"][(rep // 2) % 2],
                    int(rs.randint(1 << 30))))
    return out


@pytest.mark.parametrize("D,N,kind,target,nsplits,world,exchange,rng,seed", sharded_configs())
def test_sharded_random_shapes(D, N, kind, target, nsplits, world, exchange, rng, seed):
    """Random (shape, move, target, world size, exchange, RNG mode): `world` logical ranks on the one GPU must
    reproduce the single-rank run bit for bit -- every walker block, and every replica after the final sync."""
    import torch
    from emcee_amd.parallel import DeviceEngine, LocalGroup, block_range
    if kind == "gaussian":
        mv = so.MoveSpec("gaussian", cov=0.3 / D, mode=["vector", "random", "sequential"][seed % 3])
    else:
        mv = so.MoveSpec(kind, nsplits=nsplits, live_dangerously=True, sigma=0.05)
    cases.DIGEST_CASES["_fz"] = dict(N=N, D=D, target=target, moves=[mv], nsteps=4, seed=seed % 100000,
                                     p0="rosen" if target == "rosenbrock" else "randn")
    spec = cases.build("_fz")
    del cases.DIGEST_CASES["_fz"]
    nst = 4

    def setup(ens):
        if rng == "mt":
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(np.random.RandomState(seed).get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(seed, 0)
        ens.chain_config(nst)

    ref = make_ens(spec, spec["p0"])
    setup(ref)
    ref.run(nst, 1, True)
    ref_chain, ref_lp, ref_acc = ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst), ref.accepted_counts()
    ref.close()
    engines = []
    for r in range(world):
        ens = make_ens(spec, spec["p0"])
        setup(ens)
        engines.append(DeviceEngine(ens, r, world, torch.device("cuda", 0), exchange=exchange))

    def sync():
        for e in engines:
            e.ens.sync()
        torch.cuda.synchronize()

    for _ in range(nst):
        res = [e.step_begin(True) for e in engines]
        for split in range(res[0][1]):
            if exchange == "pull":
                caps = [e.pull_prepare(split) for e in engines]
                sync()
                LocalGroup._all_to_all(engines, caps[0] * (D + 1))
                sync()
                for e in engines:
                    e.pull_apply(split)
            else:
                for e in engines:
                    e.halfstep(split)
                sync()
                LocalGroup._all_gather(engines)
                sync()
                for e in engines:
                    e.scatter_gathered(split)
        for e in engines:
            e.step_end()
    if exchange == "pull":
        for r, e in enumerate(engines):
            lo, hi = block_range(N, r, world)
            assert e.ens.status() == 0
            assert np.array_equal(e.ens.chain_read(0, 0, nst)[:, lo:hi], ref_chain[:, lo:hi])
            assert np.array_equal(e.ens.chain_read(1, 0, nst)[:, lo:hi], ref_lp[:, lo:hi])
        per = [e.replica_pack() for e in engines]
        sync()
        LocalGroup._all_gather_n(engines, per[0] * (D + 3))
        sync()
        for e in engines:
            e.replica_unpack()
    for e in engines:
        assert e.ens.status() == 0
        if exchange == "allgather":
            assert np.array_equal(e.ens.chain_read(0, 0, nst), ref_chain)
            assert np.array_equal(e.ens.chain_read(1, 0, nst), ref_lp)
        x, lp = e.ens.get_state()
        assert np.array_equal(x, ref_chain[-1]) and np.array_equal(lp, ref_lp[-1])
        assert np.array_equal(e.ens.accepted_counts(), ref_acc)
        e.ens.close()
