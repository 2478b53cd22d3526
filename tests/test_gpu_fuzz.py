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
            targets = ["iso", "diag", "rosenbrock"] + (["dense"] if D <= 112 else [])
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
