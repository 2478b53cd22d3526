"""EMX_MOVE_GAUSS through the C ABI: the fused Gaussian Metropolis step (moves/gaussian.py + moves/mh.py).

* MT19937 mode, free-running: the reference-generated fixtures (same chain as reference emcee);
* INPUTS mode: normals, columns and accept uniforms supplied, against the oracle's arithmetic;
* Philox mode: the step replayed from the plan and the displacement rows the device generated;
  the draws themselves checked for their distribution."""
import numpy as np
import pytest

from emcee_amd import _lib
from oracle import cases
from oracle import sampler_oracle as so

from emx_testlib import move_desc
from helpers import load_golden, rng_from_fixture
from test_gpu_parity import assert_lp_close, make_ens

pytestmark = pytest.mark.gpu

FUSED = ["gauss_iso_vector_40x3", "gauss_diag_random_factor_30x4", "gauss_iso_sequential_24x3", "mix_stretch_gauss_32x3"]


def read_disp(ens):
    """The (N, D) displacement rows of the step begun, copied from HBM."""
    import torch
    from devfft_twin import _DevView
    ptr, nbytes = ens.device_ptr(6)
    assert ptr and nbytes == ens.nwalkers * ens.ndim * 8
    ens.sync()
    t = torch.as_tensor(_DevView(ptr, (ens.nwalkers, ens.ndim)), device=torch.device("cuda", torch.cuda.current_device()))
    return t.cpu().numpy().copy()


@pytest.mark.parametrize("name", FUSED)
def test_exact_mode_reproduces_reference_fixture(name):
    g = load_golden(name)
    spec = cases.build(name)
    ens = make_ens(spec, g["p0"])
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(rng_from_fixture(g).get_state())
    ens.chain_config(spec["nsteps"])
    ens.run(spec["nsteps"], 1, True)
    assert ens.status() == 0
    chain = ens.chain_read(0, 0, spec["nsteps"])
    if any(m.factor is not None for m in spec["moves"]):
        np.testing.assert_allclose(chain, g["chain"], rtol=1e-13, atol=1e-15)     # exp() of the factor: libm vs NumPy
    else:
        assert np.array_equal(chain, g["chain"])
    assert_lp_close(ens.chain_read(1, 0, spec["nsteps"]), g["log_prob"])
    assert np.array_equal(ens.accepted_counts(), g["accepted_count"])
    st = ens.get_mt19937()
    assert np.array_equal(st[1], g["rng_key1"]) and st[2] == int(g["rng_pos1"])
    assert st[3] == int(g["rng_has_gauss1"]) and st[4] == float(g["rng_cached1"])
    ens.close()


@pytest.mark.parametrize("N,D,target,cov,mode", [
    (64, 5, "iso", 0.3, "vector"), (48, 7, "diag", None, "random"), (40, 64, "dense", 0.01, "vector"),
    (33, 3, "iso", 0.7, "sequential"), (128, 130, "diag", None, "vector"), (256, 16, "dense", None, "random"),
])
def test_inputs_mode_equals_oracle(N, D, target, cov, mode):
    rs = np.random.RandomState(N * 1000 + D)
    covv = cov if cov is not None else 0.05 + rs.rand(D)
    mv = so.MoveSpec("gaussian", cov=covv, mode=mode)
    spec = dict(N=N, D=D, moves=[mv], weights=None)
    desc = {"kind": target}
    if target == "diag":
        desc.update(mu=rs.randn(D), ivar=1.0 / (0.1 + rs.rand(D)))
    elif target == "dense":
        mu, cov_, icov = cases._dense_params(D, 5)
        desc.update(mu=mu, cov=cov_, icov=icov)
    spec["desc"] = desc
    fn = cases.make_target(desc)
    p0 = rs.randn(N, D) * 0.5 + (desc.get("mu", 0.0))
    ens = make_ens(spec, p0)
    ens.set_rng_mode(_lib.RNG_INPUTS)
    x, lp = p0.copy(), np.asarray(fn(p0), dtype=np.float64)
    scale = np.sqrt(covv)
    for step in range(5):
        f = float(np.exp(rs.uniform(-0.3, 0.3)))
        normals = rs.randn(N, D)
        col = {"vector": np.full(N, -1), "random": rs.randint(D, size=N), "sequential": np.full(N, step % D)}[mode]
        uacc = rs.rand(N)
        order = rs.permutation(N).astype(np.int32)          # slots need not be in walker order
        ens.step_begin(False)
        ens.plan_set(0, dict(off=[0, N], order=order, p0=col[order], p1=order, p2=order, s0=np.zeros(N), uacc=uacc))
        ens.plan_set_noise(normals, f)
        ens.halfstep(0)
        ens.step_end()
        # oracle arithmetic: gaussian.py:87,92-101 and mh.py:56-57
        xnew = x + f * scale * normals
        q = xnew if mode == "vector" else x.copy()
        if mode != "vector":
            q[np.arange(N), col] = xnew[np.arange(N), col]
        new_lp = np.asarray(fn(q), dtype=np.float64)
        with np.errstate(divide="ignore"):
            u_of_walker = np.empty(N)
            u_of_walker[order] = uacc
            acc = np.log(u_of_walker) < new_lp - lp
        x[acc], lp[acc] = q[acc], new_lp[acc]
        gx, glp = ens.get_state()
        assert ens.status() == 0
        assert np.array_equal(ens.accepted_mask(), acc)
        assert np.array_equal(gx, x)
        assert_lp_close(glp, lp)
    ens.close()


@pytest.mark.parametrize("mode,factor", [("vector", None), ("random", 1.5), ("sequential", None)])
def test_native_step_replays_from_its_own_draws(mode, factor):
    """Philox mode: plan (columns, accept uniforms) and displacement rows read back from the device,
    the step recomputed with the oracle's formulas."""
    N, D = 96, 6
    rs = np.random.RandomState(4)
    mv = so.MoveSpec("gaussian", cov=0.2 + rs.rand(D), mode=mode, factor=factor)
    spec = dict(N=N, D=D, moves=[mv], weights=None, desc={"kind": "iso"})
    p0 = rs.randn(N, D)
    ens = make_ens(spec, p0)
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(99, 0)
    ens.set_tuning("gauss_materialize", 1)        # rows to HBM (k_gauss_disp) so that the test can read them
    x, lp = p0.copy(), so.iso_gauss(p0)
    for step in range(6):
        k, S = ens.step_begin(False)
        assert (k, S) == (0, 1)
        plan = ens.plan_get(1)
        disp = read_disp(ens)
        ens.halfstep(0)
        ens.step_end()
        assert np.array_equal(plan["order"], np.arange(N))
        col = plan["p0"]
        if mode == "vector":
            assert np.all(col == -1)
            q = x + disp
        else:
            assert np.all((col >= 0) & (col < D))
            if mode == "sequential":
                assert np.all(col == step % D)
            q = x.copy()
            q[np.arange(N), col] = x[np.arange(N), col] + disp[np.arange(N), col]
        new_lp = so.iso_gauss(q)
        acc = np.log(plan["uacc"]) < new_lp - lp
        x[acc], lp[acc] = q[acc], new_lp[acc]
        gx, glp = ens.get_state()
        assert np.array_equal(ens.accepted_mask(), acc) and np.array_equal(gx, x)
    assert int(ens.get_move(0).gammas) == (6 % D if mode == "sequential" else 0)
    ens.close()


def test_native_normals_are_standard_normal():
    N, D = 4096, 9
    mv = so.MoveSpec("gaussian", cov=1.0)
    spec = dict(N=N, D=D, moves=[mv], weights=None, desc={"kind": "iso"})
    ens = make_ens(spec, np.zeros((N, D)))
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(7, 0)
    ens.set_tuning("gauss_materialize", 1)
    blocks = []
    for _ in range(4):
        ens.step_begin(False)
        blocks.append(read_disp(ens))
        ens.step_end()
    z = np.stack(blocks)
    n = z.size
    assert abs(z.mean()) < 5 / np.sqrt(n) and abs(z.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs(np.mean(z ** 4) - 3) < 0.1 and abs(np.mean(z ** 3)) < 0.05
    c = np.corrcoef(z.reshape(-1, D).T)
    assert np.max(np.abs(c - np.eye(D))) < 0.03
    assert not np.array_equal(blocks[0], blocks[1])
    assert abs(np.mean(np.abs(z) > 3) - 0.0027) < 0.001
    ens.close()


@pytest.mark.parametrize("N,D,target,mode,factor", [
    (64, 5, "iso", "vector", None), (200, 64, "dense", "vector", 1.3), (96, 7, "diag", "random", None),
    (50, 2, "iso", "sequential", 2.0), (40, 130, "diag", "vector", None), (333, 33, "rosenbrock", "random", 1.1),
    (128, 1024, "diag", "vector", None),
])
def test_rows_generated_in_registers_equal_the_materialised_rows(N, D, target, mode, factor):
    """Default native path (normals generated inside k_halfstep, no HBM round trip) vs the same draws written
    by k_gauss_disp and read back as rows: same chain, bit for bit, in every row layout (V = 1 and 2)."""
    rs = np.random.RandomState(D)
    mv = so.MoveSpec("gaussian", cov=((0.5 + rs.rand(D)) / D) if D != 5 else 0.1, mode=mode, factor=factor)
    desc = {"kind": target}
    if target == "diag":
        desc.update(mu=rs.randn(D), ivar=1.0 / (0.1 + rs.rand(D)))
    elif target == "dense":
        mu, cov_, icov = cases._dense_params(D, 5)
        desc.update(mu=mu, cov=cov_, icov=icov)
    spec = dict(N=N, D=D, moves=[mv], weights=None, desc=desc)
    p0 = (1.0 if target == "rosenbrock" else 0.0) + 0.1 * rs.randn(N, D) + desc.get("mu", 0.0)
    out = []
    for materialise in (0, 1):
        ens = make_ens(spec, p0)
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(2024, 3)
        ens.set_tuning("gauss_materialize", materialise)
        ens.chain_config(12)
        ens.run(12, 1, True)
        assert ens.status() == 0
        out.append((ens.chain_read(0, 0, 12), ens.chain_read(1, 0, 12), ens.accepted_counts()))
        ens.close()
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    acc = out[0][2].sum() / (12 * N)
    assert 0.0 < acc < 1.0, acc


def test_gaussian_move_validation_is_loud():
    from emcee_amd.device import DeviceEnsemble, EmxError
    ens = DeviceEnsemble(16, 3)
    with pytest.raises(EmxError, match="nsplits must be 1"):
        ens.set_moves([_lib.MoveDesc(_lib.MOVE_GAUSS, 2, 0, 0, 0.0, 1.0, 0.0, 0.0)], np.array([1.0]))
    with pytest.raises(EmxError, match="unknown Gaussian mode"):
        ens.set_moves([_lib.MoveDesc(_lib.MOVE_GAUSS, 1, 0, 5, 0.0, 1.0, 0.0, 0.0)], np.array([1.0]))
    ens.set_moves([_lib.MoveDesc(_lib.MOVE_GAUSS, 1, 0, 0, 0.0, 1.0, 0.0, 0.0)], np.array([1.0]))
    with pytest.raises(EmxError, match="ndim entries"):
        ens.set_move_scale(0, np.ones(5))
    ens.close()
