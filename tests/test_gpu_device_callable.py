"""A user's log_prob_fn that runs ON the GPU (targets.DeviceCallable / emx_set_target_callback): the reference's vectorize=True
contract -- one call on the (Ns, ndim) block of a split's proposals, ensemble.py:486-487 -- without a PCIe hop or a host
synchronisation per split.  Pinned to the reference: torch re-statements of the fixture targets reproduce the golden chains."""
import time

import numpy as np
import pytest

import emcee_amd
from emcee_amd import targets
from oracle import cases

from helpers import load_golden
from test_gpu_sampler_api import make_sampler

pytestmark = pytest.mark.gpu


def torch_target(desc):
    import torch
    dev = torch.device("cuda", 0)
    k = desc["kind"]
    if k == "iso":
        return lambda q: -0.5 * (q * q).sum(1)
    if k == "dense":
        mu, icov = torch.as_tensor(desc["mu"], device=dev), torch.as_tensor(desc["icov"], device=dev)

        def dense(q):
            d = q - mu
            return -0.5 * ((d @ icov) * d).sum(1)
        return dense
    if k == "rosenbrock":
        return lambda q: -(100.0 * (q[:, 1:] - q[:, :-1] ** 2) ** 2 + (1.0 - q[:, :-1]) ** 2).sum(1) / 20.0
    if k == "box":
        return lambda q: torch.where(((q > 1) | (q < 0)).any(1), -float("inf"), 0.0).to(torch.float64)
    raise ValueError(k)


@pytest.mark.parametrize("name", ["c1_stretch_32x5_iso", "stretch_50x3_iso", "stretch_128x64_dense", "stretch_256x16_dense",
                                  "stretch_128x8_rosen", "stretch_box_32x1", "mix_stretch_de_64x5", "stretch_nsplits3_45x2",
                                  "mix_de_snooker_128x8_dense", "stretch_thin3_32x2"])
def test_device_callable_reproduces_the_reference_chain(name):
    g = load_golden(name)
    spec = cases.build(name)
    fn = torch_target(spec["desc"])
    calls = {"n": 0, "rows": 0, "cuda": True}

    def counted(q):
        calls["n"] += 1
        calls["rows"] += q.shape[0]
        calls["cuda"] &= bool(q.is_cuda) and q.dtype.is_floating_point and q.shape[1] == spec["D"]
        return fn(q)

    s = make_sampler(spec, g, log_prob=targets.DeviceCallable(counted))
    s.run_mcmc(g["p0"], spec["nsteps"], thin_by=spec["thin_by"], skip_initial_state_check=True)
    assert calls["cuda"]
    # the same accept decisions: identical accept counts, a row differs from the reference only through rounding of its own
    # log-prob (which never enters the coordinates): coordinates bit-identical for stretch / DE
    assert np.array_equal(s.backend.accepted, g["accepted_count"])
    if "snooker" in name:
        np.testing.assert_allclose(s.get_chain(), g["chain"], rtol=1e-9, atol=1e-10)
    else:
        assert np.array_equal(s.get_chain(), g["chain"])
    np.testing.assert_allclose(s.get_log_prob(), g["log_prob"], rtol=1e-11, atol=1e-13)
    st = s.random_state
    assert np.array_equal(st[1], g["rng_key1"]) and st[2] == int(g["rng_pos1"])
    # one call per split (+ one for the initial state), every proposal exactly once
    nprop = spec["nsteps"] * spec["thin_by"]
    assert calls["rows"] == spec["N"] * (nprop + 1)


def test_device_callable_sees_the_blocks_the_reference_hands_its_log_prob_fn():
    """ensemble.py:486-487 called at red_blue.py:93: the block is q for the walkers of the split in ASCENDING WALKER INDEX
    (s = coords[inds == split]).  Pinned with the reference's own split labels (tests/golden: `labels`, recorded from the live
    reference): the rows of every block that were accepted must be the chain rows of exactly those walkers, at those positions."""
    name = "stretch_50x3_iso"
    g = load_golden(name)
    spec = cases.build(name)
    blocks = []

    def fn(q):
        blocks.append(q.detach().cpu().numpy().copy())
        return -0.5 * (q * q).sum(1)

    s = make_sampler(spec, g, log_prob=targets.DeviceCallable(fn))
    s.run_mcmc(g["p0"], spec["nsteps"], skip_initial_state_check=True)
    assert np.array_equal(s.get_chain(), g["chain"])
    assert np.array_equal(blocks[0], g["p0"])                       # the initial state, in walker order
    prev = g["p0"]
    k = 1
    for t in range(spec["nsteps"]):
        for split in range(2):
            members = np.nonzero(g["labels"][t] == split)[0]          # ascending walker index
            q = blocks[k]
            k += 1
            assert q.shape == (len(members), spec["D"])
            moved = np.any(g["chain"][t][members] != prev[members], axis=1)
            assert moved.any()
            assert np.array_equal(q[moved], g["chain"][t][members][moved])
        prev = g["chain"][t]
    assert k == len(blocks)


@pytest.mark.parametrize("name", ["stretch_256x16_dense", "mix_stretch_de_64x5"])
def test_device_callable_with_graph_capture(name):
    """graph=True: the kernels the callable launches are captured after two eager calls per split shape and replayed"""
    g = load_golden(name)
    spec = cases.build(name)
    s = make_sampler(spec, g, log_prob=targets.DeviceCallable(torch_target(spec["desc"]), graph=True))
    s.run_mcmc(g["p0"], spec["nsteps"], skip_initial_state_check=True)
    assert np.array_equal(s.backend.accepted, g["accepted_count"])
    assert np.array_equal(s.get_chain(), g["chain"])
    np.testing.assert_allclose(s.get_log_prob(), g["log_prob"], rtol=1e-11, atol=1e-13)
    captured = [e[1] for e in s._ens._cb_graphs.values()]            # (the initial-state block is seen once: never captured)
    assert any(c for c in captured), captured                        # the split blocks were captured and replayed


def test_device_callable_errors_and_generator_path():
    import torch
    p0 = np.random.RandomState(2).randn(64, 3)

    def boom(q):
        raise KeyError("user bug")

    s = emcee_amd.EnsembleSampler(64, 3, targets.DeviceCallable(boom))
    with pytest.raises(KeyError):                      # the caller's own exception, not an EmxError
        s.run_mcmc(p0, 3)

    nanny = emcee_amd.EnsembleSampler(64, 3, targets.DeviceCallable(lambda q: torch.full((q.shape[0],), float("nan"), device=q.device,
                                                                                           dtype=torch.float64)))
    with pytest.raises(ValueError):                    # ensemble.py:550-551
        nanny.run_mcmc(p0, 3)

    short = emcee_amd.EnsembleSampler(64, 3, targets.DeviceCallable(lambda q: q[:5, 0]))
    with pytest.raises(ValueError):
        short.run_mcmc(p0, 3)

    # the sample() generator and compute_log_prob go through the same callback
    iso = targets.DeviceCallable(lambda q: -0.5 * (q * q).sum(1))
    a = emcee_amd.EnsembleSampler(64, 3, iso)
    a._random.seed(9)
    for _ in a.sample(p0, iterations=5):
        pass
    b = emcee_amd.EnsembleSampler(64, 3, targets.IsoGaussian())
    b._random.seed(9)
    b.run_mcmc(p0, 5)
    assert np.array_equal(a.get_chain(), b.get_chain())
    lp, blobs = a.compute_log_prob(p0)
    np.testing.assert_allclose(lp, -0.5 * (p0 ** 2).sum(1), rtol=1e-13)
    assert blobs is None


def test_device_callable_at_the_headline_size_is_a_device_path():
    """C2's shape through a torch log_prob_fn: the callable stays within a small factor of the fused target because nothing
    crosses PCIe (the split-phase path moves 2 x 16.8 MB per step and synchronises twice)."""
    import torch
    from bench import dense_gaussian
    N, D = 65536, 64
    mu, cov, icov = dense_gaussian(D)
    p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
    dev = torch.device("cuda", 0)
    mu_t, icov_t = torch.as_tensor(mu, device=dev), torch.as_tensor(icov, device=dev)

    def lp(q):
        d = q - mu_t
        return -0.5 * ((d @ icov_t) * d).sum(1)

    out = {}
    for label, target in (("fused", targets.DenseGaussian(mu, icov)), ("callable", targets.DeviceCallable(lp))):
        s = emcee_amd.EnsembleSampler(N, D, target, rng="philox")
        s._random.seed(3)
        st = s.run_mcmc(p0, 200, store=False, skip_initial_state_check=True)
        first = st.coords.copy()                         # after 200 steps from the same seed
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:               # clocks up
            st = s.run_mcmc(st, 100, store=False, skip_initial_state_check=True)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            st = s.run_mcmc(st, 200, store=False, skip_initial_state_check=True)
            s._ens.sync()
            best = min(best, (time.perf_counter() - t0) / 200)
        out[label] = (best, first)
    # same seed, same plan, same decisions up to the rounding of the log-prob: the ensembles agree almost everywhere
    same = np.mean(np.all(out["fused"][1] == out["callable"][1], axis=1))
    assert same > 0.99, same
    print("us/step: fused %.1f, device callable (torch, eager) %.1f" % (out["fused"][0] * 1e6, out["callable"][0] * 1e6))
    assert out["callable"][0] < 10 * out["fused"][0]


def _build_user_lib(tmp_path):
    import ctypes as C
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "user_logprob.hip")
    so = str(tmp_path / "libuser_logprob.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so], check=True, timeout=600,
                   capture_output=True)
    from emcee_amd import _lib
    _lib.load()                                  # one HIP runtime per process: the library's (torch's) first
    user = C.CDLL(so)
    user.user_setup.restype = C.c_void_p
    user.user_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    user.user_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    user.user_teardown.argtypes = [C.c_void_p]
    return user


def test_a_users_hip_kernel_through_the_c_abi(tmp_path):
    """The drop-in boundary for a likelihood that lives on the GPU: a user's shared library exports an emx_device_log_prob_fn
    that launches its own HIP kernel; libemx calls it between the proposal and the accept kernel of every split.  The reference
    fixture is reproduced, and at the headline size the step is three launches per split."""
    import ctypes as C
    user = _build_user_lib(tmp_path)
    name = "stretch_128x64_dense"
    g = load_golden(name)
    spec = cases.build(name)
    mu = np.ascontiguousarray(spec["desc"]["mu"])
    icov = np.ascontiguousarray(spec["desc"]["icov"])
    h = user.user_setup(mu.ctypes.data, icov.ctypes.data, 64)
    assert h
    s = make_sampler(spec, g, log_prob=targets.DeviceKernel(user.user_log_prob, h))
    s.run_mcmc(g["p0"], spec["nsteps"], skip_initial_state_check=True)
    assert np.array_equal(s.backend.accepted, g["accepted_count"])
    assert np.array_equal(s.get_chain(), g["chain"])
    np.testing.assert_allclose(s.get_log_prob(), g["log_prob"], rtol=1e-11)
    calls, rows = C.c_longlong(), C.c_longlong()
    user.user_stats(h, C.byref(calls), C.byref(rows))
    assert rows.value == spec["N"] * (spec["nsteps"] + 1) and calls.value == 2 * spec["nsteps"] + 1
    user.user_teardown(h)

    # headline size: fused target against the user's kernel
    from bench import dense_gaussian
    N, D = 65536, 64
    mu, cov, icov = dense_gaussian(D)
    p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
    h = user.user_setup(np.ascontiguousarray(mu).ctypes.data, np.ascontiguousarray(icov).ctypes.data, D)
    out = {}
    for label, target in (("fused", targets.DenseGaussian(mu, icov)), ("user kernel", targets.DeviceKernel(user.user_log_prob, h))):
        smp = emcee_amd.EnsembleSampler(N, D, target, rng="philox")
        smp._random.seed(3)
        st = smp.run_mcmc(p0, 10, store=False, skip_initial_state_check=True)
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:                       # clocks up
            st = smp.run_mcmc(st, 200, store=False, skip_initial_state_check=True)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            st = smp.run_mcmc(st, 400, store=False, skip_initial_state_check=True)
            smp._ens.sync()
            best = min(best, (time.perf_counter() - t0) / 400)
        out[label] = best
    user.user_teardown(h)
    print("us/step: fused %.1f, user HIP kernel through emx_set_target_callback %.1f" % (out["fused"] * 1e6, out["user kernel"] * 1e6))
    # three launches per split instead of one: the library's propose + commit passes are ~16.5 us of every split (1.4x the fused
    # step before the user's kernel does anything); this test kernel adds 22 us per launch (profiles/r03/callback_user_kernel_stats.csv)
    assert out["user kernel"] < 5 * out["fused"]


@pytest.mark.parametrize("exchange", ["replay", "allgather", "logprob"])
def test_device_callable_under_the_sharded_exchanges(exchange):
    """The caller's device log-prob with the ensemble sharded over logical ranks: every rank evaluates only its share of each
    split through the callback (the replay and all-gather exchanges) or the shares of the log-prob exchange; every replica's
    chain equals the single-rank chain of the same callable."""
    import torch
    from emcee_amd import _lib
    from emcee_amd.parallel import DeviceEngine
    name, world, nst = "stretch_256x16_dense", 3, 6
    g = load_golden(name)
    spec = cases.build(name)
    fn = torch_target(spec["desc"])
    rows = {"n": 0}

    def make(counted):
        from test_gpu_parity import make_ens
        ens = make_ens(spec, g["p0"])
        ens.set_target_callback((lambda q: (rows.__setitem__("n", rows["n"] + q.shape[0]), fn(q))[1]) if counted else fn)
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(99, 0)
        ens.set_tuning("small_kernel", 0)
        ens.eval_state_log_prob()
        ens.chain_config(nst)
        return ens

    ref = make(False)
    ref.run(nst, 1, True)
    ref_chain, ref_lp = ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst)
    ref.close()
    engines = [DeviceEngine(make(True), r, world, torch.device("cuda", 0), exchange=exchange) for r in range(world)]
    rows["n"] = 0
    for _ in range(nst):
        res = [e.step_begin(True) for e in engines]
        for split in range(res[0][1]):
            if exchange == "replay":
                n = [e.replay_begin(split) for e in engines][0]
            elif exchange == "logprob":
                n = [e.logprob_begin(split) for e in engines][0]
            else:
                for e in engines:
                    e.halfstep(split)
                n = engines[0].sendbuf.shape[0]
            for e in engines:
                e.ens.sync()
            for dst in engines:
                for r, src in enumerate(engines):
                    if exchange == "logprob":
                        if src is not dst:
                            dst.gathered[r * n:(r + 1) * n] = src.gathered[r * n:(r + 1) * n]
                    else:
                        dst.gathered[r * n:(r + 1) * n] = src.sendbuf[:n]
            torch.cuda.synchronize()
            for e in engines:
                (e.replay_finish if exchange == "replay" else e.logprob_finish if exchange == "logprob" else e.scatter_gathered)(split)
        for e in engines:
            e.step_end()
    for e in engines:
        assert e.ens.status() == 0
        assert np.array_equal(e.ens.chain_read(0, 0, nst), ref_chain) and np.array_equal(e.ens.chain_read(1, 0, nst), ref_lp)
        e.ens.close()
    assert rows["n"] == nst * spec["N"]                 # every proposal went through the callable exactly once across the ranks
