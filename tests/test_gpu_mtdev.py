"""Exact (MT19937) mode with the plans made ON THE DEVICE (csrc/emx_mtdev.hpp; include/emx.h emx_mtdev_info): every stage against the
serial host twin of NumPy's legacy stream -- the stream itself (jump-ahead segments), the tokenizer's positions and accepted
Fisher-Yates targets (red_blue.py:80), the finished plans (red_blue.py:85, stretch.py:30-33, red_blue.py:100), the generator state
handed back (ensemble.py:410) -- and whole runs against the host pipeline's."""
import numpy as np
import pytest

from emcee_amd import _lib
from emcee_amd.device import DeviceEnsemble

from emx_testlib import HostMT, cdf_of

pytestmark = pytest.mark.gpu


def stretch_desc(S=2, randomize=1, a=2.0):
    return _lib.MoveDesc(0, S, randomize, 0, a, 1e-5, 0.5, 1.7)


def make(N, D, md, state, device_plans=1, target="iso"):
    ens = DeviceEnsemble(N, D, device=0)
    if target == "iso":
        ens.set_target(_lib.TARGET_ISO)
    else:
        rs = np.random.RandomState(D)
        A = rs.randn(D, D)
        ens.set_target(_lib.TARGET_DENSE, rs.randn(D), A @ A.T / D + np.eye(D))
    ens.set_moves([md], cdf_of(None, 1))
    ens.set_state(np.random.RandomState(N + D).randn(N, D))
    ens.eval_state_log_prob()
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_tuning("mt_device", 2 if device_plans else 0)          # 2: whatever the size (1, the default, starts at 147 456 walkers)
    ens.set_mt19937(state)
    return ens


def stream_words(state, n):
    """the next n 32-bit words NumPy's legacy generator hands out from `state`, and the absolute position (block 0 = the state's
    current block) of the first of them"""
    rs = np.random.RandomState(0)
    rs.set_state(state)
    return np.frombuffer(rs.bytes(4 * n), dtype="<u4").astype(np.uint32), int(state[2])


def serial_tokens(words, p0, p, N, S, randomize):
    """the serial walk of one step (draw order of ensemble.py:406, red_blue.py:80, stretch.py:30,32, red_blue.py:100) over `words`
    (words[k] = stream position p0 + k): accepted Fisher-Yates targets J[i], the positions of every split's draws, the end"""
    w = words
    p += 2
    J = np.zeros(N, dtype=np.uint32)
    if randomize:
        i = N - 1
        while i > 0:
            m = (1 << int(i).bit_length()) - 1
            v = int(w[p - p0]) & m
            p += 1
            if v <= i:
                J[i] = v
                i -= 1
    pos = []
    for s in range(S):
        ns = (N - s + S - 1) // S
        nc = N - ns
        pz = p
        p += 2 * ns
        pr = p
        rng = nc - 1
        if rng == 0:
            pass
        elif nc & rng == 0:
            p += ns
        else:
            m = (1 << int(rng).bit_length()) - 1
            got = 0
            while got < ns:
                got += (int(w[p - p0]) & m) <= rng
                p += 1
        pu = p
        p += 2 * ns
        pos += [pz, pr, pu]
    return J, pos, p


@pytest.mark.parametrize("N,S,randomize,seed", [(8192, 2, 1, 1), (65536, 2, 1, 2), (10000, 2, 1, 3), (8193, 3, 1, 4), (16384, 2, 0, 5), (12289, 4, 1, 6)])
def test_stream_positions_and_targets_equal_the_serial_walk(N, S, randomize, seed):
    md = stretch_desc(S, randomize)
    rs = np.random.RandomState(seed)
    rs.random_sample(seed * 100 + 7)                      # an arbitrary position inside a block
    state = rs.get_state()
    ens = make(N, 4, md, state)
    ens.set_tuning("mt_device_lookahead", 0)              # batch 0's raw pieces (targets, positions) stay in their buffers
    assert ens.mtdev_info()["qualifies"]
    k, S_ = ens.step_begin(store=False)                   # starts the producer: the first batches are enqueued
    assert (k, S_) == (0, S) and ens.mtdev_info()["alive"]
    nsteps = 3
    nwords = int(nsteps * (7 * N + 4096)) + 2000
    words, p0 = stream_words(state, nwords)
    got = ens.mtdev_debug(0, p0, nwords)
    assert np.array_equal(got, words), "stream differs from NumPy's at word %d" % int(np.argmax(got != words))
    # a stretch of the stream far ahead: across segment boundaries (1024 blocks each) of the first round
    far0 = 1024 * 624 - 1000 + p0
    rs2 = np.random.RandomState(0)
    rs2.set_state(state)
    rs2.bytes(4 * (far0 - p0))
    far = np.frombuffer(rs2.bytes(4 * 4000), dtype="<u4").astype(np.uint32)
    assert np.array_equal(ens.mtdev_debug(0, far0, 4000), far), "stream differs at a segment boundary"
    p = p0
    for step in range(nsteps):
        J, pos, p = serial_tokens(words, p0, p, N, S, randomize)
        dp = ens.mtdev_debug(2, step, 3 * S + 1)
        assert [int(x) for x in dp[:-1]] == pos and int(dp[-1]) == p, "positions of step %d differ" % step
        if randomize:
            dJ = ens.mtdev_debug(1, step)
            assert np.array_equal(dJ[1:], J[1:]), "Fisher-Yates targets of step %d differ (first at i = %d)" % (step, 1 + int(np.argmax(dJ[1:] != J[1:])))
    for s in range(S):
        ens.halfstep(s)
    ens.step_end()
    assert ens.status() == 0
    ens.close()


@pytest.mark.parametrize("N,D,S,randomize,a,seed", [(8192, 3, 2, 1, 2.0, 11), (65536, 8, 2, 1, 2.0, 12), (10000, 5, 2, 1, 1.5, 13), (8193, 2, 3, 1, 2.0, 14),
                                                   (16384, 4, 2, 0, 2.0, 15), (12289, 6, 4, 1, 2.5, 16), (262144, 2, 2, 1, 2.0, 17)])
def test_device_plans_equal_the_host_twin(N, D, S, randomize, a, seed):
    """every column of every plan of 40 consecutive steps (three batches: the slots and buffers are reused), then the generator
    state handed back"""
    md = stretch_desc(S, randomize, a)
    state = np.random.RandomState(seed).get_state()
    ens = make(N, D, md, state)
    twin = HostMT(state)
    nsteps = 40 if N <= 65536 else 18
    for step in range(nsteps):
        assert twin.choice_cdf([1.0]) == 0                 # ensemble.py:406 draws even for one move
        want = twin.plan(N, D, md)
        k, S_ = ens.step_begin(store=False)
        got = ens.plan_get(S_)
        for key in ("off", "order", "p0"):
            assert np.array_equal(got[key], want[key]), "step %d: %s differs (first at %d)" % (step, key, int(np.argmax(got[key] != want[key])))
        assert np.array_equal(got["s0"], want["s0"]), "step %d: zz differs" % step
        assert np.array_equal(got["uacc"], want["uacc"]), "step %d: accept uniforms differ" % step
        for s in range(S_):
            ens.halfstep(s)
        ens.step_end()
    assert ens.status() == 0
    info = ens.mtdev_info()
    assert info["alive"] and info["steps"] == nsteps
    a_, b_ = ens.get_mt19937(), twin.get_state()         # retires the producer: the state after the last step TAKEN
    assert np.array_equal(a_[1], b_[1]) and a_[2] == b_[2] and a_[3] == b_[3]
    assert not ens.mtdev_info()["alive"]
    ens.close()


@pytest.mark.parametrize("N,D,target,nsteps,store", [(8192, 64, "dense", 53, True), (16384, 5, "iso", 37, False), (65536, 64, "dense", 40, False),
                                                     (10000, 64, "dense", 21, True),
                                                     (8192, 4, "iso", 650, False),          # 82 batches: the stream ring wraps ~ 14 times
                                                     (131072, 4, "iso", 90, False)])        # (round 4: the default size rule's first size)
def test_runs_equal_the_host_pipelines(N, D, target, nsteps, store):
    """emx_run (two calls: the producer carries over) with device-made plans against the same run with the host pipeline's:
    coordinates, log-probs, chain, accept counters and the final generator state, bit for bit"""
    md = stretch_desc()
    state = np.random.RandomState(N % 1000 + D).get_state()
    outs = []
    for dev in (1, 0):
        ens = make(N, D, md, state, device_plans=dev, target=target)
        if store:
            ens.chain_config(2 * nsteps)
        ens.run(nsteps, 1, store)
        ens.run(nsteps, 1, store)
        assert ens.status() == 0
        info, pinfo = ens.mtdev_info(), ens.persist_info()
        x, lp = ens.get_state()
        rec = dict(x=x, lp=lp, acc=ens.accepted_mask(), counts=ens.accepted_counts(), rng=ens.get_mt19937(), info=info, pinfo=pinfo)
        if store:
            rec["chain"] = ens.chain_read(0, 0, 2 * nsteps)
        ens.close()
        outs.append(rec)
    d, h = outs
    assert d["info"]["steps"] == 2 * nsteps and h["info"]["steps"] == 0
    assert d["pinfo"]["launches"] == 0          # (the tokenizer needs a CU of its own: no persistent grid next to it)
    for key in ("x", "lp", "acc", "counts") + (("chain",) if store else ()):
        assert np.array_equal(d[key], h[key]), key
    assert np.array_equal(d["rng"][1], h["rng"][1]) and d["rng"][2] == h["rng"][2]


def test_sampler_default_rng_takes_the_device_producer():
    """EnsembleSampler(rng="mt19937", the default) at 8192 walkers with EMX_TUNE mt_device=2 (the producer at any size; by default it
    starts at 147 456 walkers, where it overtakes the host pipeline): same chain as with mt_device=0, producer used"""
    import emcee_amd
    from emcee_amd import targets
    p0 = np.random.RandomState(3).randn(8192, 6)
    chains = []
    for tune in ("mt_device=2", "mt_device=0"):
        import os
        os.environ["EMX_TUNE"] = tune
        try:
            np.random.seed(77)
            s = emcee_amd.EnsembleSampler(8192, 6, targets.IsoGaussian())
            s.run_mcmc(p0, 20)
            info = s.backend._dev.mtdev_info()
            assert (info["steps"] > 0) == (tune == "mt_device=2")
            chains.append(s.get_chain())
        finally:
            os.environ.pop("EMX_TUNE", None)
    assert np.array_equal(chains[0], chains[1])


def test_two_producers_in_one_process():
    """Two contexts of one process, each with a device producer alive at the same time, their runs interleaved (the
    stages of a producer order themselves through words that one-wave kernels spin on, and the runtime puts a process's streams onto
    a few hardware queues -- with two producers a wait can sit in front of the kernel that would signal it).  The second producer of
    a process therefore takes the event-ordered form; both give the host pipeline's chain and generator state."""
    md = stretch_desc()
    cfgs = [(16384, 5, "iso", 101), (8192, 64, "dense", 202)]
    states = [np.random.RandomState(seed).get_state() for _, _, _, seed in cfgs]
    ens = [make(N, D, md, st, device_plans=1, target=tg) for (N, D, tg, _), st in zip(cfgs, states)]
    for _ in range(3):
        for e in ens:
            e.run(23, 1, False)
    got = []
    for e in ens:
        assert e.status() == 0
        assert e.mtdev_info()["steps"] == 69
        x, lp = e.get_state()
        got.append((x, lp, e.get_mt19937()))
    for e in ens:
        e.close()
    for (N, D, tg, _), st, g in zip(cfgs, states, got):
        h = make(N, D, md, st, device_plans=0, target=tg)
        h.run(69, 1, False)
        x, lp = h.get_state()
        r = h.get_mt19937()
        h.close()
        assert np.array_equal(g[0], x) and np.array_equal(g[1], lp)
        assert np.array_equal(g[2][1], r[1]) and g[2][2] == r[2]
