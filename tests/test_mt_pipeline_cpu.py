"""The exact-mode plan pipeline (emcee_amd/csrc/emx_mtpipe.cpp: generator / tokenizer / finisher threads) must
produce, step for step, the plans and the final MT19937 state of the serial host twin (emx_mt_choice_cdf +
emx_host_plan_mt), which tests/test_mt19937_exact.py pins word-for-word to NumPy's legacy RandomState and to the
oracle's draws.  Host only: no GPU needed."""
import ctypes as C
import zlib

import numpy as np
import pytest

from emcee_amd import _lib

from emx_testlib import HostMT


def md(kind, S=2, rs=1, a=2.0, sigma=1e-5, g0=0.3, gammas=1.7):
    return _lib.MoveDesc({"stretch": 0, "de": 1, "snooker": 2}[kind], 4 if kind == "snooker" and S < 4 else S, rs, 0, a, sigma, g0,
                         gammas)


def serial(state, N, D, moves, cdf, nsteps):
    m = HostMT(state)
    out = []
    for _ in range(nsteps):
        k = m.choice_cdf(cdf)
        out.append((k, m.plan(N, D, moves[k])))
    return out, m.get_state()


def stream(state, N, D, moves, cdf, nsteps, workers, nsinks):
    lib = _lib.load()
    m = HostMT(state)
    arr = (_lib.MoveDesc * len(moves))(*moves)
    mv = np.empty(nsteps, dtype=np.int32)
    order, p0, p1, p2 = (np.empty(nsteps * N, dtype=np.int32) for _ in range(4))
    s0, ua = np.empty(nsteps * N), np.empty(nsteps * N)
    sec = C.c_double()
    rc = lib.emx_host_plan_mt_stream(m.h, N, D, len(moves), arr, np.ascontiguousarray(cdf, dtype=np.float64), nsteps, workers, nsinks,
                                     mv.ctypes.data, order.ctypes.data, p0.ctypes.data, p1.ctypes.data, p2.ctypes.data,
                                     s0.ctypes.data, ua.ctypes.data, C.byref(sec))
    assert rc > 0, rc
    plans = [(int(mv[n]), dict(order=order[n * N:(n + 1) * N], p0=p0[n * N:(n + 1) * N], p1=p1[n * N:(n + 1) * N],
                               p2=p2[n * N:(n + 1) * N], s0=s0[n * N:(n + 1) * N], uacc=ua[n * N:(n + 1) * N])) for n in range(nsteps)]
    return plans, m.get_state(), sec.value


def same_state(a, b):
    return np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and a[4] == b[4]


CASES = [
    # name, N, D, moves, weights, nsteps, start position in the block, cached normal
    ("stretch_pow2", 4096, 8, [md("stretch")], None, 12, None, False),
    ("stretch_odd", 1001, 3, [md("stretch")], None, 9, 623, False),
    ("stretch_nsplits5", 333, 2, [md("stretch", S=5)], None, 7, 624, False),
    ("stretch_fixed_split", 64, 2, [md("stretch", rs=0)], None, 5, 1, False),
    ("de", 512, 4, [md("de")], None, 9, None, False),
    ("de_odd_with_cached_normal", 203, 4, [md("de", S=3)], None, 11, 17, True),
    ("snooker", 256, 4, [md("snooker")], None, 9, None, False),
    ("snooker_odd", 131, 3, [md("snooker", S=5)], None, 6, 300, False),
    ("mixture", 300, 5, [md("stretch"), md("de"), md("snooker")], [0.5, 0.3, 0.2], 24, 5, True),
    ("tiny", 4, 1, [md("stretch")], None, 40, None, False),
    # sizes at which the tokenizer's 16-word scans of the DE and snooker draws run (pair codes, polar candidates, the snooker move's
    # five draws a walker), with and without rejection in the ranges, odd set sizes, a cached normal across steps
    ("de_4096", 4096, 4, [md("de")], None, 7, None, False),
    ("de_5003_three_sets", 5003, 4, [md("de", S=3)], None, 7, 611, True),
    ("snooker_4096", 4096, 4, [md("snooker")], None, 7, None, False),
    ("snooker_5003_five_sets", 5003, 3, [md("snooker", S=5)], None, 6, 77, False),
    ("mixture_3000", 3000, 5, [md("stretch"), md("de"), md("snooker")], [0.3, 0.4, 0.3], 30, 5, True),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("workers,nsinks", [(1, 2), (3, 4), (2, 16)])
def test_pipeline_plans_equal_the_serial_twin(case, workers, nsinks):
    name, N, D, moves, w, nsteps, pos, cached = case
    rs = np.random.RandomState(zlib.crc32(name.encode()))
    if cached:
        rs.randn(1)                          # leaves a cached second normal in the state
    st = list(rs.get_state())
    if pos is not None:
        st[2] = pos
    st = tuple(st)
    w = np.ones(len(moves)) if w is None else np.asarray(w, dtype=float)
    cdf = np.cumsum(w / w.sum())
    cdf /= cdf[-1]
    want, want_state = serial(st, N, D, moves, cdf, nsteps)
    got, got_state, _ = stream(st, N, D, moves, cdf, nsteps, workers, nsinks)
    for n, ((ka, pa), (kb, pb)) in enumerate(zip(want, got)):
        assert ka == kb, (n, ka, kb)
        kind = moves[ka].kind
        keys = ["order", "p0", "uacc", "s0"] + (["p1", "p2"] if kind != 0 else [])
        for key in keys:
            assert np.array_equal(pa[key], pb[key]), (name, n, key)
    assert same_state(want_state, got_state)


def test_pipeline_at_the_headline_size_and_its_throughput():
    """65 536 walkers (BASELINE configs[1]): a few steps, bit-identical; prints the host-side production rate."""
    N, D, nsteps = 65536, 64, 6
    moves = [md("stretch")]
    st = np.random.RandomState(11).get_state()
    want, want_state = serial(st, N, D, moves, np.array([1.0]), nsteps)
    got, got_state, sec = stream(st, N, D, moves, np.array([1.0]), nsteps, 0, 16)
    for (ka, pa), (kb, pb) in zip(want, got):
        for key in ("order", "p0", "s0", "uacc"):
            assert np.array_equal(pa[key], pb[key]), key
    assert same_state(want_state, got_state)
    print("pipeline: %.3f ms per step of 65536 walkers (thread start-up included)" % (sec * 1e3 / nsteps))


def test_pipeline_random_shapes_equal_the_serial_twin():
    """Random ensemble sizes (8 ... 40 000, powers of two among them), set counts, moves, thread and slot counts, start positions in
    the generator's block, with and without a cached normal: the boundaries of the tokenizer's vector scans (refills of the ring,
    windows that end inside a walker's draws, batches met exactly) fall differently in every case."""
    rs0 = np.random.RandomState(20260925)
    for it in range(16):
        N = int(rs0.choice([rs0.randint(8, 300), rs0.randint(300, 5000), rs0.randint(5000, 40000), 2 ** rs0.randint(5, 16)]))
        kind = rs0.choice(["de", "snooker", "stretch", "mix"])
        S = int(rs0.choice([2, 3, 4, 5]))
        if kind == "snooker" and S < 4:
            S = 4
        N = max(N, 4 * S + 8)
        moves = {"de": [md("de", S=S)], "snooker": [md("snooker", S=S)], "stretch": [md("stretch", S=S)],
                 "mix": [md("stretch", S=2), md("de", S=S), md("snooker", S=max(4, S))]}[kind]
        w = np.ones(len(moves))
        cdf = np.cumsum(w / w.sum())
        cdf /= cdf[-1]
        rs = np.random.RandomState(rs0.randint(1 << 30))
        if rs0.rand() < 0.5:
            rs.randn(1)
        st = list(rs.get_state())
        st[2] = int(rs0.randint(0, 625))
        st = tuple(st)
        nsteps = int(rs0.randint(3, 7))
        workers, nsinks = int(rs0.choice([1, 2, 3, 6])), int(rs0.choice([2, 4, 16]))
        want, want_state = serial(st, N, 4, moves, cdf, nsteps)
        got, got_state, _ = stream(st, N, 4, moves, cdf, nsteps, workers, nsinks)
        for n, ((ka, pa), (kb, pb)) in enumerate(zip(want, got)):
            assert ka == kb, (it, kind, N, S, n)
            for key in ["order", "p0", "uacc", "s0"] + (["p1", "p2"] if moves[ka].kind != 0 else []):
                assert np.array_equal(pa[key], pb[key]), (it, kind, N, S, n, key)
        assert same_state(want_state, got_state), (it, kind, N, S)


@pytest.mark.parametrize("N,moves,w,pos,regen_min", [
    (65536, [md("stretch")], None, None, 16384),           # BASELINE configs[1]: every step a regen step
    (16384, [md("stretch")], None, 623, 16384),
    (32768, [md("stretch")], None, 1, 16384),
    (16384, [md("stretch"), md("de")], [0.6, 0.4], 300, 16384),      # a mixture: regen for its stretch steps only
    (24576, [md("stretch")], None, 77, 16384),             # half an ensemble that is no power of two: raw, with accepted randint values
    (8192, [md("stretch")], None, None, 4096),             # a lower threshold
    (4096, [md("stretch")], None, 5, 0),                   # regen off: raw steps
    (1001, [md("stretch", S=3)], None, 17, 16),            # three splits: raw, never regen
])
def test_device_finish_hand_over_through_the_host_twins(monkeypatch, N, moves, w, pos, regen_min):
    """Round 6: with device finish the pipeline hands a stretch step over as generator WORDS in the plan's columns (raw) or -- ensembles
    of `mt_regen_min_walkers` and more whose half is a power of two -- as `order` plus the generator's STATE at every eighth block of
    the region of the stream the step's 5 N fixed-length draws stand in (regen); the consumer's kernels (k_plan_regen, k_plan_raw) make
    the plan from that.  Their host twins (csrc/emx.hip, test mode EMX_TEST_PIPE_DEVFIN of emx_host_plan_mt_stream) are run here: the
    plans and the final generator state must be the serial twin's bit for bit -- i.e. the tokenizer handed over the right states, the
    right offset, and stepped over exactly the right words."""
    monkeypatch.setenv("EMX_TEST_PIPE_DEVFIN", str(regen_min))
    nsteps = 9
    st = list(np.random.RandomState(N + 7).get_state())
    if pos is not None:
        st[2] = pos
    st = tuple(st)
    wts = np.ones(len(moves)) if w is None else np.asarray(w, dtype=float)
    cdf = np.cumsum(wts / wts.sum())
    cdf /= cdf[-1]
    want, want_state = serial(st, N, 4, moves, cdf, nsteps)
    got, got_state, _ = stream(st, N, 4, moves, cdf, nsteps, 3, 4)
    nregen = nraw = 0
    for n, ((ka, pa), (kb, pb)) in enumerate(zip(want, got)):
        assert ka == (kb & 255), (n, ka, kb)
        nraw += bool(kb & 256)
        nregen += bool(kb & 512)
        for key in ["order", "p0", "uacc", "s0"] + (["p1", "p2"] if moves[ka].kind != 0 else []):
            assert np.array_equal(pa[key], pb[key]), (N, n, key, bool(kb & 512))
    assert same_state(want_state, got_state)
    nstretch = sum(1 for k, _ in want if moves[k].kind == 0)
    assert nraw == nstretch
    half_pow2 = N % 2 == 0 and (N // 2) & (N // 2 - 1) == 0 and moves[0].nsplits == 2
    assert nregen == (nstretch if (regen_min and N >= regen_min and half_pow2) else 0)
