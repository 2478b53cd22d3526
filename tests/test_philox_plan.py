"""Native-mode (Philox) plan properties, evaluated by the host twin of the device function."""
import numpy as np
import pytest

from emcee_amd import _lib

from emx_testlib import philox_plan


def md(kind, S=2, a=2.0, sigma=1e-5, g0=0.3, gammas=1.7):
    return _lib.MoveDesc(kind, S, 1, 0, a, sigma, g0, gammas)


@pytest.mark.parametrize("N", [2, 3, 32, 50, 65, 1000, 4096, 65536])
@pytest.mark.parametrize("S", [2, 3, 4])
def test_split_is_a_balanced_partition(N, S):
    if S > N:
        pytest.skip("more splits than walkers")
    p = philox_plan(1234, 7, N, md(0, S))
    assert np.array_equal(np.sort(p["order"]), np.arange(N))
    sizes = np.diff(p["off"])
    assert np.array_equal(sizes, [(N - j + S - 1) // S for j in range(S)])   # arange(N) % S set sizes


def _labels(p, N):
    lab = np.empty(N, dtype=int)
    for j in range(len(p["off"]) - 1):
        lab[p["order"][p["off"][j]:p["off"][j + 1]]] = j
    return lab


def test_stretch_partners_are_in_the_complement_and_uniform():
    N = 4096
    p = philox_plan(99, 3, N, md(0, 2))
    lab = _labels(p, N)
    assert np.all(lab[p["p0"]] != lab[p["order"]])
    # zz = ((a-1)u+1)^2/a in [1/a, a]; g(z) ~ 1/sqrt(z) => u uniform
    zz = p["s0"]
    assert zz.min() >= 0.5 and zz.max() <= 2.0
    u = np.sqrt(zz * 2.0) - 1.0
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    assert abs(p["uacc"].mean() - 0.5) < 0.02
    # partner index roughly uniform over the complement
    cnt = np.bincount(p["p0"], minlength=N)
    assert cnt.max() <= 8


def test_plans_differ_between_steps_and_seeds():
    N = 1024
    a = philox_plan(5, 0, N, md(0))
    b = philox_plan(5, 1, N, md(0))
    c = philox_plan(6, 0, N, md(0))
    assert not np.array_equal(a["order"], b["order"]) and not np.array_equal(a["order"], c["order"])
    assert not np.array_equal(a["uacc"], b["uacc"])
    la, lb = _labels(a, N), _labels(b, N)
    agree = np.mean(la == lb)
    assert 0.4 < agree < 0.6                # independent random splits agree on ~half the walkers


def test_split_pair_statistics_over_steps():
    """Two fixed walkers share a sub-ensemble ~(n/2-1)/(n-1) of the time, like a uniform shuffle."""
    N = 64
    same = 0
    first = 0
    T = 2000
    for step in range(T):
        lab = _labels(philox_plan(42, step, N, md(0)), N)
        same += lab[3] == lab[17]
        first += lab[0] == 0
    assert abs(same / T - (N / 2 - 1) / (N - 1)) < 0.04
    assert abs(first / T - 0.5) < 0.04


def test_de_pairs_distinct_and_in_complement():
    N = 2048
    p = philox_plan(7, 11, N, md(1, 2, sigma=0.5, g0=0.4))
    lab = _labels(p, N)
    assert np.all(p["p0"] != p["p1"])
    assert np.all(lab[p["p0"]] != lab[p["order"]]) and np.all(lab[p["p1"]] != lab[p["order"]])
    g = (p["s0"] / 0.4 - 1.0) / 0.5
    assert abs(g.mean()) < 0.08 and abs(g.std() - 1.0) < 0.08


def test_snooker_picks_one_from_each_complement_set():
    N = 1024
    p = philox_plan(8, 2, N, md(2, 4))
    lab = _labels(p, N)
    own = lab[p["order"]]
    trio = np.stack([lab[p["p0"]], lab[p["p1"]], lab[p["p2"]]], axis=1)
    for k in range(N):
        assert sorted(trio[k].tolist() + [own[k]]) == [0, 1, 2, 3]
    # the role of z is spread over the three sets (uniform permutation)
    frac = np.mean(trio[:, 0] == np.where(own == 0, 1, 0))
    assert 0.2 < frac < 0.47


def test_persistent_grid_shape_rule():
    """the grid k_persist takes (include/emx.h emx_host_persist_shape): one 16-walker tile per wave, about one workgroup per CU,
    at least eight workgroups (one arrival counter per XCD), never more than fit co-resident; 0 = the per-half-step launches"""
    import ctypes as C
    lib = _lib.load()

    def shape(N, S=2, cu=256):
        w, g = C.c_int32(-1), C.c_int32(-1)
        assert lib.emx_host_persist_shape(N, S, cu, C.byref(w), C.byref(g)) == 0
        return w.value, g.value

    assert shape(65536) == (8, 256)                 # BASELINE config 2: 2 048 tiles, 8-wave groups, one per CU
    assert shape(49152) == (8, 192)
    assert shape(32768) == (4, 256)
    assert shape(16384) == (2, 256)
    assert shape(8192) == (1, 256)
    assert shape(512) == (1, 16)
    assert shape(65536, 4) == (4, 256)              # the snooker move's quarter ensembles
    assert shape(8192, 4) == (1, 128)
    assert shape(131072) == (0, 0)                  # more tiles than waves of a co-resident grid
    assert shape(1000) == (0, 0)                    # half an ensemble that is not whole tiles
    assert shape(65537) == (0, 0)
    assert shape(128) == (0, 0)                     # fewer than eight workgroups
    assert shape(4096, 3) == (0, 0)
    assert shape(4080, 3) == (1, 85)
    for N in range(32, 70000, 992):                 # every wave exactly one tile, whatever the size
        for S in (2, 4):
            w, g = shape(N, S)
            if w:
                assert w in (1, 2, 4, 8) and w * g * 16 * S == N and 8 <= g <= 512 and (w < 8 or g <= 256)
    assert shape(65536, 2, 304) == (8, 256) and shape(77824, 2, 304) == (8, 304)      # a device with more CUs
