"""The GF(2) machinery behind the device-side exact-plan producer (csrc/emx_mtjump.hpp, include/emx.h emx_host_mt_jump), on the CPU:
a jumped MT19937 state must equal the state NumPy's legacy generator reaches by stepping (the reference's stream:
ensemble.py:166-167, moves/red_blue.py:80, moves/stretch.py:30-32 all draw from it)."""
import numpy as np
import pytest

from emcee_amd import _lib


def stepped_key(key0, nwords):
    """state key after drawing `nwords` 32-bit words from a generator whose current block (key0) is exhausted"""
    rs = np.random.RandomState(0)
    rs.set_state(("MT19937", key0, 624, 0, 0.0))
    left = nwords
    while left > 0:
        take = min(left, 1 << 22)
        rs.bytes(4 * take)
        left -= take
    st = rs.get_state()
    return np.asarray(st[1], dtype=np.uint32), int(st[2])


@pytest.mark.parametrize("seed,stride_blocks,k", [(1, 1, 1), (2, 1, 5), (3, 7, 3), (4, 64, 2), (5, 1024, 1), (6, 1024, 3), (7, 1024, 31)])
def test_jump_polynomial_equals_stepping(seed, stride_blocks, k):
    lib = _lib.load()
    key0 = np.asarray(np.random.RandomState(seed).get_state()[1], dtype=np.uint32).copy()
    out = np.zeros(624, dtype=np.uint32)
    stride = stride_blocks * 624
    assert lib.emx_host_mt_jump(key0, stride, k, out) == 0
    # the jumped state is the key of block 1 + k * stride_blocks: draw through block 1 + k * stride_blocks's first word
    key, pos = stepped_key(key0, k * stride + 1)
    assert pos == 1
    assert np.array_equal(out, key)


def test_jump_of_a_seed_block_state_needs_no_special_case():
    """init_by_array / init_genrand states (whose word 0 has arbitrary low bits) jump correctly too: the window starts at the block
    AFTER the given key, where every word is a product of the recurrence"""
    lib = _lib.load()
    for seed in (0, 12345, 2 ** 32 - 1):
        key0 = np.asarray(np.random.RandomState(seed).get_state()[1], dtype=np.uint32).copy()
        out = np.zeros(624, dtype=np.uint32)
        assert lib.emx_host_mt_jump(key0, 3 * 624, 2, out) == 0
        key, pos = stepped_key(key0, 6 * 624 + 1)
        assert np.array_equal(out, key)
