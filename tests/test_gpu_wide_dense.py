"""Dense Gaussian targets wider than the LDS-resident precision matrix (padded ndim > 112; emx_wide.hip):
propose -> k_wide_lp (f64 MFMA, L streamed through LDS) -> k_wide_commit, all on the device.

The reference fixtures of such targets (stretch_48x130_dense, mix_de_snooker_40x113_dense, the 256- and 520-dim digest
cases) run through the generic parity tests (test_gpu_parity.py, test_gpu_sharded.py).  Here: the wide path against
the FUSED kernel on dimensions both can take -- same contraction order, so the chains must agree bit for bit -- and
ragged / multi-macro-block shapes against the oracle."""
import numpy as np
import pytest

from emcee_amd import _lib
from oracle import cases
from oracle import sampler_oracle as so

from emx_testlib import cdf_of, move_desc
from test_gpu_parity import LP_RTOL, assert_lp_close, make_ens

pytestmark = pytest.mark.gpu

_S = so.MoveSpec


def _spec(N, D, moves, weights=None, seed=3):
    mu, cov, icov = cases._dense_params(D, seed + 1000)
    rs = np.random.RandomState(seed)
    p0 = mu + rs.randn(N, D) @ np.linalg.cholesky(cov).T
    return dict(N=N, D=D, desc=dict(kind="dense", mu=mu, cov=cov, icov=icov), moves=moves, weights=weights, p0=p0, seed=seed)


@pytest.mark.parametrize("N,D,moves,weights", [
    (600, 64, [_S("stretch")], None),
    (333, 100, [_S("stretch", nsplits=3)], None),
    (512, 48, [_S("de"), _S("snooker")], [0.6, 0.4]),
    (400, 17, [_S("stretch"), _S("gaussian", cov=0.01)], [0.5, 0.5]),
    (4100, 112, [_S("stretch")], None),
])
@pytest.mark.parametrize("store", [True, False])
def test_wide_path_equals_fused_kernel_bit_for_bit(N, D, moves, weights, store):
    spec = _spec(N, D, moves, weights)
    outs = []
    for wide in (0, 1):
        ens = make_ens(spec, spec["p0"])
        ens.set_tuning("dense_wide", wide)
        ens.set_tuning("small_kernel", 0)
        ens.set_tuning("graph", 0)
        ens.eval_state_log_prob()
        lp0 = ens.get_state()[1]
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(90210, 0)
        nst = 12
        ens.chain_config(nst)
        ens.run(nst, 1, store)
        assert ens.status() == 0
        x, lp = ens.get_state()
        rec = dict(lp0=lp0, x=x, lp=lp, acc=ens.accepted_mask().copy())
        if store:
            rec.update(chain=ens.chain_read(0, 0, nst), clp=ens.chain_read(1, 0, nst), cnt=ens.accepted_counts())
        outs.append(rec)
        ens.close()
    a, b = outs
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    if store:
        assert a["cnt"].sum() > 0


@pytest.mark.parametrize("N,D", [(64, 113), (300, 130), (1000, 160), (77, 257), (2100, 384), (40, 1000)])
def test_wide_stretch_steps_against_oracle(N, D):
    """Exact mode, free-running from a seeded generator: accept decisions and coordinates equal to the oracle's
    (stretch proposals involve no reduction), log-probs within the dense-target tolerance."""
    spec = _spec(N, D, [_S("stretch", live_dangerously=True)], seed=D)
    fn = cases.make_target(spec["desc"])
    nst = 3
    rs = np.random.RandomState(1234 + D)
    out = so.run(spec["p0"], nst, fn, np.random.RandomState(1234 + D), moves=spec["moves"], weights=None, thin_by=1)
    ens = make_ens(spec, spec["p0"])
    ens.set_rng_mode(_lib.RNG_MT19937)
    ens.set_mt19937(rs.get_state())
    ens.chain_config(nst)
    ens.run(nst, 1, True)
    assert ens.status() == 0
    assert np.array_equal(ens.accepted_counts(), out["accepted_count"])
    assert np.array_equal(ens.chain_read(0, 0, nst), out["chain"])
    assert_lp_close(ens.chain_read(1, 0, nst), out["log_prob"], 1e-10)
    ens.close()


def test_wide_non_finite_proposal_is_rejected_and_reported():
    """ensemble.py:476-479: a non-finite coordinate raises; on the device the proposal is rejected and the sticky status
    bit is set, whichever path evaluates the target."""
    N, D = 64, 130
    spec = _spec(N, D, [_S("stretch", live_dangerously=True)])
    p0 = np.full((N, D), 1.7e308)          # x_j - x_k overflows for every pair of opposite sign
    p0[1::2] = -1.7e308
    ens = make_ens(spec, spec["p0"])
    ens.set_state(p0)
    ens.eval_state_log_prob()
    ens.set_rng_mode(_lib.RNG_PHILOX)
    ens.set_philox(5, 0)
    ens.run(2, 1, False)
    assert ens.status() & 2, "bad-coordinate bit"
    x, lp = ens.get_state()
    assert np.all(np.isfinite(x))
    ens.close()


@pytest.mark.parametrize("N,D,moves", [(65536 + 40, 64, [_S("stretch")]), (2 * 8 * 256 * 16 + 16, 130, [_S("de", live_dangerously=True)]),
                                       (600, 300, [_S("stretch", live_dangerously=True)]), (90, 1000, [_S("stretch", live_dangerously=True)]),
                                       (5000, 520, [_S("de", live_dangerously=True)])])
def test_role_split_log_prob_kernel_equals_the_single_role_one(N, D, moves):
    """Ensembles of >= 8 row tiles per CU take k_wide_lp_ws (four waves multiply two tiles each, four waves stage), ensembles
    of few row tiles and more than 128 columns k_wide_lp_ms (the macro blocks of a tile on different waves); the same
    contraction order as k_wide_lp, so the chains agree bit for bit -- and, at ndim 64, with the fused kernel too."""
    spec = _spec(N, D, moves, seed=11)
    outs = []
    for wide in ((0, 1, 2) if D <= 112 else (1, 2)):
        ens = make_ens(spec, spec["p0"])
        ens.set_tuning("dense_wide", wide)
        ens.eval_state_log_prob()
        lp0 = ens.get_state()[1]
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(31, 0)
        ens.run(4, 1, False)
        assert ens.status() == 0
        x, lp = ens.get_state()
        outs.append((lp0, x, lp, ens.accepted_mask().copy()))
        ens.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)
    assert outs[0][3].sum() > 0


@pytest.mark.parametrize("N,D,moves,weights", [
    (4096, 128, [_S("stretch")], None),
    (4100, 112, [_S("stretch")], None),                    # a ragged last tile
    (2000, 96, [_S("stretch", nsplits=3)], None),
    (1536, 66, [_S("stretch")], None),                     # padded 80
    (3000, 128, [_S("de")], None),
    (2048, 100, [_S("stretch"), _S("de")], [0.5, 0.5]),
    (65536, 128, [_S("stretch")], None),                   # the shape of the bench's dense_65536x128_fused entry
])
@pytest.mark.parametrize("store", [True, False])
def test_slab_kernel_equals_per_tile_kernel_bit_for_bit(N, D, moves, weights, store):
    """padded ndim 80 ... 128, even ndim: k_halfstep_slab (csrc/emx_slab.hip: the tile's proposals in registers, a 32-column LDS
    slab, eight waves a CU) against k_halfstep<16, 2, 4, MOVE, DPB, 1> (tuning key slab = 0): same arithmetic in the same order,
    so coordinates, log-probs, accept masks, chain rows and accept counters must agree bit for bit -- with and without the skewed
    start of the second wave of every SIMD (tuning key slab_skew, round 5)"""
    spec = _spec(N, D, moves, weights)
    outs = []
    for slab, skew in ((2, 1), (2, 0), (0, 0)):            # 2: the slab form from padded ndim 80 on (by default it starts at 112)
        ens = make_ens(spec, spec["p0"])
        ens.set_tuning("slab", slab)
        ens.set_tuning("slab_skew", skew)
        ens.set_tuning("small_kernel", 0)
        ens.set_tuning("graph", 0)
        ens.eval_state_log_prob()
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(424242, 0)
        nst = 9 if N <= 8192 else 5
        ens.chain_config(nst)
        ens.run(nst, 1, store)
        assert ens.status() == 0
        x, lp = ens.get_state()
        rec = dict(x=x, lp=lp, acc=ens.accepted_mask().copy())
        if store:
            rec.update(chain=ens.chain_read(0, 0, nst), clp=ens.chain_read(1, 0, nst), cnt=ens.accepted_counts())
        outs.append(rec)
        ens.close()
    a, a0, b = outs
    assert a["acc"].any() and not a["acc"].all()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a0[key], b[key]), key


