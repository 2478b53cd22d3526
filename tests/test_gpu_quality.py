"""Sampling quality against reference emcee: mean acceptance fraction and mean integrated autocorrelation time
(reference estimator, c = 5) within 2 % (BASELINE.json north_star; BASELINE.md section 3.4: runs >= 50 tau long).

The reference side ran in the build container (`tools/quality.py --ref`, the generating script; /root/reference does
not exist on the GPU box) and its numbers are committed as tests/golden/quality_ref.json / quality_mix_ref.json.
The GPU side runs here with the same target, start, ensemble size, burn-in, length and thinning.  The streams differ
(different seeds, and Philox is a different generator altogether), so the comparison is statistical: with 2048 walkers x
50 000 steps the standard error of either mean tau is well under 1 %."""
import json
import os

import numpy as np
import pytest

import emcee_amd

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dense_gaussian(ndim, seed=0):
    rs = np.random.RandomState(seed)
    mu = rs.randn(ndim)
    A = rs.randn(ndim, ndim)
    cov = A @ A.T / ndim + 0.1 * np.eye(ndim)
    icov = np.linalg.inv(cov)
    return mu, cov, 0.5 * (icov + icov.T)


def run_gpu(cfg, rng):
    N, D = cfg["nwalkers"], cfg["ndim"]
    mu, cov, icov = dense_gaussian(D)
    p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
    mv = None if cfg["moves"] == "stretch" else [(emcee_amd.moves.DEMove(), 0.8), (emcee_amd.moves.DESnookerMove(), 0.2)]
    s = emcee_amd.EnsembleSampler(N, D, emcee_amd.targets.DenseGaussian(mu, icov), rng=rng, moves=mv)
    s._random.seed(12)
    st = s.run_mcmc(p0, cfg["burn"], skip_initial_state_check=True, store=False)
    s.run_mcmc(st, cfg["nsteps"], thin_by=cfg["thin_by"], skip_initial_state_check=True)
    tau = cfg["thin_by"] * s.get_autocorr_time(quiet=True)          # device-resident chain: batched FFTs next to it
    return float(np.mean(s.acceptance_fraction)), tau, s


@pytest.mark.parametrize("fixture,rng", [("quality_ref.json", "philox"), ("quality_ref.json", "mt19937"),
                                         ("quality_mix_ref.json", "philox"), ("quality_mix_ref.json", "mt19937")])
def test_acceptance_and_autocorrelation_time_within_two_percent_of_reference(fixture, rng):
    ref = json.load(open(os.path.join(GOLDEN, fixture)))
    cfg, r = ref["config"], ref["results"][0]
    assert r["chain_over_tau"] >= 50.0, "the reference run itself must span >= 50 tau"
    acc, tau, s = run_gpu(cfg, rng)
    nprop = cfg["nsteps"] * cfg["thin_by"]
    assert nprop / np.mean(tau) >= 50.0
    d_acc = acc / r["accept_mean"] - 1.0
    d_tau = float(np.mean(tau)) / r["tau_mean"] - 1.0
    print("%s %s: accept %.5f (ref %.5f, %+.2f %%), tau %.1f (ref %.1f, %+.2f %%), chain = %.0f tau"
          % (fixture, rng, acc, r["accept_mean"], 100 * d_acc, np.mean(tau), r["tau_mean"], 100 * d_tau, nprop / np.mean(tau)))
    assert abs(d_acc) < 0.02, d_acc
    assert abs(d_tau) < 0.02, d_tau
