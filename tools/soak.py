"""Soak run: long native-mode runs at several shapes; checks the sticky status, finiteness, that stored
log-probs equal the target at the stored coordinates, and that the acceptance fraction is stationary."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402
from emcee_amd import moves, targets  # noqa: E402


def dense(D, seed=0):
    rs = np.random.RandomState(seed)
    mu = rs.randn(D)
    A = rs.randn(D, D)
    cov = A @ A.T / D + 0.1 * np.eye(D)
    icov = np.linalg.inv(cov)
    return mu, cov, 0.5 * (icov + icov.T)


def run(label, N, D, target, mv, nsteps, thin_by, p0, check, rng="philox"):
    s = emcee_amd.EnsembleSampler(N, D, target, moves=mv, rng=rng)
    t0 = time.perf_counter()
    st = s.run_mcmc(p0, nsteps, thin_by=thin_by, skip_initial_state_check=True)
    dt = time.perf_counter() - t0
    chain, lp = s.get_chain(), s.get_log_prob()
    assert np.all(np.isfinite(chain)) and np.all(np.isfinite(lp)), label
    k = len(chain) - 1
    np.testing.assert_allclose(lp[k], check(chain[k]), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(st.log_prob, check(st.coords), rtol=1e-9, atol=1e-9)
    moved = np.any(chain[1:] != chain[:-1], axis=2).mean(axis=1)       # per stored step
    h = len(moved) // 2
    out = dict(label=label, N=N, D=D, proposals=nsteps * thin_by, seconds=dt, wu_per_s=N * nsteps * thin_by / dt,
               acc=float(np.mean(s.acceptance_fraction)), moved_first_half=float(moved[:h].mean()), moved_second_half=float(moved[h:].mean()),
               mean_abs=float(np.abs(chain[h:].mean(axis=(0, 1))).max()))
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    rs = np.random.RandomState(5)
    res = []
    res.append(run("c1 32x5 iso stretch", 32, 5, targets.IsoGaussian(), moves.StretchMove(), 4000, 50, rs.randn(32, 5),
                   lambda x: -0.5 * np.sum(x * x, axis=1)))
    mu, cov, icov = dense(64)
    p0 = mu + rs.randn(65536, 64) @ np.linalg.cholesky(cov).T
    dg = lambda x: -0.5 * np.einsum("ij,jk,ik->i", x - mu, icov, x - mu)  # noqa: E731
    res.append(run("c2 65536x64 dense stretch", 65536, 64, targets.DenseGaussian(mu, icov), moves.StretchMove(), 40, 500, p0, dg))
    res.append(run("c4 65536x64 dense de+snooker", 65536, 64, targets.DenseGaussian(mu, icov),
                   [(moves.DEMove(), 0.8), (moves.DESnookerMove(), 0.2)], 40, 250, p0, dg))
    res.append(run("gauss 65536x64 dense", 65536, 64, targets.DenseGaussian(mu, icov), moves.GaussianMove(0.0009), 40, 250, p0, dg))
    rosen = lambda x: -np.sum(100.0 * (x[:, 1:] - x[:, :-1] ** 2) ** 2 + (1 - x[:, :-1]) ** 2, axis=1) / 20.0  # noqa: E731
    res.append(run("c3 262144x32 rosenbrock", 262144, 32, targets.Rosenbrock(20.0), moves.StretchMove(), 20, 250,
                   1 + 0.1 * rs.randn(262144, 32), rosen))
    res.append(run("c2 65536x64 dense stretch, exact MT19937 mode (plan pipeline)", 65536, 64, targets.DenseGaussian(mu, icov),
                   moves.StretchMove(), 20, 250, p0, dg, rng="mt19937"))
    muw, covw, icovw = dense(256, 3)
    pw = muw + rs.randn(8192, 256) @ np.linalg.cholesky(covw).T
    dgw = lambda x: -0.5 * np.einsum("ij,jk,ik->i", x - muw, icovw, x - muw)  # noqa: E731
    res.append(run("wide dense 8192x256 stretch", 8192, 256, targets.DenseGaussian(muw, icovw), moves.StretchMove(), 20, 250, pw, dgw))
    json.dump(res, open("gpurun_out/soak.json", "w"), indent=1)
