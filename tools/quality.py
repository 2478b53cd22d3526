"""Sampling-quality parity: acceptance fraction and integrated autocorrelation time (reference
autocorr.integrated_time, c=5) of the 64-dim correlated Gaussian, StretchMove a=2, for
(a) reference emcee on the CPU (build container: --ref) and (b) emcee_amd on the GPU (--gpu).
Same target, same start, same number of walkers / steps / thin_by; different RNG streams, so the
comparison is statistical (within 2 % on mean acceptance and mean tau)."""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def dense_gaussian(ndim, seed=0):
    rs = np.random.RandomState(seed)
    mu = rs.randn(ndim)
    A = rs.randn(ndim, ndim)
    cov = A @ A.T / ndim + 0.1 * np.eye(ndim)
    icov = np.linalg.inv(cov)
    return mu, cov, 0.5 * (icov + icov.T)


def summarize(chain, acc, thin_by, label, dt, nprop):
    # chain (nstored, N, D); tau via the reference estimator, in steps
    sys.path.insert(0, ".")
    from emcee_amd import autocorr
    tau = thin_by * autocorr.integrated_time(chain, quiet=True)
    return {"label": label, "accept_mean": float(np.mean(acc)), "tau_mean": float(np.mean(tau)),
            "tau_min": float(np.min(tau)), "tau_max": float(np.max(tau)), "nsteps": int(nprop),
            "chain_over_tau": float(nprop / np.mean(tau)), "seconds": dt,
            "mean_abs_err_of_mean": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", action="store_true")
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--nwalkers", type=int, default=2048)
    ap.add_argument("--ndim", type=int, default=64)
    ap.add_argument("--nsteps", type=int, default=4000, help="stored steps")
    ap.add_argument("--thin-by", type=int, default=5)
    ap.add_argument("--burn", type=int, default=2000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--moves", default="stretch", choices=["stretch", "mix"], help="mix = DEMove 0.8 + DESnookerMove 0.2 (BASELINE config 4)")
    a = ap.parse_args()
    N, D = a.nwalkers, a.ndim
    mu, cov, icov = dense_gaussian(D)
    p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
    res = []
    if a.ref:
        from oracle import ref_shim
        emcee = ref_shim.import_reference()
        lp = lambda x: -0.5 * np.einsum("ij,ij->i", (x - mu) @ icov, x - mu)  # noqa: E731
        mv = None if a.moves == "stretch" else [(emcee.moves.DEMove(), 0.8), (emcee.moves.DESnookerMove(), 0.2)]
        s = emcee.EnsembleSampler(N, D, lp, vectorize=True, moves=mv)
        s._random.seed(11)
        t0 = time.time()
        st = s.run_mcmc(p0, a.burn, skip_initial_state_check=True, store=False)
        s.run_mcmc(st, a.nsteps, thin_by=a.thin_by, skip_initial_state_check=True)
        res.append(summarize(s.get_chain(), s.acceptance_fraction, a.thin_by, "reference emcee (CPU, MT19937)", time.time() - t0,
                             a.nsteps * a.thin_by))
    if a.gpu:
        import emcee_amd
        for rng in ("philox", "mt19937"):
            mv = None if a.moves == "stretch" else [(emcee_amd.moves.DEMove(), 0.8), (emcee_amd.moves.DESnookerMove(), 0.2)]
            s = emcee_amd.EnsembleSampler(N, D, emcee_amd.targets.DenseGaussian(mu, icov), rng=rng, moves=mv)
            s._random.seed(12)
            t0 = time.time()
            st = s.run_mcmc(p0, a.burn, skip_initial_state_check=True, store=False)
            s.run_mcmc(st, a.nsteps, thin_by=a.thin_by, skip_initial_state_check=True)
            res.append(summarize(s.get_chain(), s.acceptance_fraction, a.thin_by, "emcee_amd (MI355X, %s)" % rng,
                                 time.time() - t0, a.nsteps * a.thin_by))
    for r in res:
        print(json.dumps(r))
    if a.out:
        json.dump({"config": vars(a), "results": res}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
