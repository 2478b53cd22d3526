"""One rank of the two-process test of EnsembleSampler(..., distributed=True, exchange="replay_push") (tests/test_gpu_direct_ipc.py).

Launched by torch.distributed.run with 2 processes that both use cuda:0: every rank builds the same sampler, `run_mcmc` is the
sharded emx_run -- each rank updates its share of every split, the decisions go into the other rank's buffer through hipIpc, the
accepted updates are replayed -- and every rank must end with the chain of the single-process sampler, stored steps included
(replicas stay complete under this exchange).  Device targets: a closed-form one and a torch callable."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import emcee_amd
    from emcee_amd import moves, targets

    dist.init_process_group("gloo")
    rank = dist.get_rank()
    N, D, nst = 2048, 8, 14
    rs = np.random.RandomState(5)
    mu = rs.randn(D)
    A = rs.randn(D, D)
    icov = np.linalg.inv(A @ A.T / D + 0.3 * np.eye(D))
    icov = 0.5 * (icov + icov.T)
    p0 = mu + rs.randn(N, D)
    mu_t, icov_t = torch.as_tensor(mu, device="cuda"), torch.as_tensor(icov, device="cuda")

    def lp_torch(q):
        d = q - mu_t
        return -0.5 * ((d @ icov_t) * d).sum(1)

    ok = True
    for label, mk in (("dense", lambda: targets.DenseGaussian(mu, icov)), ("callable", lambda: targets.DeviceCallable(lp_torch))):
        def run(**kw):
            np.random.seed(4321)
            s = emcee_amd.EnsembleSampler(N, D, mk(), device=0, rng="mt19937",
                                          moves=[(moves.StretchMove(), 0.5), (moves.DEMove(), 0.3), (moves.DESnookerMove(), 0.2)], **kw)
            st = s.run_mcmc(p0, nst, skip_initial_state_check=True)
            st = s.run_mcmc(st, 5, thin_by=2, skip_initial_state_check=True)
            return s.get_chain(), s.get_log_prob(), s.acceptance_fraction, np.asarray(st.coords)
        ref = run()
        got = run(distributed=True, exchange="replay_push")
        if label == "dense":
            same = all(np.array_equal(a, b) for a, b in zip(ref, got))
        else:
            # a torch callable's rounding may depend on the size of the block it is handed (the library behind `@` picks its
            # tiling by shape), and a rank hands it its SHARE of a split: the log-probs agree to rounding, the decisions -- hence
            # the coordinates -- exactly
            same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[2], got[2]) and np.array_equal(ref[3], got[3]) and \
                np.allclose(ref[1], got[1], rtol=1e-12, atol=1e-13)
        if not same:
            print("SAMPLER_REPLAY rank %d %s: chain equal %s (max |d| %.3g), log-prob max |d| %.3g, acceptance equal %s" % (
                rank, label, np.array_equal(ref[0], got[0]), float(np.max(np.abs(ref[0] - got[0]))), float(np.max(np.abs(ref[1] - got[1]))),
                np.array_equal(ref[2], got[2])), flush=True)
        print("SAMPLER_REPLAY rank %d %s %s" % (rank, label, "OK" if same else "MISMATCH"), flush=True)
        ok = ok and same
        dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
