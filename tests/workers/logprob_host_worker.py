"""One rank of the two-process test of exchange='logprob' with a PYTHON log_prob_fn (tests/test_gpu_logprob_host.py).

Launched by torch.distributed.run with 2 processes that both use cuda:0: every process runs the whole sampler (proposal,
accept and commit on its own device context, replicated RNG stream), calls log_prob_fn on its half of the proposals only and
gathers the other half over the process group -- the reference's pool.map model (ensemble.py:486-496) with ranks for workers.
Each rank compares its chain with a plain single-process sampler it runs itself."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CALLS = {"rows": 0}


def main():
    import torch.distributed as dist
    import emcee_amd

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, D, nst = 96, 6, 25
    rs = np.random.RandomState(5)
    mu = rs.randn(D)
    A = rs.randn(D, D)
    icov = np.linalg.inv(A @ A.T / D + 0.3 * np.eye(D))
    p0 = mu + rs.randn(N, D)

    def log_prob(x):                      # vectorised Python callable with a blob per walker
        CALLS["rows"] += len(x)
        d = x - mu
        lp = -0.5 * np.einsum("ij,jk,ik->i", d, icov, d)
        return np.column_stack([lp, np.sum(x, axis=1)])

    def run(**kw):
        np.random.seed(1234)
        s = emcee_amd.EnsembleSampler(N, D, log_prob, vectorize=True, device=0,
                                      moves=[(emcee_amd.moves.StretchMove(), 0.6), (emcee_amd.moves.DEMove(), 0.4)], **kw)
        s.run_mcmc(p0, nst)
        return s.get_chain(), s.get_log_prob(), s.get_blobs(), s.acceptance_fraction

    CALLS["rows"] = 0
    ref = run()
    rows_single = CALLS["rows"]
    CALLS["rows"] = 0
    got = run(distributed=True, exchange="logprob")
    rows_shared = CALLS["rows"]
    ok = all(np.array_equal(a, b) for a, b in zip(ref, got))
    ok = ok and rows_single == N * (nst + 1) and rows_shared * world == rows_single
    dist.barrier()
    print("LOGPROB_HOST rank %d rows %d of %d %s" % (rank, rows_shared, rows_single, "OK" if ok else "MISMATCH"), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
