"""One rank of the two-process device-side REPLAY exchange test (tests/test_gpu_direct_ipc.py).

Launched by torch.distributed.run with 2 processes that BOTH use cuda:0 (the GPU box has one GPU): every process updates its
share of each split, STORES its decisions into the other process's receive buffer -- mapped through hipIpc handles exchanged
over gloo -- meets it at the device-side barrier (no host synchronisation between the processes inside the loop) and replays
the other's accepted updates.  Each rank checks its WHOLE replica of the chain against a single-context run it makes itself."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    from emcee_amd import _lib
    from emcee_amd.device import DeviceEnsemble
    from emcee_amd.parallel import block_range, import_direct_peers

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, D, nst = 4096, 16, 12
    rs = np.random.RandomState(3)
    p0 = rs.randn(N, D)
    mu = rs.randn(D)
    A = rs.randn(D, D)
    icov = np.linalg.inv(A @ A.T / D + 0.2 * np.eye(D))
    icov = 0.5 * (icov + icov.T)
    moves = [_lib.MoveDesc(0, 2, 1, 0, 2.0, 1e-5, 0.3, 1.7), _lib.MoveDesc(1, 2, 1, 0, 2.0, 1e-5, 0.3, 1.7),
             _lib.MoveDesc(2, 4, 1, 0, 2.0, 1e-5, 0.3, 1.7)]
    cdf = np.array([0.5, 0.8, 1.0])

    def make():
        ens = DeviceEnsemble(N, D, device=0)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        ens.set_moves(moves, cdf)
        ens.set_rng_mode(_lib.RNG_PHILOX)
        ens.set_philox(424242, 0)
        ens.set_state(p0)
        ens.eval_state_log_prob()
        ens.chain_config(nst)
        return ens

    ref = make()
    ref.run(nst, 1, True)
    ref_chain, ref_lp = ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst)
    ref.close()

    ens = make()
    ens.set_exchange("replay")
    ens.set_shard(rank, world)
    ens.set_tuning("direct_timeout_ms", 4000)
    import_direct_peers(ens, dist)            # IPC handles over gloo, hipIpcOpenMemHandle in every process
    dist.barrier()
    for _ in range(nst):
        k, S = ens.step_begin(True)
        for split in range(S):
            ens.replay_begin(split)
            ens.replay_exchange(split)
            ens.replay_finish(split)
        ens.step_end()
    ens.sync()
    status = ens.status()
    lo, hi = block_range(N, rank, world)
    ok = status == 0 and np.array_equal(ens.chain_read(0, 0, nst), ref_chain) and np.array_equal(ens.chain_read(1, 0, nst), ref_lp)
    dist.barrier()                            # nobody unmaps while a peer may still read
    print("REPLAY_IPC rank %d status %d block [%d, %d) %s" % (rank, status, lo, hi, "OK" if ok else "MISMATCH"), flush=True)
    ens.close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
