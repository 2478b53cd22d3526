"""One rank of the device-side exchange tests between PROCESSES (tests/test_gpu_sharded.py::test_*_device_side*).

Launched by torch.distributed.run with `world` processes that all use cuda:0 (the GPU box has one GPU).  Every process is one
rank -- the deployment's shape: one process, one context, one stream per rank -- so the barrier kernels of different ranks can
never sit in one hardware queue behind each other (what contexts of ONE process can do: the runtime multiplexes a process's
streams onto a few queues).  A barrier that is not met is therefore a FAILURE here, never a skip.

Cases come in EMX_DS_CASES as "exchange:fixture:rng,..." (exchange = direct | replay; fixture = a name under tests/golden;
rng = mt | philox); every rank runs the single-context reference itself and compares what it owns:
direct -- its walker block of the chain and the log-probs, replay -- the whole replica, accept counts included.
Semantics held: split k + 1 sees split k's commits (reference moves/red_blue.py:85,104)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    from emcee_amd import _lib
    from emcee_amd.parallel import block_range, import_direct_peers
    from oracle import cases
    from helpers import load_golden, rng_from_fixture
    from test_gpu_parity import make_ens

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    todo = [c.split(":") for c in os.environ["EMX_DS_CASES"].split(",") if c]
    bad = 0
    for exchange, name, rng in todo:
        g = load_golden(name)
        spec = cases.build(name)
        nst = min(8, spec["nsteps"])

        def make():
            ens = make_ens(spec, g["p0"])
            if rng == "mt":
                ens.set_rng_mode(_lib.RNG_MT19937)
                ens.set_mt19937(rng_from_fixture(g).get_state())
            else:
                ens.set_rng_mode(_lib.RNG_PHILOX)
                ens.set_philox(777, 0)
            ens.set_tuning("small_kernel", 0)
            ens.chain_config(nst)
            return ens

        ref = make()
        ref.run(nst, 1, True)
        ref_chain, ref_lp, ref_acc = ref.chain_read(0, 0, nst), ref.chain_read(1, 0, nst), ref.accepted_counts()
        ref.close()

        ens = make()
        ens.set_exchange(exchange)
        ens.set_shard(rank, world)
        ens.set_tuning("direct_timeout_ms", 6000)
        import_direct_peers(ens, dist)            # IPC handles over gloo, hipIpcOpenMemHandle in every process
        dist.barrier()
        for _ in range(nst):
            k, S = ens.step_begin(True)
            for split in range(S):
                if exchange == "direct":
                    ens.direct_halfstep(split, barrier=True)
                else:
                    ens.replay_begin(split)
                    ens.replay_exchange(split)     # push + barrier kernels: the ranks meet on the device
                    ens.replay_finish(split)
            ens.step_end()
        ens.sync()
        status = ens.status()
        N = ref_chain.shape[1]
        lo, hi = block_range(N, rank, world) if exchange == "direct" else (0, N)
        ok = status == 0 and np.array_equal(ens.chain_read(0, 0, nst)[:, lo:hi], ref_chain[:, lo:hi]) \
            and np.array_equal(ens.chain_read(1, 0, nst)[:, lo:hi], ref_lp[:, lo:hi]) \
            and np.array_equal(ens.accepted_counts()[lo:hi], ref_acc[lo:hi])
        dist.barrier()                            # nobody unmaps while a peer may still read
        print("DEVICE_SIDE %s %s %s world %d rank %d status %d rows [%d, %d) %s"
              % (exchange, name, rng, world, rank, status, lo, hi, "OK" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
        ens.close()
        dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
