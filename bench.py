#!/usr/bin/env python
"""Contract bench: walker-updates/sec of the fused red/blue stretch-move step on MI355X.

`python bench.py --gpus N --steps K --warmup W`  (N>1: launched by torch.distributed.run, one
rank per GPU).  A "step" is one full ensemble step = both half-steps of the stretch move over
one ensemble of synthetic walkers resident in HBM (BASELINE.json configs[1]:
nwalkers=65536 per GPU, ndim=64, correlated Gaussian with dense precision matrix, a=2.0).
Weak scaling: the ensemble grows with N (65536 walkers per GPU).  Two exchange protocols exist
(emcee_amd/parallel.py): every rank updates a slot range and one all-gather per half-step replicates
all updated rows; or every rank owns a walker block and one all-to-all per half-step moves only the
partner rows that are read.  At N>1 both are measured (same seed: their final states must agree) and
the faster is reported; `exchange` in the JSON line carries both.

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields, incl. `roofline` and
`cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WALKERS_PER_GPU = 65536
NDIM = 64
HBM_PEAK_GBPS = 8000.0      # MI355X spec (MI355X_MICROARCH.md); ~6300 achievable


def dense_gaussian(ndim, seed=0):
    """Sigma = A A^T / D + 0.1 I, dense Sigma^-1."""
    rs = np.random.RandomState(seed)
    mu = rs.randn(ndim)
    A = rs.randn(ndim, ndim)
    cov = A @ A.T / ndim + 0.1 * np.eye(ndim)
    icov = np.linalg.inv(cov)
    return mu, cov, 0.5 * (icov + icov.T)


def initial_walkers(n, mu, cov, seed=1):
    rs = np.random.RandomState(seed)
    return mu + rs.randn(n, len(mu)) @ np.linalg.cholesky(cov).T   # equilibrium start


def cpu_baseline(mu, cov, icov, budget_s=15.0):
    """NumPy oracle (port of reference emcee's vectorised path) on the host cores, bounded sample."""
    from oracle import sampler_oracle as so
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # noqa: BLE001
        threadpool_limits = None
    n = WALKERS_PER_GPU
    p0 = initial_walkers(n, mu, cov)
    fn = lambda x: so.dense_gauss(x, mu, icov)  # noqa: E731
    rs = np.random.RandomState(7)

    def go():
        out = so.run(p0, 1, fn, rs, store=False)                  # warm-up (page faults, BLAS initialisation)
        t0 = time.perf_counter()
        out = so.run(out["coords"], 2, fn, rs, store=False, log_prob0=out["lp"])
        t1 = (time.perf_counter() - t0) / 2
        nst = int(min(1000, max(3, budget_s / max(t1, 1e-3))))     # ~15 s of CPU work whatever the host
        t0 = time.perf_counter()
        so.run(out["coords"], nst, fn, rs, store=False, log_prob0=out["lp"])
        return nst, time.perf_counter() - t0

    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            nst, dt = go()
        cores = 1
    else:
        nst, dt = go()
        cores = os.cpu_count()
    return {"value": n * nst / dt, "unit": "walker-updates/s", "cores": cores, "kind": "port",
            "sample": "oracle/sampler_oracle.py (NumPy restatement of emcee's vectorize=True path), "
                      "%d steps of the same 65536x64 dense-Gaussian stretch workload, %.1f s, BLAS threads=%d, host has %d cores"
                      % (nst, dt, cores, os.cpu_count())}


# The contract is ONE JSON line on stdout.  Libraries in the process write there too (gloo announces its mesh, RCCL its
# version ...), so file descriptor 1 is pointed at stderr for the whole run and the line goes to a private copy of the
# original stdout.
_REAL_STDOUT = None


def _claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _REAL_STDOUT


def main():
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--store", action="store_true", help="append every step to the device chain")
    ap.add_argument("--rng", default="philox", choices=["philox", "mt19937"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded RCCL path even at world size 1 (testing)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="sharded runs: collectives enqueued by libemx itself (default) or torch.distributed")
    ap.add_argument("--exchange", default="both", choices=["both", "allgather", "pull"],
                    help="sharded runs: all-gather of every updated row, all-to-all of the partner rows read "
                         "(emcee_amd/parallel.py), or measure both and report the faster (default)")
    ap.add_argument("--pull-timeout", type=float, default=150.0,
                    help="'both': seconds after which a stuck pull measurement is abandoned for the all-gather result")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    K, W = args.steps, args.warmup

    import torch
    from emcee_amd import _lib
    from emcee_amd.device import DeviceEnsemble

    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_dist
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.comm == "torch":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")      # bootstrap / barriers only; the data path is libemx -> RCCL

    n = WALKERS_PER_GPU * world
    mu, cov, icov = dense_gaussian(NDIM)
    p0 = initial_walkers(n, mu, cov)

    def measure(exchange):
        """One full measurement (fresh context): spin-up, W warm-up steps, K timed steps."""
        ens = DeviceEnsemble(n, NDIM, device=local_rank)
        ens.set_target(_lib.TARGET_DENSE, mu, icov)
        ens.set_moves([_lib.MoveDesc(_lib.MOVE_STRETCH, 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * NDIM), 1.7)], np.array([1.0]))
        if args.rng == "philox":
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(20260923, 0)
        else:
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(np.random.RandomState(20260923).get_state())
        ens.set_state(p0)
        ens.eval_state_log_prob()
        if args.store:
            ens.chain_config(K + W)
        if sharded:
            ens.set_exchange(exchange)

        def torch_path(group=None):
            from emcee_amd.parallel import DeviceEngine, PullStepper, ShardedStepper
            ens.set_stream(torch.cuda.current_stream().cuda_stream)   # kernels + RCCL ordered on one stream
            eng = DeviceEngine(ens, rank, world, torch.device("cuda", local_rank), exchange=exchange)
            gather = lambda out, inp: dist.all_gather_into_tensor(out, inp, group=group)  # noqa: E731
            if exchange == "pull":
                stepper = PullStepper(eng, lambda out, inp: dist.all_to_all_single(out, inp, group=group), gather)
            else:
                stepper = ShardedStepper(eng, gather)
            return lambda k, st=None: stepper.run(k, 1, args.store if st is None else st)

        comm_used = None
        if sharded and args.comm == "torch":
            run = torch_path()
            comm_used = "torch.distributed(nccl)"
        elif sharded:
            ok = 1
            try:
                uid = [DeviceEnsemble.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                ens.comm_init(rank, world, uid[0])      # ncclCommInitRank; emx_run now exchanges per half-step
            except Exception as e:  # noqa: BLE001
                ok = 0
                print("[bench] library-driven RCCL unavailable on rank %d (%s); falling back to torch.distributed" % (rank, e),
                      file=sys.stderr)
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 1:
                run = lambda k, st=None: ens.run(k, 1, args.store if st is None else st)  # noqa: E731
                comm_used = "libemx->RCCL"
            else:
                try:
                    ens.comm_destroy()
                except Exception:  # noqa: BLE001
                    pass
                run = torch_path(dist.new_group(backend="nccl"))
                comm_used = "torch.distributed(nccl) [fallback]"
        else:
            run = lambda k, st=None: ens.run(k, 1, args.store if st is None else st)  # noqa: E731

        def fence():
            ens.sync()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        # Untimed spin-up before the contract's W warm-up steps: the first ~50 ms of work on a fresh context run
        # slower (clock ramp, first touch of the plan ring, lazy code-object loading); tools/stall_probe.py.
        if sharded:
            for _ in range(10):            # a FIXED count: every rank must issue the same collectives
                run(5, False)
                ens.sync()
        else:
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < 0.15:
                run(50, False)
                ens.sync()
        run(W)
        fence()
        ens.timer_start()
        t0 = time.perf_counter()
        run(K)
        gpu_ms = ens.timer_stop()          # hipEvents on the stream the kernels are launched on
        fence()
        wall = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([wall, gpu_ms], dtype=torch.float64, device="cuda" if args.comm == "torch" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, gpu_ms = float(t[0]), float(t[1])

        res = {"wall": wall, "gpu_ms": gpu_ms, "comm": comm_used, "exchange": exchange if sharded else None,
               "accept_frac": float(ens.accepted_mask().mean()), "status": ens.status(), "per_launch_us": None, "digest": None}
        if sharded:
            # every rank must hold the same ensemble, whatever the exchange: a checksum of the state
            x, lp = ens.get_state()
            res["digest"] = "%.17g/%.17g" % (float(np.sum(x * np.arange(1, NDIM + 1))), float(np.sum(lp)))
            every = [None] * world
            dist.all_gather_object(every, res["digest"])
            res["replicas_agree"] = len(set(every)) == 1
        else:
            # per-launch hipEvent durations of the half-step kernel (separate pass: event records perturb)
            ens.profile_enable(128)
            ens.run(64, 1, False)
            pl = ens.profile_read(128)
            if len(pl):
                res["per_launch_us"] = float(np.median(pl) * 1e3)
        if sharded and comm_used == "libemx->RCCL":
            ens.comm_destroy()
        ens.close()
        return res

    def emit(res, extra=None):
        nsplits = 2
        wall, gpu_ms = res["wall"], res["gpu_ms"]
        launches = K * nsplits
        slots_per_launch = WALKERS_PER_GPU // nsplits                 # per GPU
        B = 24 * NDIM + 17 + ((8 * NDIM + 8) if args.store else 0)    # algorithmic bytes / walker-update
        avg_launch_s = gpu_ms * 1e-3 / launches                        # timed-region events / launches
        achieved = slots_per_launch * B / avg_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("c2_stretch_dense_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        how = ""
        if sharded:
            how = ", %s via %s" % ("all-to-all of the partner rows (pull exchange)" if res["exchange"] == "pull"
                                   else "all-gather of the updated rows", res["comm"])
        line = {
            "metric": "walker-updates/sec (whole node), 64-dim correlated Gaussian, StretchMove a=2",
            "value": n * K / wall, "unit": "walker-updates/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: nwalkers=%d (65536/GPU), ndim=64, dense-precision Gaussian, "
                                   "StretchMove a=2.0, nsplits=2, rng=%s, store=%s" % (n, args.rng, args.store),
                       "nwalkers": n, "ndim": NDIM, "parallelism": "walker-sharded x%d%s" % (world, how)},
            "steps_per_s": K / wall, "accept_frac": res["accept_frac"], "device_status": res["status"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "emx::k_halfstep<8,2,4,STRETCH,DPB=4> (G=8 lanes/walker, V=2, CH=4; f64 MFMA dense target)",
                         "algorithmic_bytes_per_walker_update": B, "walker_updates_per_launch": slots_per_launch,
                         "avg_launch_us": avg_launch_s * 1e6, "per_launch_event_us": res["per_launch_us"],
                         "note": "avg_launch_us = hipEvent time of the timed region / half-step launches: it includes the "
                                 "inter-kernel gaps and the batched plan kernel (k_native_plan_batch, 1 launch per 8 steps)"
                                 + (" and, on sharded runs, the exchange" if sharded else "") +
                                 "; per_launch_event_us brackets single half-step launches with hipEvents"},
        }
        if extra:
            line["exchange"] = extra
        if world == 1 and not sharded and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(mu, cov, icov)
        else:
            line["cpu_baseline"] = None
        out = _claim_stdout()
        out.write(json.dumps(line) + "\n")
        out.flush()

    def summary(r):
        return {"ms_per_step": r["wall"] * 1e3 / K, "value": n * K / r["wall"], "comm": r["comm"], "device_status": r["status"],
                "replicas_agree": r.get("replicas_agree")}

    if not sharded:
        res = measure(None)
        emit(res)
    elif args.exchange != "both":
        res = measure(args.exchange)
        if rank == 0:
            emit(res)
    else:
        # Both protocols, identical seed and step count: the all-gather result is in hand before the pull
        # exchange is tried, and a watchdog falls back to it if that attempt does not come back.
        res_ag = measure("allgather")
        import threading
        done = threading.Event()

        def bail():
            if done.is_set():
                return
            if rank == 0:
                emit(res_ag, {"allgather": summary(res_ag), "pull": "no result after %.0f s" % args.pull_timeout,
                              "reported": "allgather"})
            sys.stdout.flush()
            os._exit(0)

        timer = threading.Timer(args.pull_timeout, bail)
        timer.daemon = True
        timer.start()
        res_pull, err = None, None
        try:
            res_pull = measure("pull")
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        done.set()
        timer.cancel()
        if rank == 0:
            extra = {"allgather": summary(res_ag)}
            best = res_ag
            if res_pull is None:
                extra["pull"] = "failed: %s" % err
            else:
                extra["pull"] = summary(res_pull)
                same = res_pull["digest"] == res_ag["digest"] and res_pull["status"] == 0 and bool(res_pull.get("replicas_agree"))
                extra["same_final_state"] = same
                if same and res_pull["wall"] < res_ag["wall"]:
                    best = res_pull
            extra["reported"] = best["exchange"]
            emit(best, extra)
        if res_pull is None:          # a rank that failed must not leave the others in a collective
            sys.stdout.flush()
            os._exit(0 if rank == 0 else 1)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
