"""Test double for bench.py's GPU legs (EMX_BENCH_STUB=tests.stubs.bench_stub): `run_child` -- the one function of the N > 1
orchestrator that starts a GPU process -- answers from here.  Everything else of the N > 1 path runs for real: the launcher,
the rank processes, their gloo rendezvous, the agreement collectives, the line rank 0 prints."""
import json
import os
import time


def install(bench):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    outdir = os.environ.get("EMX_BENCH_STUB_DIR")
    fail = set(filter(None, os.environ.get("EMX_BENCH_STUB_FAIL", "").split(",")))          # exchanges whose preflight fails
    hang = set(filter(None, os.environ.get("EMX_BENCH_STUB_HANG", "").split(",")))          # ... or never answers
    short = os.environ.get("EMX_BENCH_STUB_RANKS")                                          # census lie: RCCL saw fewer ranks
    speed = {"allgather": 4.0, "pull": 2.0, "direct": 1.0, "replay": 0.8, "replay_push": 0.7, "logprob": 3.0}

    def run_child(args, key, ex, port, timeout_s, keep_partial=False):
        if outdir:
            with open(os.path.join(outdir, "rank%d.log" % rank), "a") as f:
                f.write(json.dumps({"rank": rank, "world": world, "key": key, "ex": ex, "local_rank": os.environ.get("LOCAL_RANK")}) + "\n")
        if key == "preflight":
            done = {"p2p": {"ok": True, "devices_visible": world}}
            for e in ex.split(","):
                if e in hang:
                    return {"error": "no result within %.0f s (hung; child killed)" % timeout_s, "preflight": done}
                done[e] = {"ok": e not in fail, "seconds": 0.01, "device_status": 0, "replicas_agree": e not in fail}
            return {"preflight": done}
        time.sleep(float(os.environ.get("EMX_BENCH_STUB_SLEEP", "0.01")))               # (a slow child: the time-budget test)
        wall = 1e-3 * speed.get(ex, 1.0) * args.steps
        return {"wall_s": wall, "gpu_ms": wall * 1e3, "blocks": 3, "comm": "stub", "exchange": ex, "accept_frac": 0.17, "status": 0,
                "digest": "stub-digest", "replicas_agree": True, "rccl_ranks": int(short) if short else world,
                "distinct_devices": world}

    bench.run_child = run_child
    bench._sharded.run_child = run_child          # (where the orchestrator looks it up: tools/benchkit/sharded.py)


    if os.environ.get("EMX_BENCH_STUB_SINGLE"):
        # N = 1: the measurement itself is replaced (no GPU), everything that builds the record and the line is the real code
        import numpy as np
        from tools.benchkit import single as _single
        real_wl = bench.Workload

        def cheap_workload(key, n, make_p0=True):
            return real_wl(key, n, make_p0=False)          # the description only: no gigabytes of start state

        def measure_single(wl, K, W, device=0, rng="philox", store=False, single_block=False, want_kernel=True, spin_s=0.15, tuning=None):
            wall = 21.0e-6 * K * (wl.N * wl.D / (65536.0 * 64.0)) * (2.2 if rng == "mt19937" else 1.0)
            hpl = 32.0 if (wl.key in ("c2", "c4") and wl.N <= 65536 and rng == "philox") else 1.0
            res = {"wall_s": wall, "gpu_ms": wall * 1e3 * 0.977, "blocks": 123, "wall_min_s": wall * 0.99, "accept_frac": 0.3456789012,
                   "status": 0, "per_launch_us": 190.8 if want_kernel else None, "walls_s": [wall] * 123, "halfsteps_per_launch": hpl,
                   "per_launch_halfsteps": 32.0, "persist_total": {"launches": 900, "halfsteps": 18000, "gaveup": 0, "p2p": 0}}
            if rng == "mt19937":
                res["pipeline"] = {("stage%d_us_per_step" % i): 40.123456 + i for i in range(24)}
                res["mtdev"] = {"steps": 50, "tokenizer": {"windows": 100, "rounds": 230, "kernel_us": 9000.0}}
            return res

        def quality_entry(device, rng="philox"):
            return {"workload": "stub " * 30, "accept": 0.3123456, "tau_mean": 57.123456, "tau_min": 50.0, "tau_max": 61.0, "nsteps_over_tau": 68.1,
                    "run_seconds": 3.0, "tau_seconds": 0.2, "reference": {"source": "stub " * 30, "accept": 0.3126, "tau_mean": 56.9},
                    "accept_rel_diff": -0.001, "tau_rel_diff": 0.0046, "within_2pct": True}

        def cpu_baseline(wl, budget_s=14.0):
            modes = [{"mode": "vectorize=True, %d BLAS thread%s" % (c, "s" * (c > 1)), "wu_per_s": 6.1e5 + c, "ms_per_step": 107.0, "steps": 130,
                      "seconds": 14.0, "cores": c} for c in (1, 16)]
            modes.append({"mode": "per-walker log_prob_fn, multiprocessing.Pool(16)", "wu_per_s": 3.7e5, "ms_per_step": 175.0, "steps": 40,
                          "seconds": 7.0, "cores": 16})
            return {"value": 6.17e5, "unit": "walker-updates/s", "cores": 16, "kind": "reference", "sample": "reference emcee itself " * 12,
                    "modes": modes}

        for mod in (bench, _single):
            mod.measure_single = measure_single
            mod.Workload = cheap_workload
        bench.quality_entry = quality_entry
        bench.cpu_baseline = cpu_baseline
