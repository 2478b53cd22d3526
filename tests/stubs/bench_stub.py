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
