"""bench.py at N > 1: one measurement of one exchange protocol (a child process per rank), the preflight, the orchestration of
every (configuration, protocol) pair under watchdogs and a wall-clock budget."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools.benchkit.model import *  # noqa: F401,F403
from tools.benchkit.out import _claim_stdout, log
from tools.benchkit.single import MIN_TIMED_MS, MAX_BLOCKS  # noqa: F401

BENCH_PY = os.path.join(ROOT, "bench.py")



# ------------------------------------------------------------------------------------------------ multi-GPU measurement
EXCHANGES = ("allgather", "pull", "direct", "replay", "replay_push")      # measured by default at N > 1
# "logprob" (proposal / commit replicated, log-prob evaluations shared out: for targets that dominate the step) is measured
# on request only: on the closed-form BASELINE targets the replicated part is most of the step
ALL_EXCHANGES = EXCHANGES + ("logprob",)
# the compute-heavy configuration (65 536 x 512 dense per GPU, weak scaling): the protocols that share out the evaluation
HEAVY_EXCHANGES = ("replay", "replay_push", "logprob")


_NCCL_GROUP = {}


def _torch_nccl_group(dist):
    """torch.distributed's own RCCL communicator, next to the gloo bootstrap group (fallback data path)"""
    if "g" not in _NCCL_GROUP:
        _NCCL_GROUP["g"] = dist.new_group(backend="nccl")
    return _NCCL_GROUP["g"]


def measure_sharded(wl, K, W, exchange, rank, world, local_rank, dist, comm_mode, single_block=False, direct_timeout_ms=None):
    """One sharded measurement (fresh context): spin-up, W warm-up steps, K-step blocks."""
    import torch
    from emcee_amd.device import DeviceEnsemble
    ens = DeviceEnsemble(wl.N, wl.D, device=local_rank)
    wl.install(ens, "philox")
    push = exchange == "replay_push"          # the replay exchange with the decisions stored into the peers' buffers (no collective)
    if push:
        exchange = "replay"
    ens.set_exchange(exchange)
    if direct_timeout_ms:
        ens.set_tuning("direct_timeout_ms", int(direct_timeout_ms))
    comm_used = None
    if comm_mode == "torch" and not push:
        if exchange == "direct":
            raise RuntimeError("the direct exchange is driven by libemx itself (--comm rccl)")
        from emcee_amd.parallel import DeviceEngine, PullStepper, ShardedStepper
        ens.set_stream(torch.cuda.current_stream().cuda_stream)   # kernels + RCCL ordered on one stream
        eng = DeviceEngine(ens, rank, world, torch.device("cuda", local_rank), exchange=exchange)
        grp = None if dist.get_backend() == "nccl" else _torch_nccl_group(dist)
        gather = lambda out, inp: dist.all_gather_into_tensor(out, inp, group=grp)  # noqa: E731
        if exchange == "logprob":
            from emcee_amd.parallel import LogProbStepper
            stepper = LogProbStepper(eng, gather)
        elif exchange == "replay":
            from emcee_amd.parallel import ReplayStepper
            stepper = ReplayStepper(eng, gather)
        elif exchange == "pull":
            stepper = PullStepper(eng, lambda out, inp: dist.all_to_all_single(out, inp, group=grp), gather)
        else:
            stepper = ShardedStepper(eng, gather)
        run = lambda k: stepper.run(k, 1, False)  # noqa: E731
        comm_used = "torch.distributed(nccl)"
    elif push:
        # no collective library anywhere on this path: the ranks map each other's receive buffers and barrier flags (hipIpc
        # handles over the gloo bootstrap group) and emx_run exchanges the decisions with plain stores + the device-side barrier
        from emcee_amd.parallel import import_direct_peers
        ens.set_shard(rank, world)
        import_direct_peers(ens, dist)
        run = lambda k: ens.run(k, 1, False)  # noqa: E731
        comm_used = "hipIpc stores + device-side barrier (no collective library)"
    else:
        uid = [DeviceEnsemble.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ens.comm_init(rank, world, uid[0])      # ncclCommInitRank; emx_run now exchanges per half-step
        if exchange == "direct":                # map the peers' coordinate arrays and barrier flags (IPC handles over gloo)
            from emcee_amd.parallel import import_direct_peers
            import_direct_peers(ens, dist)
        run = lambda k: ens.run(k, 1, False)  # noqa: E731
        comm_used = "libemx->RCCL"

    def fence():
        ens.sync()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    dist.barrier()                # the ranks enter the first step together (the device-side barriers are bounded, not patient)
    for _ in range(6):            # a FIXED count: every rank must issue the same collectives
        run(5)
        ens.sync()
    run(W)
    fence()
    walls, gpus = [], []
    total, nblk = 0.0, 0
    digest, every = None, None
    while True:
        fence()
        ens.timer_start()
        t0 = time.perf_counter()
        run(K)
        gpu_ms = ens.timer_stop()
        fence()
        wall = time.perf_counter() - t0
        t = torch.tensor([wall, gpu_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # gloo (bootstrap group) or nccl: both fine for 2 doubles
        walls.append(float(t[0]))
        gpus.append(float(t[1]))
        total += float(t[0]) * 1e3
        nblk += 1
        if digest is None:
            # checksum of the ensemble after a FIXED number of steps (spin-up + W + K): every rank and every exchange
            # protocol must arrive at the same state
            x, lp = ens.get_state()
            digest = "%.17g/%.17g" % (float(np.sum(x * np.arange(1, wl.D + 1))), float(np.sum(lp)))
            every = [None] * world
            dist.all_gather_object(every, digest)
        if single_block or (total >= MIN_TIMED_MS and nblk >= 3) or nblk >= 60:     # same decision on every rank: t is reduced
            break
    res = {"wall_s": float(np.median(walls)), "gpu_ms": float(np.median(gpus)), "blocks": nblk, "comm": comm_used,
           "exchange": "replay_push" if push else exchange, "accept_frac": float(ens.accepted_mask().mean()), "status": ens.status(),
           "digest": digest, "replicas_agree": len(set(every)) == 1}
    res.update(_rank_census(ens, dist, "peers" if push else comm_mode, local_rank))
    if comm_mode != "torch" and not push:
        ens.comm_destroy()
    if push:
        dist.barrier()                          # nobody unmaps while a peer may still store into its buffers
    ens.close()
    return res


def _rank_census(ens, dist, comm_mode, local_rank):
    """How many ranks the communicator that carried the exchange really has (ncclCommCount of libemx's communicator, or the
    torch process group's size) and how many DISTINCT devices the ranks sit on: n_gpus = N is only claimed when both say N."""
    import torch
    try:
        if comm_mode == "peers":               # device-side replay exchange: the ranks whose buffers this rank mapped (itself included)
            ranks = dist.get_world_size()
        else:
            ranks = ens.comm_count() if comm_mode != "torch" else dist.get_world_size()
    except Exception as e:  # noqa: BLE001
        log("comm_count failed:", e)
        ranks = None
    try:
        p = torch.cuda.get_device_properties(local_rank)
        ident = "%s/%s" % (getattr(p, "uuid", None), "%x:%x:%x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0),
                                                                    getattr(p, "pci_device_id", 0)))
    except Exception:  # noqa: BLE001
        ident = "device%d" % local_rank
    every = [None] * dist.get_world_size()
    dist.all_gather_object(every, ident)
    return {"rccl_ranks": ranks, "distinct_devices": len(set(every))}



def preflight_child(args, rank, world, local_rank, dist):
    """`--child preflight:<exchange,...>`: first contact with the node, seconds per item instead of a 120 s watchdog each.
    Checks, in order: peer access between the devices, then every exchange protocol asked for on a tiny ensemble (8 steps,
    ensembles compared across the ranks).  One line per finished item goes out immediately, so a hang is attributed to the item
    in flight."""
    import torch
    out = _claim_stdout()

    def say(item, verdict):
        out.write("EMX_PREFLIGHT %s %s\n" % (item, json.dumps(verdict)))
        out.flush()

    res = {}
    try:
        ndev = torch.cuda.device_count()
        peers = [bool(torch.cuda.can_device_access_peer(local_rank, q)) for q in range(min(ndev, world)) if q != local_rank] \
            if args.all_on_device is None else []
        res["p2p"] = {"ok": all(peers), "devices_visible": ndev, "peer_access": peers}
    except Exception as e:  # noqa: BLE001
        res["p2p"] = {"ok": False, "error": repr(e)}
    say("p2p", res["p2p"])
    wl = TinyWorkload(world)
    for ex in args.child.split(":", 1)[1].split(","):
        if not ex:
            continue
        t0 = time.perf_counter()
        try:
            r = measure_sharded(wl, 8, 2, ex, rank, world, local_rank, dist, args.comm, single_block=True, direct_timeout_ms=2000)
            ok = r["status"] == 0 and r["replicas_agree"]
            res[ex] = {"ok": bool(ok), "seconds": time.perf_counter() - t0, "device_status": r["status"], "replicas_agree": r["replicas_agree"],
                       "digest": r["digest"], "rccl_ranks": r.get("rccl_ranks"), "distinct_devices": r.get("distinct_devices")}
        except Exception as e:  # noqa: BLE001
            res[ex] = {"ok": False, "seconds": time.perf_counter() - t0, "error": repr(e)[:300]}
        allok = torch_all_ok(dist, res[ex]["ok"])
        if not allok and res[ex]["ok"]:
            res[ex] = {"ok": False, "error": "failed on another rank"}
        say(ex, res[ex])
    return res


def run_preflight(args, world, dist, port0, exchanges):
    """-> {item: verdict}.  A child that hangs is killed after --preflight-timeout; what it had finished counts, the item in
    flight is marked failed and the rest is tried again in a fresh child."""
    verdicts = {}
    todo = list(exchanges)
    attempt = 0
    while True:
        r = run_child(args, "preflight", ",".join(todo), port0 + attempt, args.preflight_timeout + (180.0 if attempt == 0 else 0.0),
                      keep_partial=True)
        attempt += 1
        done = r.get("preflight", {}) if isinstance(r, dict) else {}
        for k, v in done.items():
            verdicts.setdefault(k, v)
        left = [e for e in todo if e not in verdicts]
        # every rank must take the same decision: agree on the shortest list of finished items
        n_done = len(todo) - len(left)
        import torch
        t = torch.tensor([n_done])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        n_done = int(t[0])
        for e in todo[n_done:]:
            verdicts.pop(e, None)
        left = todo[n_done:]
        if not left or attempt >= 4:
            for e in left:
                verdicts[e] = {"ok": False, "error": "not reached"}
            break
        verdicts[left[0]] = {"ok": False, "error": r.get("error") or "hung or crashed during the preflight (child killed)"}
        todo = left[1:]
        if not todo:
            break
    return verdicts


def sharded_workload(key, world, args, make_p0=True):
    scaling = {"c2": "weak", "c3": "strong", "c5": "strong", "w512": "weak"}[key] if args.scaling == "auto" else args.scaling
    base = {"c2": 65536, "c3": 262144, "c5": 16384, "w512": 65536}[key]
    return Workload(key, base * world if scaling == "weak" else base, make_p0=make_p0), scaling


def child_main(args, rank, world, local_rank):
    """One (configuration, exchange) measurement in a process of its own: a protocol that crashes the GPU runtime or hangs in a
    collective takes this child with it, not the rank's orchestrating parent (which never touches the GPU at N > 1)."""
    import torch
    import torch.distributed as dist
    key, ex = args.child.split(":", 1)
    torch.cuda.set_device(local_rank)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % args.child_port, rank=rank, world_size=world)
    if key == "preflight":
        out = {"preflight": preflight_child(args, rank, world, local_rank, dist)}
        _claim_stdout().write("EMX_CHILD_RESULT " + json.dumps(out) + "\n")
        _claim_stdout().flush()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        return
    wl, _ = sharded_workload(key, world, args)
    out = {"error": None}
    try:
        out = measure_sharded(wl, args.steps, args.warmup, ex, rank, world, local_rank, dist, args.comm, args.single_block)
    except Exception as e:  # noqa: BLE001
        first = repr(e)
        log("rank %d: exchange '%s' on %s failed: %s" % (rank, ex, key, first))
        out = {"error": first}
    if args.comm == "rccl" and ex not in ("direct", "replay_push"):
        # library-driven RCCL unavailable on some rank: the same protocol over torch.distributed's communicator
        if not torch_all_ok(dist, out.get("error") is None):
            try:
                out = measure_sharded(wl, args.steps, args.warmup, ex, rank, world, local_rank, dist, "torch", args.single_block)
            except Exception as e:  # noqa: BLE001
                out = {"error": (out.get("error") or "failed on another rank") + " | torch.distributed fallback: " + repr(e)}
    sys.stdout.flush()
    _claim_stdout().write("EMX_CHILD_RESULT " + json.dumps(out) + "\n")
    _claim_stdout().flush()
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


def _partial_preflight(text):
    done = {}
    for line in (text or "").splitlines():
        if line.startswith("EMX_PREFLIGHT "):
            try:
                _, item, verdict = line.split(" ", 2)
                done[item] = json.loads(verdict)
            except Exception:  # noqa: BLE001
                pass
    return done


def run_child(args, key, ex, port, timeout_s, keep_partial=False):
    """-> the child's result dict, or {"error": ...} (non-zero exit, no result line, or the timeout)."""
    import subprocess
    cmd = [sys.executable, BENCH_PY, "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--child", "%s:%s" % (key, ex), "--child-port", str(port), "--comm", args.comm, "--scaling", args.scaling]
    if args.single_block:
        cmd.append("--single-block")
    if args.all_on_device is not None:
        cmd += ["--all-on-device", str(args.all_on_device)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=None, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired as e:
        out = {"error": "no result within %.0f s (hung; child killed)" % timeout_s}
        if keep_partial:
            txt = e.stdout.decode() if isinstance(e.stdout, bytes) else e.stdout
            out["preflight"] = _partial_preflight(txt)
        return out
    for line in (r.stdout or "").splitlines():
        if line.startswith("EMX_CHILD_RESULT "):
            try:
                return json.loads(line[len("EMX_CHILD_RESULT "):])
            except Exception as e:  # noqa: BLE001
                return {"error": "unreadable child result: %r" % (e,)}
    out = {"error": "child exited with code %d and no result" % r.returncode}
    if keep_partial:
        out["preflight"] = _partial_preflight(r.stdout)
    return out


_CHILDREN_RUN = []
_DEADLINE = [None]          # N > 1: perf_counter value by which the orchestrator wants to be done (--time-budget)


def agreed_remaining(dist):
    """seconds left of the time budget, the same number on every rank (the minimum over their clocks); None without a budget"""
    if _DEADLINE[0] is None:
        return None
    import torch
    t = torch.tensor([_DEADLINE[0] - time.perf_counter()], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t[0])


def worst_case_seconds(args, keys, exchanges):
    """what the watchdogs alone would allow: every preflight attempt and every (configuration, exchange) child running into its
    timeout -- the number the time budget exists to cut down"""
    pre = 0.0 if args.no_preflight else (args.preflight_timeout + 180.0) + 3 * args.preflight_timeout
    total = 0.0
    first = True
    for key in keys:
        for ex in ((HEAVY_EXCHANGES if key == "w512" else EXCHANGES) if args.exchange == "all" else (args.exchange,)):
            total += args.exchange_timeout + (180.0 if first else 0.0) + (180.0 if key == "w512" else 0.0)
            first = False
    return {"preflight_s": pre, "measurements_s": total, "unbounded_s": pre + total, "time_budget_s": args.time_budget,
            "note": "unbounded = every watchdog firing (a protocol that failed once is not tried again, so at most one timeout per "
                    "protocol in practice); the orchestrator stops starting children once the budget is spent and says what it skipped"}



def sharded_config(key, world, K, rank, dist, args, port0, skip):
    """Every exchange protocol on one workload, each in its own child process per rank; the fastest whose final ensemble
    agrees on all ranks (and with the first protocol's) is reported.  `skip`: protocols that already failed on an earlier
    configuration (not tried again)."""
    wl, scaling = sharded_workload(key, world, args, make_p0=False)       # the orchestrator only needs the description
    results, errors = {}, {}
    exchanges = (HEAVY_EXCHANGES if key == "w512" else EXCHANGES) if args.exchange == "all" else (args.exchange,)
    for n, ex in enumerate(exchanges):
        if ex in skip:
            errors[ex] = "skipped: failed on an earlier configuration (%s)" % skip[ex]
            continue
        # the very first child also pays for cold caches (kernel modules, code objects, a slower first torch import)
        first = not _CHILDREN_RUN
        # (the weak-scaled 512-dimensional ensemble is 2 GB of start state per rank to generate and upload: give it time)
        timeout_s = args.exchange_timeout + (180.0 if first else 0.0) + (180.0 if key == "w512" else 0.0)
        rem = agreed_remaining(dist)
        if rem is not None:
            if rem < 30.0:
                errors[ex] = "skipped: the run's time budget (--time-budget %.0f s) is spent" % args.time_budget
                continue
            timeout_s = min(timeout_s, rem - 10.0)
        _CHILDREN_RUN.append((key, ex))
        r = run_child(args, key, ex, port0 + n, timeout_s)
        ok = r.get("error") is None and "wall_s" in r
        if not torch_all_ok(dist, ok):             # the parents' own gloo group: CPU only
            errors[ex] = r.get("error") or "failed on another rank"
            skip[ex] = "%s: %s" % (key, errors[ex][:200])
            log("rank %d: exchange '%s' on %s: %s" % (rank, ex, key, errors[ex]))
        else:
            results[ex] = r
    ref_digest = None
    best = None
    summary = {}
    for ex in exchanges:
        r = results.get(ex)
        if r is None:
            summary[ex] = {"error": errors.get(ex, "?")}
            continue
        if ref_digest is None:
            ref_digest = r["digest"]
        census_ok = args.all_on_device is not None or (r.get("rccl_ranks") == world and r.get("distinct_devices") == world)
        valid = r["status"] == 0 and r["replicas_agree"] and r["digest"] == ref_digest and census_ok
        wu = wl.N * K / r["wall_s"]
        xb = xgmi_bytes_per_update(wl, ex, world, r.get("accept_frac"))
        B = wl.bytes_per_update(False)
        summary[ex] = {"ms_per_step": r["wall_s"] * 1e3 / K, "wu_per_s": wu, "comm": r["comm"],
                       "device_status": r["status"], "replicas_agree": r["replicas_agree"],
                       "same_final_state_as_first": r["digest"] == ref_digest, "blocks_timed": r["blocks"],
                       "rccl_ranks": r.get("rccl_ranks"), "distinct_devices": r.get("distinct_devices"),
                       "roofline_frac_per_gpu": wu * B / 1e9 / HBM_PEAK_GBPS / world,
                       # bytes every GPU RECEIVES over xGMI per step, and the rate that is against the 7-link ingress cap
                       "xgmi_bytes_per_walker_update": xb, "xgmi_bytes_per_step_per_gpu": xb * wl.N / world,
                       "xgmi_ingress_GBps_per_gpu": xb * wu / world / 1e9,
                       "xgmi_ingress_frac_of_cap": xb * wu / world / 1e9 / XGMI_INGRESS_GBPS}
        if not census_ok:
            summary[ex]["error"] = "rank census failed: %s RCCL ranks on %s distinct devices, expected %d" % (
                r.get("rccl_ranks"), r.get("distinct_devices"), world)
        if valid and (best is None or r["wall_s"] < best["wall_s"]):
            best = r
    entry = {"workload": wl.label, "nwalkers": wl.N, "ndim": wl.D, "scaling": scaling, "exchange": summary}
    pred = PREDICTED_US_PER_STEP.get(key, {}).get(world)
    if pred:
        entry["predicted_us_per_step"] = {"value": pred, "source": "DESIGN.md section 6 (written before any N>1 run)"}
    if best is not None:
        B = wl.bytes_per_update(False)
        wu = wl.N * K / best["wall_s"]
        if key == "w512":      # MFMA-bound: D^2 + 3 D flop per walker-update against the f64 matrix peak
            entry["mfma_frac_per_gpu"] = wu * (float(wl.D) ** 2 + 3.0 * wl.D) / 1e12 / MFMA_F64_PEAK_TFLOPS / world
        entry.update({"reported": best["exchange"], "ms_per_step": best["wall_s"] * 1e3 / K, "wu_per_s": wu,
                      "steps_per_s": K / best["wall_s"], "accept_frac": best["accept_frac"],
                      "rccl_ranks": best.get("rccl_ranks"), "distinct_devices": best.get("distinct_devices"),
                      "roofline_frac_per_gpu": wu * B / 1e9 / HBM_PEAK_GBPS / world})
    return wl, best, entry


def torch_all_ok(dist, ok):
    import torch
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag[0]) == 1
