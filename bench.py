#!/usr/bin/env python
"""Contract bench: walker-updates/sec of the fused red/blue half-step path on MI355X.

`python bench.py --gpus N --steps K --warmup W`  (N>1: launched by torch.distributed.run, one rank per GPU).

A "step" is one full ensemble step (every half-step of the move chosen for it) over an ensemble of synthetic
walkers resident in HBM.  The HEADLINE (`metric`, `value`, `ms_per_step`, `roofline`) is BASELINE.json configs[1]:
nwalkers=65536 per GPU, ndim=64, correlated Gaussian with a dense precision matrix, StretchMove a=2.0, counter-based
RNG, float64 -- weak-scaled over the GPUs.  The same JSON line also carries (rank 0, N=1):

  configs     C3 262144x32 Rosenbrock, C4 65536x64 DE+snooker mixture, C5 16384x1024 diagonal Gaussian and C2 with
              the chain stored every step: ms_per_step, wu_per_s, roofline fraction (SURVEY.md 8d bytes formulas)
  exact_mode  C2 under rng=mt19937 (the mode that reproduces reference emcee's chain for a seed)
  quality     acceptance fraction and integrated autocorrelation time of a 1024x64 run 68 tau long, next to the
              reference's numbers for the same configuration (tests/golden/quality_ref.json, build container)
  cpu_baseline reference emcee itself when /root/reference is importable (build container), otherwise the NumPy
              port (oracle/) timed here + the committed reference timings (profiles/r02/cpu_reference.json)

and at N>1 `multi_gpu`: C2 weak, C3 (262144 walkers sharded) and C5 (16384x1024, strong scaling), each under every
exchange protocol (emcee_amd/parallel.py), the fastest valid one reported per config.

Timing: after W warm-up steps, blocks of EXACTLY K steps are timed, each bracketed by barrier + synchronize, until
>= 1 s has been measured; the MEDIAN block is reported (max over ranks per block).  `--single-block` restores the
one-block protocol.

stdout carries exactly one JSON line, and a COMPACT one (< 6 KB: tools/benchkit/emit.py -- the contract keys, `roofline` and
`cpu_baseline` as objects of numbers, one small object per further configuration); the full record with every audit field and
note goes to bench_detail.json (and gpurun_out/bench_detail.json when that directory exists) and to stderr.  `roofline.achieved`
/ `frac` are on the clock of `value` (the wall clock of the timed blocks); the hipEvent figures of the same region are beside
them as `*_event_clock`, `avg_halfstep_us`, `avg_launch_us`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the pieces (tools/benchkit/): re-exported here because tools/ and tests/ address them as bench.<name>
from tools.benchkit.model import *  # noqa: E402,F401,F403
from tools.benchkit.out import _claim_stdout, log  # noqa: E402,F401
from tools.benchkit.cpu import cpu_baseline, usable_cores  # noqa: E402,F401
from tools.benchkit.single import (MAX_BLOCKS, MIN_TIMED_MS, config_entry, exact_mode_c4_entry, exact_mode_entry, exact_mode_large_entry, exact_mode_mid_entry, hbm_traffic,  # noqa: E402,F401
                                   measure_single, quality_entry, refresh_pmc_traffic, wide_entry)
from tools.benchkit import sharded as _sharded  # noqa: E402
from tools.benchkit.sharded import (ALL_EXCHANGES, EXCHANGES, HEAVY_EXCHANGES, _DEADLINE, agreed_remaining, child_main, measure_sharded,  # noqa: E402,F401
                                    preflight_child, run_preflight, sharded_config, sharded_workload, torch_all_ok, worst_case_seconds)
from tools.benchkit.launcher import emit_line, error_line, self_launch  # noqa: E402,F401
from tools.benchkit.emit import LINE_LIMIT, compact, emit_record  # noqa: E402,F401


def run_child(*a, **k):          # (tests replace bench.run_child through EMX_BENCH_STUB: keep the name here and forward)
    return _sharded.run_child(*a, **k)




def main(argv=None):
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", default="all", choices=["all", "c2", "c3", "c4", "c5", "hbm_dense", "hbm_wide", "w512", "w128"],
                    help="which BASELINE configuration(s) to measure beside the C2 headline (N>1: c2, c3, c5)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="N>1: auto = C2 weak (65536 walkers per GPU), C3 / C5 strong (BASELINE's fixed totals)")
    ap.add_argument("--store", action="store_true", help="headline with the chain appended every step (32D+25 bytes)")
    ap.add_argument("--rng", default="philox", choices=["philox", "mt19937"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (no configs / exact_mode / quality)")
    ap.add_argument("--pmc", action="store_true",
                    help="N=1: re-measure roofline.traffic (HBM bytes per launch) with two rocprofv3 --pmc passes of this command "
                         "instead of reading profiles/pmc_traffic.json")
    ap.add_argument("--single-block", action="store_true", help="time one K-step block instead of the median of >= 1 s of blocks")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded RCCL path even at world size 1 (testing)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="sharded runs: collectives enqueued by libemx itself (default) or torch.distributed")
    ap.add_argument("--exchange", default="all", choices=["all"] + list(ALL_EXCHANGES))
    ap.add_argument("--all-on-device", type=int, default=None,
                    help="testing on a one-GPU box: every rank uses this device (RCCL refuses duplicate devices, so only the "
                         "control flow -- failure handling, watchdogs, the emitted line -- is exercised)")
    ap.add_argument("--exchange-timeout", type=float, default=120.0,
                    help="seconds after which a stuck exchange measurement (a child process per rank) is killed")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="--gpus N > 1 started without torchrun: seconds after which the rank processes started here are killed")
    ap.add_argument("--preflight-timeout", type=float, default=45.0,
                    help="N > 1: seconds a preflight child (peer access + every exchange on a tiny ensemble) may take")
    ap.add_argument("--time-budget", type=float, default=840.0,
                    help="N > 1: wall-clock seconds the whole run may take; what does not fit (the later configurations' slower "
                         "protocols) is skipped and named in the line -- the watchdogs alone would allow far more")
    ap.add_argument("--no-preflight", action="store_true")
    ap.add_argument("--preflight", action="store_true", help="N > 1: run the preflight only and print its verdicts")
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)          # internal: "<config>:<exchange>"
    ap.add_argument("--child-port", type=int, default=0, help=argparse.SUPPRESS)
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)

    stub = os.environ.get("EMX_BENCH_STUB")       # tests only: a module whose install(bench) replaces the GPU legs
    if stub:
        import importlib
        importlib.import_module(stub).install(sys.modules[__name__])

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.child:
        # started as plain `python bench.py --gpus N`: this process becomes the launcher of the N ranks
        return self_launch(args, argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.all_on_device is not None:
        local_rank = args.all_on_device
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # the line would claim n_gpus = --gpus while measuring WORLD_SIZE ranks: refuse instead of printing a wrong number
        if rank == 0:
            emit_line(error_line(args, "--gpus %d but WORLD_SIZE=%d: start one rank per GPU (torch.distributed.run "
                                       "--nproc-per-node %d) or run bench.py without a launcher" % (args.gpus, world, args.gpus)))
        return 2
    K, W = args.steps, args.warmup
    if args.child:
        return child_main(args, rank, world, local_rank)

    sharded = world > 1 or args.force_dist
    if not sharded and not stub:
        import torch
        torch.cuda.set_device(local_rank)
    # HBM traffic of the headline kernel: per launch of k_halfstep, per HALF-STEP of k_persist (whose launches differ in length)
    traffic_static = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic_static = json.load(open(tpath))
        except Exception:  # noqa: BLE001
            traffic_static = {}
    traffic_fresh = None          # (bytes, per_halfstep)
    traffic_source = ("profiles/pmc_traffic.json (static: `bench.py --pmc` of the builder's closing session of round 4, profiles/r04/bench_n1_pmc.json -- "
                      "two rocprofv3 --pmc passes of this command, 2 x FETCH_SIZE + WRITE_SIZE per half-step of k_persist; re-measured in round 5 "
                      "by tools/pmc_r05.sh, profiles/r05/pmc_traffic_r05.json, and by `bench.py --pmc` in round 6's closing session, "
                      "profiles/r06/bench_n1_pmc_z2.json: 1 224 B per walker-update = 40.1 MB per half-step each time; not re-measured by "
                      "THIS run; `bench.py --pmc` re-measures)")
    if args.pmc and not sharded:
        fresh, why, per_hs = refresh_pmc_traffic(args)
        if fresh is not None:
            traffic_fresh, traffic_source = (fresh, per_hs), why
        else:
            log("--pmc:", why)
            traffic_source += " [--pmc failed: %s]" % why

    def traffic_per_launch(hpl):
        if traffic_fresh is not None:
            nbytes, per_hs = traffic_fresh
            return nbytes * hpl if per_hs else (nbytes if hpl == 1.0 else None)
        if hpl > 1.0:
            per_hs = traffic_static.get("c2_persist_bytes_per_halfstep")
            return per_hs * hpl if per_hs else None
        return traffic_static.get("c2_stretch_dense_bytes_per_launch")

    def headline(wl, wall_s, gpu_ms, per_launch_us, accept, status, how, extra, hpl=1.0, event_hpl=None, persist_total=None):
        B = wl.bytes_per_update(args.store)
        lps = wl.launches_per_step() / hpl                            # hpl: half-steps per launch (k_persist; 1 otherwise)
        slots_per_launch = (wl.N // world) / lps                      # per GPU
        # ONE clock for `value` and `roofline`: the wall clock of the timed K-step blocks (round-5 verdict: the event clock gave a
        # fraction 2.3 % above what `value` implies).  roofline.frac == value x B / 8e12 / n_gpus by construction; the hipEvent
        # figures of the same region stay beside it under *_event_clock
        avg_launch_s = wall_s / (K * lps)
        avg_launch_ev_s = gpu_ms * 1e-3 / (K * lps)
        achieved = slots_per_launch * B / avg_launch_s / 1e9
        achieved_ev = slots_per_launch * B / avg_launch_ev_s / 1e9
        line = {
            "metric": "walker-updates/sec (whole node), 64-dim correlated Gaussian, StretchMove a=2",
            "value": wl.N * K / wall_s, "unit": "walker-updates/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall_s * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.label + ", rng=%s, store=%s" % (args.rng, args.store),
                       "nwalkers": wl.N, "ndim": wl.D, "parallelism": "walker-sharded x%d%s" % (world, how)},
            "timing": "median of K-step blocks, each bracketed by barrier + synchronize, >= %.0f ms measured" % MIN_TIMED_MS
                      if not args.single_block else "one K-step block",
            "steps_per_s": K / wall_s, "accept_frac": accept, "device_status": status,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic_per_launch(hpl),
                         "clock": "wall (the clock of value); *_event_clock: hipEvents on the kernels' stream over the same region",
                         "achieved_event_clock": achieved_ev, "frac_event_clock": achieved_ev / HBM_PEAK_GBPS,
                         "traffic_source": traffic_source,
                         "kernel": ("emx::k_persist<8,2,4,DPB=4>: %.1f half-steps per launch, device-wide barrier between them" % hpl) if hpl > 1.0 else
                                   "emx::k_halfstep<8,2,4,STRETCH,DPB=4,LEAN>",
                         "kernel_is": "G=8 lanes per walker, V=2, CH=4; f64 MFMA dense target",
                         "halfsteps_per_launch": hpl, "avg_halfstep_us": avg_launch_ev_s * 1e6 / hpl,
                         "avg_halfstep_us_wall": avg_launch_s * 1e6 / hpl,
                         "per_launch_event_halfsteps": event_hpl,
                         "algorithmic_bytes_per_walker_update": B, "walker_updates_per_launch": slots_per_launch,
                         "avg_launch_us": avg_launch_ev_s * 1e6, "avg_launch_us_wall": avg_launch_s * 1e6, "per_launch_event_us": per_launch_us,
                         "note": "avg_launch_us / avg_halfstep_us = hipEvent time of the timed region / launches of the kernel (the figure "
                                 "rocprofv3's kernel statistics must agree with): it includes the inter-kernel gaps and the batched plan kernel "
                                 "(k_native_plan_batch_stretch, 1 launch per 16 steps)"
                                 + (" and, on sharded runs, the exchange" if sharded else "") +
                                 "; per_launch_event_us brackets single launches (of per_launch_event_halfsteps half-steps when "
                                 "persistent) with hipEvents"},
        }
        roofline_audit(line["roofline"], wl, args.store, accept, slots_per_launch / avg_launch_s, line["roofline"]["traffic"], avg_launch_s)
        line.update(extra)
        if persist_total:
            line["persist"] = persist_total
        if args.all_on_device is not None:
            line["test_mode"] = ("--all-on-device %d: every rank shares ONE GPU -- a control-flow / protocol test of the N > 1 path, NOT a "
                                 "multi-GPU measurement" % args.all_on_device)
        return line

    def emit(line):
        emit_record(line)

    if not sharded:
        wl = Workload("c2", 65536)
        res = measure_single(wl, K, W, device=local_rank, rng=args.rng, store=args.store, single_block=args.single_block)
        extra = {"timed_blocks": res["blocks"], "timed_ms": float(np.sum(res["walls_s"]) * 1e3), "best_block_ms_per_step": res["wall_min_s"] * 1e3 / K}
        line = headline(wl, res["wall_s"], res["gpu_ms"], res["per_launch_us"], res["accept_frac"], res["status"], "", extra,
                        hpl=res.get("halfsteps_per_launch", 1.0), event_hpl=res.get("per_launch_halfsteps"),
                        persist_total=res.get("persist_total"))
        if not args.no_extras:
            cfgs = {}
            plan = [("c3", 262144, False), ("c4", 65536, False), ("c5", 16384, False), ("c2", 65536, True),
                    # beyond the BASELINE list: two ensembles no cache can hold (the HBM roofline taken literally) and the
                    # MFMA-bound wide dense targets
                    ("hbm_dense", 1048576, False), ("hbm_wide", 262144, False), ("w512", 65536, False), ("w128", 65536, False)]
            for key, n, st in plan:
                if args.config not in ("all", key):
                    continue
                name = {"c3": "c3_262144x32_rosen", "c4": "c4_de_snooker", "c5": "c5_16384x1024_diag", "c2": "c2_store",
                        "hbm_dense": "hbm_1048576x64_dense", "hbm_wide": "hbm_262144x1024_diag", "w512": "wide_65536x512_dense",
                        "w128": "dense_65536x128_fused"}[key]
                try:
                    w2 = wl if key == "c2" else Workload(key, n)
                    Ks = K if not st else min(K, 200)          # stored chain: 33.5 MB per step
                    if key in ("hbm_dense", "hbm_wide", "w512"):
                        Ks = max(4, min(K, 20))                # 0.2 - 1.5 ms per step: short blocks, the timed second is filled all the same
                    r2 = measure_single(w2, Ks, min(W, Ks), device=local_rank, rng="philox", store=st, single_block=args.single_block,
                                        spin_s=0.05 if key.startswith(("hbm", "w")) else 0.15)
                    cfgs[name] = wide_entry(w2, r2, Ks) if key == "w512" else config_entry(w2, r2, Ks, st)
                    if key == "w128":     # fused: HBM roofline as for C2; the contraction is 5.4 flop per byte here (C2: 2.7)
                        fl = float(w2.D) ** 2 + 3.0 * w2.D
                        cfgs[name]["roofline"]["mfma_f64_frac_wall_clock"] = cfgs[name]["wu_per_s"] * fl / 1e12 / MFMA_F64_PEAK_TFLOPS
                        if r2.get("halfsteps_per_launch", 1.0) > 1.0:
                            cfgs[name]["roofline"]["kernel"] = ("emx::k_persist_slab<DPB=8,STRETCH> (csrc/emx_pslab.hip: %.1f half-steps per launch; the tile's proposals in "
                                                                "registers, a 32-column LDS slab, eight waves a CU; 144 f64 MFMAs per 16-row tile)" % r2["halfsteps_per_launch"])
                        else:
                            cfgs[name]["roofline"]["kernel"] = ("emx::k_halfstep_slab<DPB=8,STRETCH> (csrc/emx_slab.hip: the tile's proposals in registers, a "
                                                                "32-column LDS slab, eight waves a CU; 144 f64 MFMAs per 16-row tile)")
                except Exception as e:  # noqa: BLE001
                    cfgs[name] = {"error": repr(e)}
                    log("config %s failed: %r" % (name, e))
            line["configs"] = cfgs
            if args.config == "all":          # (a single --config: that configuration only, e.g. under rocprofv3)
                try:
                    line["exact_mode"] = exact_mode_entry(wl, K, W, local_rank)
                except Exception as e:  # noqa: BLE001
                    line["exact_mode"] = {"error": repr(e)}
                try:
                    line["exact_mode_c4"] = exact_mode_c4_entry(K, W, local_rank)
                except Exception as e:  # noqa: BLE001
                    line["exact_mode_c4"] = {"error": repr(e)}
                try:
                    line["exact_mode_c3"] = exact_mode_large_entry(K, W, local_rank)
                except Exception as e:  # noqa: BLE001
                    line["exact_mode_c3"] = {"error": repr(e)}
                try:
                    line["exact_mode_mid"] = exact_mode_mid_entry(K, W, local_rank)
                except Exception as e:  # noqa: BLE001
                    line["exact_mode_mid"] = {"error": repr(e)}
                try:
                    line["quality"] = quality_entry(local_rank)
                except Exception as e:  # noqa: BLE001
                    line["quality"] = {"error": repr(e)}
        line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(wl)
        emit(line)
        return

    # N > 1: this process only orchestrates (CPU; a gloo group of the parents for agreement on what failed).  Every
    # (configuration, exchange) measurement is a child process per rank with a rendezvous of its own.
    import torch.distributed as dist
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    port_base = int(os.environ["MASTER_PORT"]) + 17

    keys = ["c2", "c3", "c5", "w512"] if args.config == "all" else [args.config if args.config in ("c2", "c3", "c5", "w512") else "c2"]
    if "c2" not in keys:
        keys = ["c2"] + keys                 # the headline is always C2
    multi = {}
    line = None
    skip = {}
    exchanges = (EXCHANGES + ("logprob",) if "w512" in keys else EXCHANGES) if args.exchange == "all" else (args.exchange,)
    t_start = time.perf_counter()
    _DEADLINE[0] = t_start + args.time_budget if args.time_budget > 0 else None
    budget = worst_case_seconds(args, keys, exchanges)
    log("rank %d: time budget %.0f s (the watchdogs alone would allow %.0f s)" % (rank, args.time_budget, budget["unbounded_s"]))
    pre = None
    if not args.no_preflight or args.preflight:
        t0 = time.perf_counter()
        pre = run_preflight(args, world, dist, port_base + 200, exchanges)
        pre_s = time.perf_counter() - t0
        for ex in exchanges:
            v = pre.get(ex, {})
            if not v.get("ok"):
                skip[ex] = "preflight: " + str(v.get("error") or "final ensembles differ / device status %s" % v.get("device_status"))[:200]
        if not pre.get("p2p", {}).get("ok", True):
            log("preflight: no peer access between the devices -- the direct exchange cannot work")
            skip.setdefault("direct", "preflight: hipDeviceCanAccessPeer is false for some pair of devices")
        pre = {"seconds": pre_s, "items": pre, "disabled": dict(skip), "time_budget": budget}
        log("rank %d preflight (%.1f s): %s" % (rank, pre_s, {k: v.get("ok") for k, v in pre["items"].items()}))
        if args.preflight:
            if rank == 0:
                emit_line({"preflight": pre, "n_gpus": world})
            dist.barrier()
            dist.destroy_process_group()
            return 0
    for kn, key in enumerate(keys):
        wl, best, entry = sharded_config(key, world, K, rank, dist, args, port_base + 8 * kn, skip)
        multi[{"c2": "c2_weak_65536_per_gpu", "c3": "c3_262144x32_rosen_sharded", "c5": "c5_16384x1024_strong",
               "w512": "wide_65536x512_dense_weak"}[key]
              if args.scaling == "auto" else "%s_%s" % (key, entry["scaling"])] = entry
        if key == "c2":
            if best is None:
                if rank == 0:
                    emit(error_line(args, "no exchange protocol produced a valid measurement of the headline configuration",
                                    {"multi_gpu": multi, "preflight": pre}))
                dist.barrier()
                dist.destroy_process_group()
                return
            how = ", %s via %s" % ({"pull": "all-to-all of the partner rows (pull exchange)",
                                    "allgather": "all-gather of the updated rows",
                                    "direct": "partner rows read in place from the peers' HBM (direct exchange)",
                                    "replay": "all-gather of the decisions, accepted updates recomputed on every replica (replay exchange)",
                                    "replay_push": "decisions stored into the peers' buffers, accepted updates recomputed on every replica "
                                                   "(device-side replay exchange)"}.get(
                                        best["exchange"], best["exchange"]), best["comm"])
            line = headline(wl, best["wall_s"], best["gpu_ms"], None, best["accept_frac"], best["status"], how,
                            {"timed_blocks": best["blocks"], "rccl_ranks": best.get("rccl_ranks"),
                             "distinct_devices": best.get("distinct_devices")})
            line["scaling"] = entry["scaling"]
            line["cpu_baseline"] = None
            line["multi_gpu"] = multi
    if rank == 0 and line is not None:
        line["multi_gpu"] = multi
        budget["used_s"] = time.perf_counter() - t_start
        line["time_budget"] = budget
        if pre is not None:
            line["preflight"] = pre
        emit(line)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
