"""bench.py: the one-GPU measurements (headline, the other configurations, exact mode, sampling quality, PMC traffic)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools.benchkit.model import *  # noqa: F401,F403
from tools.benchkit.out import log



# ------------------------------------------------------------------------------------------------ single-GPU measurement
def measure_single(wl, K, W, device=0, rng="philox", store=False, single_block=False, want_kernel=True, spin_s=0.15, tuning=None):
    """W warm-up steps, then K-step blocks (each: sync, hipEvent + wall clock around emx_run(K), sync) until >= MIN_TIMED_MS (1 s);
    median block.  Returns per-step times, per-launch event duration of the half-step kernel, accept fraction."""
    from emcee_amd.device import DeviceEnsemble
    ens = DeviceEnsemble(wl.N, wl.D, device=device)
    wl.install(ens, rng)
    for key, val in (tuning or {}).items():
        ens.set_tuning(key, val)
    if store:
        ens.chain_config(max(K, W))
    # untimed spin-up: the first ~50 ms on a fresh context run slower (clock ramp, first touch of the plan ring, lazy
    # code-object loading); tools/stall_probe.py
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < spin_s:
        ens.run(min(50, max(1, K)), 1, False)
        ens.sync()
    if store:
        ens.chain_reset()
    ens.run(W, 1, store)
    ens.sync()
    walls, gpus = [], []
    total = 0.0
    pinfo0 = ens.persist_info()
    while True:
        if store:
            ens.chain_reset()
        ens.sync()
        ens.timer_start()
        t0 = time.perf_counter()
        ens.run(K, 1, store)
        gpu_ms = ens.timer_stop()          # hipEvents on the stream the kernels are launched on; synchronises
        ens.sync()
        wall = time.perf_counter() - t0
        walls.append(wall)
        gpus.append(gpu_ms)
        total += wall * 1e3
        if single_block or (total >= MIN_TIMED_MS and len(walls) >= 3) or len(walls) >= MAX_BLOCKS:
            break
    wall = float(np.median(walls))
    gpu_ms = float(np.median(gpus))
    res = {"wall_s": wall, "gpu_ms": gpu_ms, "blocks": len(walls), "wall_min_s": float(np.min(walls)),
           "accept_frac": float(ens.accepted_mask().mean()), "status": ens.status(), "per_launch_us": None,
           "walls_s": [float(w) for w in walls], "halfsteps_per_launch": 1.0}
    pinfo1 = ens.persist_info()
    persistent = pinfo1["launches"] > pinfo0["launches"]
    if persistent:      # k_persist: several half-steps per launch (16 steps when a call is aligned with the plan batches)
        res["halfsteps_per_launch"] = (pinfo1["halfsteps"] - pinfo0["halfsteps"]) / float(pinfo1["launches"] - pinfo0["launches"])
    if rng == "mt19937":
        try:
            res["pipeline"] = ens.pipeline_stats()
        except Exception as e:  # noqa: BLE001
            log("pipeline stats unavailable:", e)
        try:
            res["mtdev"] = dict(ens.mtdev_info(), tokenizer=ens.mtdev_tok_stats())
        except Exception as e:  # noqa: BLE001
            log("device producer stats unavailable:", e)
    if want_kernel:
        # per-launch hipEvent durations of the half-step kernel (separate pass: event records perturb)
        if persistent and rng == "philox":
            seed, step = ens.get_philox()
            ens.set_philox(seed, step)          # forget the plans evaluated ahead: the launches below are whole 16-step batches
        pinfo2 = ens.persist_info()
        ens.profile_enable(128)
        ens.run(48, 1, False)
        pl = ens.profile_read(128)
        if len(pl):
            res["per_launch_us"] = float(np.median(pl) * 1e3)
            pinfo3 = ens.persist_info()
            if pinfo3["launches"] > pinfo2["launches"]:
                res["per_launch_halfsteps"] = (pinfo3["halfsteps"] - pinfo2["halfsteps"]) / float(pinfo3["launches"] - pinfo2["launches"])
    res["persist_total"] = ens.persist_info()
    ens.close()
    return res


def wide_entry(wl, res, K):
    """Dense Gaussian beyond ndim 112 (emx_wide.hip): propose -> k_wide_lp -> commit.  The MFMA log-prob kernel dominates and is
    bound by the f64 matrix pipe, not by HBM: D^2 + 3 D flop per walker-update in the Cholesky form, against
    24 D + 17 bytes."""
    D, N = wl.D, wl.N
    flops = float(D) * D + 3.0 * D
    ms = res["wall_s"] * 1e3 / K
    wu = N * K / res["wall_s"]
    out = {"workload": wl.label, "nwalkers": N, "ndim": D, "ms_per_step": ms, "wu_per_s": wu, "steps_per_s": K / res["wall_s"],
           "blocks_timed": res["blocks"], "accept_frac": res["accept_frac"], "device_status": res["status"]}
    rl = {"bound": "mfma_f64", "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "algorithmic_flops_per_walker_update": flops,
          "walker_updates_per_launch": N / 2.0, "kernel": "emx::k_wide_lp* (Y = R L by v_mfma_f64_16x16x4_f64, L streamed through LDS)",
          "achieved": wu * flops / 1e12, "frac": wu * flops / 1e12 / MFMA_F64_PEAK_TFLOPS,
          "frac_wall_clock": wu * flops / 1e12 / MFMA_F64_PEAK_TFLOPS,
          "hbm_frac_wall_clock": wu * wl.bytes_per_update(False) / 1e9 / HBM_PEAK_GBPS}
    if res["per_launch_us"]:
        rl["avg_launch_us"] = res["per_launch_us"]
        rl["achieved_kernel"] = (N / 2.0) * flops / (res["per_launch_us"] * 1e-6) / 1e12
        rl["frac_kernel"] = rl["achieved_kernel"] / MFMA_F64_PEAK_TFLOPS
        rl["note"] = "frac prices the WHOLE step (propose + log-prob + commit passes) on the wall clock against the matrix peak; " \
                     "frac_kernel = the log-prob kernel alone, avg_launch_us = hipEvents around single k_wide_lp launches (median of 128)"
    out["roofline"] = rl
    return out


def config_entry(wl, res, K, store):
    B = wl.bytes_per_update(store)
    lps = wl.launches_per_step()
    ms = res["wall_s"] * 1e3 / K
    wu = wl.N * K / res["wall_s"]
    ev_ms = res["gpu_ms"] / K
    out = {"workload": wl.label + (", chain stored every step" if store else ""), "nwalkers": wl.N, "ndim": wl.D,
           "ms_per_step": ms, "wu_per_s": wu, "steps_per_s": K / res["wall_s"], "blocks_timed": res["blocks"],
           "accept_frac": res["accept_frac"], "device_status": res["status"],
           # `frac` on the wall clock of the timed blocks -- the clock of ms_per_step / wu_per_s (round-5 verdict); the hipEvent
           # figures of the same region under *_event_clock
           "roofline": {"bound": "hbm", "algorithmic_bytes_per_walker_update": B,
                        "achieved": wu * B / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": wu * B / 1e9 / HBM_PEAK_GBPS,
                        "achieved_event_clock": wl.N * B / (ev_ms * 1e-3) / 1e9,
                        "frac_event_clock": wl.N * B / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "frac_wall_clock": wu * B / 1e9 / HBM_PEAK_GBPS,
                        "avg_launch_us": ev_ms * 1e3 / lps, "per_launch_event_us": res["per_launch_us"],
                        "launches_per_step": lps}}
    hpl = res.get("halfsteps_per_launch", 1.0)
    if hpl > 1.0 and len(wl.moves) == 1:
        # every step through the persistent kernel: a launch is hpl half-steps
        out["roofline"].update({"kernel": "emx::k_persist<8,2,4,DPB=4>: %.1f half-steps per launch" % hpl,
                                "avg_launch_us": ev_ms * 1e3 / lps * hpl, "launches_per_step": lps / hpl, "avg_halfstep_us": ev_ms * 1e3 / lps,
                                "per_launch_event_halfsteps": res.get("per_launch_halfsteps")})
    elif hpl > 1.0:
        # a mixture: the consecutive steps of one move share a persistent launch (k_persist<..., MOVE_DE / MOVE_SNOOKER>: two half-steps
        # per DE step, four per snooker step); avg_launch_us stays the timed region / half-steps
        out["roofline"]["kernel"] = ("emx::k_persist<8,2,4,DPB=4,MOVE_DE> and <...,MOVE_SNOOKER>: %.1f half-steps per launch (a run of consecutive "
                                     "steps of one move)" % hpl)
        out["roofline"]["avg_launch_us_is"] = "hipEvent time of the timed region / half-steps (persistent launches counted by their half-steps)"
    state_mb = wl.N * wl.D * 8 / 1e6
    traffic = None
    if state_mb <= 256.0:
        # the cache-resident configurations: HBM bytes per walker-update from the round-5 counter passes (tools/pmc_r05.sh, static:
        # the driver does not run rocprofv3), per launch = x walker-updates of a launch
        per_wu = pmc_r05_traffic(wl, store)
        if per_wu:
            traffic = per_wu * wl.N / lps
            out["roofline"].update({"traffic": traffic, "traffic_bytes_per_walker_update": per_wu,
                                    "traffic_source": "profiles/r05/pmc_traffic_r05.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_probe.py "
                                                      "in the builder's session of round 5, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 over all half-step kernels / walker-updates; "
                                                      "not re-measured by this run)"})
    if state_mb > 256.0:
        traffic = hbm_traffic(wl.key)
        out["roofline"].update({"state_MB": state_mb, "beyond_infinity_cache": True,
                                "frac_of_achievable_6300": wu * B / 1e9 / HBM_ACHIEVABLE_GBPS,
                                "traffic": traffic, "traffic_source": "profiles/pmc_traffic.json (static; rocprofv3 PMC passes)"})
    roofline_audit(out["roofline"], wl, store, res["accept_frac"], wu, traffic, ms * 1e-3 / lps)
    if state_mb > 256.0:
        out["roofline"]["frac_moved_of_achievable_6300"] = out["roofline"]["achieved_moved"] / HBM_ACHIEVABLE_GBPS
    return out


def pmc_r05_traffic(wl, store):
    """HBM bytes per walker-update of a bench configuration at its BASELINE size (profiles/r05/pmc_traffic_r05.json), or None."""
    key = {"c2": "c2_store" if store else "c2", "c3": "c3", "c4": "c4", "c5": "c5", "w128": "w128"}.get(wl.key)
    sizes = {"c2": 65536, "c3": 262144, "c4": 65536, "c5": 16384, "w128": 65536}
    if key is None or wl.N != sizes.get(wl.key) or (store and wl.key != "c2"):
        return None
    try:
        cfg = json.load(open(os.path.join(ROOT, "profiles", "r05", "pmc_traffic_r05.json")))["configs"][key]
        return float(cfg["bytes_per_walker_update"])
    except Exception:  # noqa: BLE001
        return None


def hbm_traffic(key):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key + "_bytes_per_launch")
    except Exception:  # noqa: BLE001
        return None


def exact_mode_entry(wl, K, W, device):
    """C2 under rng=mt19937 (same seed => reference emcee's chain).  The host produces every draw of the step from the
    serial NumPy-legacy stream; host_plan_ms times that producer alone (no GPU involved)."""
    from emcee_amd import _lib
    Kx = max(100, min(K, 400))          # one emx_run per block: long enough that the pipeline's thread start-up (0.2 ms) is amortised
    res = measure_single(wl, Kx, max(W, 10), device=device, rng="mt19937", spin_s=0.05)
    lib = _lib.load()
    host_ms = None
    try:
        st = np.random.RandomState(5).get_state()
        key = np.ascontiguousarray(st[1], dtype=np.uint32)
        m = lib.emx_mt_create(key, int(st[2]), int(st[3]), float(st[4]))
        N = wl.N
        off = np.zeros(3, dtype=np.int32)
        order, p0, p1, p2 = (np.empty(N, dtype=np.int32) for _ in range(4))
        s0, ua = np.empty(N), np.empty(N)
        mv = wl.moves[0][1]
        import ctypes as C
        lib.emx_host_plan_mt(m, N, wl.D, C.byref(mv), off, order, p0, p1, p2, s0, ua)
        t0 = time.perf_counter()
        for _ in range(20):
            lib.emx_host_plan_mt(m, N, wl.D, C.byref(mv), off, order, p0, p1, p2, s0, ua)
        host_ms = (time.perf_counter() - t0) * 1e3 / 20
        lib.emx_mt_destroy(m)
    except Exception as e:  # noqa: BLE001
        log("host plan timing failed:", e)
    B = wl.bytes_per_update(False)
    wu = wl.N * Kx / res["wall_s"]
    return {"workload": wl.label + ", rng=mt19937 (NumPy legacy stream, chain identical to reference emcee's)",
            "steps": Kx, "blocks_timed": res["blocks"], "ms_per_step": res["wall_s"] * 1e3 / Kx, "wu_per_s": wu,
            "best_block_ms_per_step": res["wall_min_s"] * 1e3 / Kx,
            "host_plan_ms": host_ms, "kernel_us": res["per_launch_us"], "accept_frac": res["accept_frac"],
            "roofline_frac_wall_clock": wu * B / 1e9 / HBM_PEAK_GBPS,
            "block_spread": float(np.max(res["walls_s"]) / np.min(res["walls_s"])),
            "pipeline_stage_us_per_step": res.get("pipeline"),
            "persistent_halfsteps_per_launch": res.get("halfsteps_per_launch"),
            "note": "round 6: the step's 5 N fixed-length draws are made again ON THE DEVICE from the generator's state (k_plan_regen; the host "
                    "pipeline hands over `order` and the state at every eighth block of their region of the stream) and the persistent kernel "
                    "takes the plans sixteen steps a launch (profiles/r06/exact_regen.md: 45-52 -> 34 us/step).  "
                    "host_plan_ms = one step's plan made inline by ONE host thread (emx_host_plan_mt, no GPU) -- round 1's path; emx_run "
                    "takes its plans from the host pipeline (csrc/emx_mtpipe.cpp: generator thread twisting MT19937 STATE words into a cache-"
                    "resident ring, tokenizer thread deciding the rejections and copying the fixed-length draws out, finisher threads -- six where "
                    "the L3 domain has room, five with device finish: one thread per core of an eight-core domain -- for the swaps and `order`); round 5: a stretch step's uniforms go up as those generator words, "
                    "in the plan's columns, one copy per step on an upload stream of its own, and k_plan_raw tempers / converts / resolves "
                    "partners / takes the logs on the consumer's stream (profiles/r05/exact_c2.md: 57.8 -> 45.8 us/step and the variants "
                    "dropped).  pipeline_stage_us_per_step says which stage bounds it on THIS host (the stages run concurrently: the largest "
                    "of generator / tokenizer / finishers-summed over the thread count is the pipeline's floor; box to box 45.8 ... 54)"}


def exact_mode_c4_entry(K, W, device):
    """BASELINE configs[3] (65 536 x 64, DEMove 0.8 + DESnookerMove 0.2) under rng=mt19937: the reference's own draws for these moves --
    pair codes and polar normals (de.py:46-56), three randint and a shuffle per walker (de_snooker.py:36-39) -- are finished by the
    host pipeline's threads; round-5 verdict asked for the number in the line."""
    wl = Workload("c4", 65536)
    Kx = max(50, min(K, 200))
    res = measure_single(wl, Kx, max(W, 10), device=device, rng="mt19937", spin_s=0.05, want_kernel=False)
    B = wl.bytes_per_update(False)
    wu = wl.N * Kx / res["wall_s"]
    return {"workload": wl.label + ", rng=mt19937", "steps": Kx, "blocks_timed": res["blocks"], "ms_per_step": res["wall_s"] * 1e3 / Kx, "wu_per_s": wu,
            "roofline_frac_wall_clock": wu * B / 1e9 / HBM_PEAK_GBPS, "accept_frac": res["accept_frac"], "device_status": res["status"],
            "pipeline_stage_us_per_step": res.get("pipeline")}


def exact_mode_large_entry(K, W, device):
    """rng=mt19937 at C3's size (262 144 x 32 Rosenbrock): the ensemble size from which the plans of the reference's own stream are
    made ON THE DEVICE (csrc/emx_mtdev.hpp: jump-ahead MT19937 segments, tokenizer and finisher kernels; no host thread touches a
    draw), next to the host pipeline on the same box (tuning mt_device = 0)."""
    wl = Workload("c3", 262144)
    Kx = max(50, min(K, 200))
    out = {"workload": wl.label + ", rng=mt19937 (NumPy legacy stream, chain identical to reference emcee's)", "steps": Kx}
    B = wl.bytes_per_update(False)
    # (round 6: at this size the host pipeline -- its stretch steps go up as generator states -- is the default again, the device producer
    # starts at 786 432 walkers; mt_device = 2 forces it here)
    for name, tune in (("device_producer", {"mt_device": 2}), ("host_pipeline", {"mt_device": 0})):
        res = measure_single(wl, Kx, max(W, 10), device=device, rng="mt19937", spin_s=0.05, want_kernel=False, tuning=tune)
        wu = wl.N * Kx / res["wall_s"]
        e = {"ms_per_step": res["wall_s"] * 1e3 / Kx, "wu_per_s": wu, "blocks_timed": res["blocks"], "device_status": res["status"],
             "roofline_frac_wall_clock": wu * B / 1e9 / HBM_PEAK_GBPS, "accept_frac": res["accept_frac"]}
        md = res.get("mtdev") or {}
        if name == "device_producer":
            e["host_threads"] = 0
            e["producer_used"] = bool(md.get("steps", 0) > 0)
            tk = md.get("tokenizer") or {}
            if tk.get("windows"):
                steps = max(1, md.get("steps", 1))
                e["tokenizer_us_per_step"] = tk["kernel_us"] / steps
                e["tokenizer_rounds_per_window"] = tk["rounds"] / float(tk["windows"])
        else:
            e["pipeline_stage_us_per_step"] = res.get("pipeline")
        out[name] = e
    if "ms_per_step" in out.get("device_producer", {}) and "ms_per_step" in out.get("host_pipeline", {}):
        out["speedup_device_over_host"] = out["host_pipeline"]["ms_per_step"] / out["device_producer"]["ms_per_step"]
    out["note"] = ("the tokenizer (the masked rejection of random.shuffle, red_blue.py:80: the one serial part of a step) bounds the device "
                   "producer; the host pipeline with regen steps is the faster one up to half a million walkers (profiles/r06/mtdev_sizes_r06.txt)")
    return out


def exact_mode_mid_entry(K, W, device):
    """rng=mt19937 at the sizes of the reference's own suite (1 024 and 4 096 walkers x 64, dense Gaussian, StretchMove): the host
    pipeline's plans fetched sixteen steps at a time by ONE kernel and run by the one-XCD persistent kernel (tuning persist_exact,
    round 4) next to the launch-per-half-step path with its upload per step (persist_exact = 0), and the Philox rate of the same shape."""
    out = {"what": "C2's target and move at mid-size ensembles, rng=mt19937 (chain identical to reference emcee's): ms_per_step of emx_run"}
    Kx = max(200, min(K, 400))
    for N in (1024, 4096):
        wl = Workload("c2", N)
        e = {}
        for name, rng, tune in (("persistent_one_xcd", "mt19937", {"persist_exact": 1}), ("per_half_step", "mt19937", {"persist_exact": 0}),
                                ("philox_same_shape", "philox", {})):
            res = measure_single(wl, Kx, max(W, 10), device=device, rng=rng, spin_s=0.05, want_kernel=False, tuning=tune)
            e[name] = {"ms_per_step": res["wall_s"] * 1e3 / Kx, "blocks_timed": res["blocks"], "device_status": res["status"],
                       "accept_frac": res["accept_frac"]}
            if rng == "mt19937":
                e[name]["pipeline_stage_us_per_step"] = res.get("pipeline")
        e["speedup"] = e["per_half_step"]["ms_per_step"] / e["persistent_one_xcd"]["ms_per_step"]
        out["%dx64" % N] = e
    # round 5: a move MIXTURE in exact mode (the reference's recommended usage, docs/tutorials/moves.ipynb) on the persistent kernels --
    # the next step's move is read off the pipeline's plan before it is taken; DE + snooker share launches (k_persist_mix)
    for N in (1024, 4096):
        wl = Workload("c4", N)
        e = {}
        for name, rng, tune in (("persistent", "mt19937", {"persist_exact_mix": 1}), ("upload_per_step", "mt19937", {"persist_exact_mix": 0}),
                                ("philox_same_shape", "philox", {})):
            res = measure_single(wl, Kx, max(W, 10), device=device, rng=rng, spin_s=0.05, want_kernel=False, tuning=tune)
            e[name] = {"ms_per_step": res["wall_s"] * 1e3 / Kx, "blocks_timed": res["blocks"], "device_status": res["status"],
                       "accept_frac": res["accept_frac"]}
            if rng == "mt19937":
                e[name]["pipeline_stage_us_per_step"] = res.get("pipeline")
        e["speedup"] = e["upload_per_step"]["ms_per_step"] / e["persistent"]["ms_per_step"]
        out["mix_de0.8_snooker0.2_%dx64" % N] = e
    out["note"] = ("profiles/r04/exact_mid.txt: what stood in the way (per-step uploads, sleeping stage threads, a shared hardware queue, "
                   "lazily resolved events); mixtures: profiles/r05/exact_mix_probe.txt -- the tokenizer makes the DE move's pair codes and "
                   "polar candidates and the snooker move's draws 16 stream words at a time (round 5: 4 096 x 64 DE + snooker 29.8 -> 16.5 "
                   "us/step); what is left there is the consumer (compare philox_same_shape)")
    return out


def quality_entry(device, rng="philox"):
    """Acceptance fraction and integrated autocorrelation time (reference estimator, c=5) of the 64-dim correlated
    Gaussian, StretchMove a=2, in the configuration the reference itself was run in (tests/golden/quality_ref.json, made by
    `tools/quality.py --ref` in the build container: 1024 walkers, 2000 burn-in + 100 000 steps = 68 tau, thin_by 25), next to
    the reference's numbers.  Different random streams: the comparison is statistical (2 % bar, BASELINE.json)."""
    import emcee_amd
    D = 64
    ref = None
    path = os.path.join(ROOT, "tests", "golden", "quality_ref.json")
    try:
        ref = json.load(open(path))
        cfg = ref["config"]
        nwalkers, nsteps, thin_by, burn = cfg["nwalkers"], cfg["nsteps"], cfg["thin_by"], cfg["burn"]
    except Exception:  # noqa: BLE001
        nwalkers, nsteps, thin_by, burn = 1024, 4000, 25, 2000
    mu, cov, icov = dense_gaussian(D)
    p0 = mu + np.random.RandomState(1).randn(nwalkers, D) @ np.linalg.cholesky(cov).T
    s = emcee_amd.EnsembleSampler(nwalkers, D, emcee_amd.targets.DenseGaussian(mu, icov), rng=rng, device=device)
    s._random.seed(12)
    t0 = time.perf_counter()
    st = s.run_mcmc(p0, burn, skip_initial_state_check=True, store=False)
    s.run_mcmc(st, nsteps, thin_by=thin_by, skip_initial_state_check=True)
    t_run = time.perf_counter() - t0
    t0 = time.perf_counter()
    tau = np.asarray(s.get_autocorr_time(quiet=True)) * thin_by          # the backend counts in stored samples
    t_tau = time.perf_counter() - t0
    acc = float(np.mean(s.acceptance_fraction))
    out = {"workload": "%d walkers x 64-dim correlated Gaussian, StretchMove a=2, %d burn-in + %d steps, thin_by=%d, rng=%s"
                       % (nwalkers, burn, nsteps * thin_by, thin_by, rng),
           "accept": acc, "tau_mean": float(np.mean(tau)), "tau_min": float(np.min(tau)), "tau_max": float(np.max(tau)),
           "nsteps_over_tau": float(nsteps * thin_by / np.mean(tau)), "run_seconds": t_run, "tau_seconds": t_tau}
    if ref is not None:
        r = ref["results"][0]
        out["reference"] = {"source": "tests/golden/quality_ref.json (reference emcee in the build container, tools/quality.py --ref; static)",
                            "accept": r["accept_mean"], "tau_mean": r["tau_mean"], "nsteps_over_tau": r["chain_over_tau"],
                            "seconds": r["seconds"]}
        out["accept_rel_diff"] = acc / r["accept_mean"] - 1.0
        out["tau_rel_diff"] = float(np.mean(tau)) / r["tau_mean"] - 1.0
        out["within_2pct"] = bool(abs(out["accept_rel_diff"]) < 0.02 and abs(out["tau_rel_diff"]) < 0.02)
    return out




# ------------------------------------------------------------------------------------------------ --pmc
def refresh_pmc_traffic(args):
    """`--pmc`: HBM traffic of the headline kernel measured NOW instead of read from profiles/pmc_traffic.json -- this command is
    re-run under rocprofv3 with FETCH_SIZE and with WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md, HBM section: the two
    counters are not collected together; FETCH_SIZE is doubled on gfx950), medians per launch of the stretch / dense half-step.
    -> (bytes per launch -- per half-step when the kernel is the persistent one --, source text, per_halfstep) or (None, reason, False)."""
    import collections
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if not prof:
        return None, "rocprofv3 not on PATH", False
    med = {}
    per_halfstep = False
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="emx_pmc_")
        cmd = [prof, "--pmc", ctr, "-d", d, "-o", "p", "-f", "csv", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10",
               "--warmup", "2", "--no-cpu-baseline", "--no-extras"]
        env = dict(os.environ)
        env.setdefault("TMPDIR", "/tmp")
        # the child's FULL record (its stdout line is the compact one, without the launch counts): a file of its own
        env["EMX_BENCH_DETAIL"] = os.path.join(d, "child_detail.json")
        try:
            cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, env=env, cwd=ROOT)
            child = None
            if os.path.exists(env["EMX_BENCH_DETAIL"]):
                child = json.load(open(env["EMX_BENCH_DETAIL"]))
            else:
                for ln in cp.stdout.decode(errors="replace").splitlines():
                    if ln.startswith("{") and '"metric"' in ln:
                        child = json.loads(ln)
            ptot = (child or {}).get("persist") or {}
            agg = collections.defaultdict(list)
            persist_sum = 0.0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != ctr:
                        continue
                    if "k_halfstep<8, 2, 4, 0, 4, 1>" in r["Kernel_Name"]:
                        agg[r.get("Grid_Size", "")].append(float(r["Counter_Value"]))
                    elif "k_persist<" in r["Kernel_Name"]:
                        persist_sum += float(r["Counter_Value"])
            if ptot.get("halfsteps", 0) > 0 and persist_sum > 0.0:
                # the persistent kernel: launches run different numbers of half-steps, so the sum over every launch of the
                # process / the half-steps they ran (the child's own count), per HALF-STEP
                med[ctr] = persist_sum / float(ptot["halfsteps"])
                per_halfstep = True
                continue
            vals = max(agg.values(), key=len) if agg else []
            if len(vals) < 8:
                return None, "rocprofv3 --pmc %s produced no samples of the half-step kernel" % ctr, False
            med[ctr] = statistics.median(vals)
        except Exception as e:  # noqa: BLE001
            return None, "rocprofv3 --pmc %s failed: %r" % (ctr, e), False
        finally:
            shutil.rmtree(d, ignore_errors=True)
    nbytes = (2.0 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024.0
    return nbytes, ("measured by this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of this command, %s: "
                    "%.1f KB / %.1f KB; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, FETCH_SIZE doubled per the gfx950 correction)"
                    % ("sum over the k_persist launches / the half-steps they ran, i.e. per HALF-STEP" if per_halfstep else "medians per launch",
                       med["FETCH_SIZE"], med["WRITE_SIZE"])), per_halfstep
