"""bench.py: the ONE stdout line is a compact summary (< 6 KB, numbers and short names); the full record -- every configuration's
roofline audit, pipeline statistics, notes -- goes to bench_detail.json and to stderr.

Round-5 verdict: the line had grown to 23.5 KB and the driver recorded `parsed: null`.  Every emitter (N = 1, N > 1, the error lines,
the self-launcher's pass-through) goes through `emit_record`, and `tests/test_bench_line_cpu.py` holds the size and the keys."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools.benchkit.out import _claim_stdout, log

LINE_LIMIT = 6000           # bytes of the stdout line (the driver parsed 18.4 KB in round 4 and not 23.5 KB in round 5; stay far below)
DETAIL_NAME = "bench_detail.json"

_CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data")
_HEAD_EXTRA = ("steps_per_s", "accept_frac", "device_status", "timed_blocks", "timed_ms", "best_block_ms_per_step", "rccl_ranks",
               "distinct_devices", "error", "launcher", "test_mode", "devices_visible")
_ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "clock", "achieved_event_clock", "frac_event_clock",
             "kernel", "algorithmic_bytes_per_walker_update", "walker_updates_per_launch", "halfsteps_per_launch", "avg_halfstep_us",
             "avg_launch_us", "frac_moved")


def _num(x, digits=6):
    """floats to `digits` significant digits (the detail file keeps every bit)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or abs(x) == float("inf"):
            return None
        if x.is_integer() and abs(x) < 1e9:          # (counts stay exact)
            return x
        return float("%.*g" % (digits, x))
    return x


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def _pick(d, keys, strlen=160):
    out = {}
    for k in keys:
        if k in (d or {}):
            v = d[k]
            out[k] = _short(v, strlen) if isinstance(v, str) else _num(v)
    return out


def _config_summary(e):
    """one configuration -> ms_per_step, wu_per_s, the roofline fraction on value's clock (+ the event clock's) and what bounds it"""
    if not isinstance(e, dict):
        return e
    if "error" in e and "ms_per_step" not in e:
        return {"error": _short(e["error"], 120)}
    rl = e.get("roofline") or {}
    out = _pick(e, ("nwalkers", "ndim", "ms_per_step", "wu_per_s"))
    if e.get("device_status"):
        out["device_status"] = e["device_status"]
    out.update(_pick(rl, ("bound", "frac", "frac_event_clock", "frac_traffic", "algorithmic_bytes_per_walker_update",
                          "algorithmic_flops_per_walker_update", "mfma_f64_frac_wall_clock", "frac_kernel", "frac_of_achievable_6300")))
    return out


def _exact_summary(rec):
    """the same-seed (MT19937) mode: one number per size"""
    out = {}
    e = rec.get("exact_mode")
    if isinstance(e, dict):
        out["c2"] = _pick(e, ("ms_per_step", "wu_per_s", "roofline_frac_wall_clock", "best_block_ms_per_step", "error"), 120)
    e = rec.get("exact_mode_c4")
    if isinstance(e, dict):
        out["c4"] = _pick(e, ("ms_per_step", "wu_per_s", "roofline_frac_wall_clock", "error"), 120)
    e = rec.get("exact_mode_c3")
    if isinstance(e, dict):
        o = {}
        for k in ("device_producer", "host_pipeline"):
            if isinstance(e.get(k), dict):
                o[k + "_ms_per_step"] = _num(e[k].get("ms_per_step"))
        if "error" in e:
            o["error"] = _short(e["error"], 120)
        out["c3"] = o
    e = rec.get("exact_mode_mid")
    if isinstance(e, dict):
        o = {}
        for name, v in e.items():
            if isinstance(v, dict):
                o[name] = {k2: _num(v2.get("ms_per_step")) for k2, v2 in v.items() if isinstance(v2, dict) and "ms_per_step" in v2}
        if "error" in e:
            o["error"] = _short(e["error"], 120)
        out["mid_ms_per_step"] = o
    return out


def _cpu_summary(c):
    if not isinstance(c, dict):
        return c
    out = _pick(c, ("value", "unit", "cores", "kind"))
    out["sample"] = _short(c.get("sample", ""), 260)
    modes = {}
    for m in c.get("modes") or []:
        if isinstance(m, dict) and "wu_per_s" in m:
            modes[_short(m.get("mode", "?"), 60)] = {"wu_per_s": _num(m["wu_per_s"]), "cores": m.get("cores"), "steps": m.get("steps"),
                                                    "seconds": _num(m.get("seconds"), 3)}
    if modes:
        out["modes"] = modes
    return out


def _multi_summary(multi):
    out = {}
    for name, e in (multi or {}).items():
        o = _pick(e, ("nwalkers", "ndim", "scaling", "reported", "ms_per_step", "wu_per_s", "roofline_frac_per_gpu", "mfma_frac_per_gpu"))
        p = e.get("predicted_us_per_step")
        if isinstance(p, dict):
            o["predicted_us_per_step"] = p.get("value")
        ex = {}
        for k, v in (e.get("exchange") or {}).items() if isinstance(e.get("exchange"), dict) else ():
            if "ms_per_step" in v and "error" not in v:
                ex[k] = _num(v["ms_per_step"])
            else:
                ex[k] = "error: " + _short(v.get("error", "?"), 70)
        if isinstance(e.get("exchange"), str):
            o["error"] = _short(e["exchange"], 120)
        o["us_per_step_by_exchange"] = {k: (_num(v * 1e3) if isinstance(v, float) else v) for k, v in ex.items()}
        out[name] = o
    return out


def compact(rec):
    """the full record -> the line the driver parses.  Keeps every contract key, `roofline` and `cpu_baseline` as objects of numbers
    and short names, one small object per further configuration."""
    line = _pick(rec, _CONTRACT, 120)
    for k in _CONTRACT:                       # the contract keys are always there (an error line says null)
        line.setdefault(k, rec.get(k))
    cfg = rec.get("config")
    if isinstance(cfg, dict):
        line["config"] = _pick(cfg, ("workload", "nwalkers", "ndim", "parallelism"), 200)
    line.update(_pick(rec, _HEAD_EXTRA, 300))
    if "timing" in rec:
        line["timing"] = _short(rec["timing"], 120)
    if isinstance(rec.get("roofline"), dict):
        line["roofline"] = _pick(rec["roofline"], _ROOFLINE, 150)
    if "cpu_baseline" in rec:
        line["cpu_baseline"] = _cpu_summary(rec["cpu_baseline"])
    if isinstance(rec.get("configs"), dict):
        line["configs"] = {k: _config_summary(v) for k, v in rec["configs"].items()}
    ex = _exact_summary(rec)
    if ex:
        line["exact_mode"] = ex
    q = rec.get("quality")
    if isinstance(q, dict):
        line["quality"] = _pick(q, ("accept", "tau_mean", "nsteps_over_tau", "accept_rel_diff", "tau_rel_diff", "within_2pct", "error"), 120)
        if isinstance(q.get("reference"), dict):
            line["quality"]["reference"] = _pick(q["reference"], ("accept", "tau_mean"))
    if isinstance(rec.get("multi_gpu"), dict):
        line["multi_gpu"] = _multi_summary(rec["multi_gpu"])
    tb = rec.get("time_budget")
    if isinstance(tb, dict):
        line["time_budget"] = _pick(tb, ("time_budget_s", "used_s", "unbounded_s"))
    pre = rec.get("preflight")
    if isinstance(pre, dict):
        items = pre.get("items") or {}
        line["preflight"] = {"seconds": _num(pre.get("seconds"), 3), "ok": {k: bool(v.get("ok")) for k, v in items.items() if isinstance(v, dict)},
                             "disabled": {k: _short(v, 80) for k, v in (pre.get("disabled") or {}).items()}}
    line["detail"] = rec.get("detail", DETAIL_NAME)
    # the limit holds whatever a future section adds: drop the optional sections, least important first
    for drop in ("preflight", "time_budget", "quality", "exact_mode", "configs", "multi_gpu"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line[drop] = "see " + line["detail"]
    return line


def detail_paths():
    """where the full record goes: $EMX_BENCH_DETAIL, else <repo>/bench_detail.json -- and a copy under gpurun_out/ when that exists
    (the only directory a gpurun call brings back)"""
    p = os.environ.get("EMX_BENCH_DETAIL")
    if p:
        return [p]
    paths = [os.path.join(ROOT, DETAIL_NAME)]
    g = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(g):
        paths.append(os.path.join(g, DETAIL_NAME))
    return paths


def emit_record(rec):
    """full record -> bench_detail.json + stderr; compact line -> stdout (exactly one line)"""
    text = json.dumps(rec)
    for p in detail_paths():
        try:
            with open(p, "w") as f:
                f.write(text + "\n")
        except OSError as e:
            log("bench detail not written to %s: %r" % (p, e))
    print("[bench-detail] " + text, file=sys.stderr, flush=True)
    line = json.dumps(compact(rec))
    out = _claim_stdout()
    out.write(line + "\n")
    out.flush()
    return line
