"""`python bench.py --gpus N` must run N ranks whether or not a launcher is around it (round-2 verdict: without torchrun it
measured ONE GPU and printed n_gpus: 1).  The GPU legs are stubbed (tests/stubs/bench_stub.py); the launcher, the rank
processes, their gloo rendezvous, the preflight bookkeeping and the emitted line are the real ones."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, tmp_path, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"EMX_BENCH_STUB": "tests.stubs.bench_stub", "EMX_BENCH_STUB_DIR": str(tmp_path), "PYTHONPATH": ROOT,
                "EMX_BENCH_DETAIL": os.path.join(str(tmp_path), "bench_detail.json")})
    env.update(extra_env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, lines


def _detail(tmp_path):
    """the full record (the stdout line is its compact summary: tools/benchkit/emit.py, tests/test_bench_line_cpu.py)"""
    return json.load(open(os.path.join(str(tmp_path), "bench_detail.json")))


def _ranks_seen(tmp_path):
    seen = {}
    for f in os.listdir(tmp_path):
        if f.startswith("rank"):
            for ln in open(os.path.join(tmp_path, f)):
                d = json.loads(ln)
                seen.setdefault(d["rank"], []).append(d)
    return seen


def test_self_launch_runs_n_ranks_and_prints_one_line(tmp_path):
    r, lines = _run([sys.executable, "bench.py", "--gpus", "4", "--steps", "5", "--warmup", "1"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout                     # the contract: exactly one JSON line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["rccl_ranks"] == 4 and line["distinct_devices"] == 4
    assert line["steps"] == 5 and line["warmup"] == 1 and line["value"] > 0
    assert "self-launch" in line["launcher"]
    seen = _ranks_seen(tmp_path)
    assert sorted(seen) == [0, 1, 2, 3]                  # four rank processes, each with its own LOCAL_RANK ...
    assert sorted(int(v[0]["local_rank"]) for v in seen.values()) == [0, 1, 2, 3]
    assert all(v[0]["world"] == 4 for v in seen.values())
    # ... that went through the same sequence of measurements (they agree through the gloo group they formed)
    seqs = {tuple((d["key"], d["ex"]) for d in v) for v in seen.values()}
    assert len(seqs) == 1
    assert len(lines[0]) < 8192 and line["multi_gpu"]["c2_weak_65536_per_gpu"]["predicted_us_per_step"] > 0
    line = _detail(tmp_path)
    multi = line["multi_gpu"]
    assert set(multi) == {"c2_weak_65536_per_gpu", "c3_262144x32_rosen_sharded", "c5_16384x1024_strong", "wide_65536x512_dense_weak"}
    assert set(multi["wide_65536x512_dense_weak"]["exchange"]) == {"replay", "replay_push", "logprob"}       # the protocols that share out the evaluation
    assert set(c2_ex := multi["c2_weak_65536_per_gpu"]["exchange"]) == {"allgather", "pull", "direct", "replay", "replay_push"} and c2_ex
    c2 = multi["c2_weak_65536_per_gpu"]
    assert c2["nwalkers"] == 4 * 65536 and c2["scaling"] == "weak" and line["scaling"] == "weak"
    assert c2["reported"] == min(c2["exchange"], key=lambda e: c2["exchange"][e]["ms_per_step"])
    for ex, e in c2["exchange"].items():                 # per protocol: the xGMI accounting and the per-GPU roofline fraction
        assert e["xgmi_bytes_per_walker_update"] > 0 and 0 < e["roofline_frac_per_gpu"]
        assert e["xgmi_ingress_frac_of_cap"] == pytest.approx(e["xgmi_ingress_GBps_per_gpu"] / (7 * 76.8))
    assert c2["predicted_us_per_step"]["value"] > 0
    assert line["preflight"]["items"]["p2p"]["ok"] and not line["preflight"]["disabled"]
    # the time budget: what the watchdogs alone would allow is stated, and the run stayed inside the budget
    tb = line["time_budget"]
    assert tb["time_budget_s"] == 840.0 and tb["unbounded_s"] > tb["time_budget_s"] and 0 < tb["used_s"] < tb["time_budget_s"]
    assert line["preflight"]["time_budget"]["unbounded_s"] == tb["unbounded_s"]


def test_a_spent_time_budget_skips_the_rest_and_says_so(tmp_path):
    """--time-budget: with (almost) nothing left the orchestrator starts no further children; the headline configuration's first
    protocols are measured, what was skipped is named -- on every rank alike (the ranks agree on the remaining time)"""
    r, lines = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1", "--time-budget", "40", "--no-preflight"], tmp_path,
                    extra_env={"EMX_BENCH_STUB_SLEEP": "2.0"})          # 2 s per measurement: the budget's last 30 s are reached after ~5 of 18
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(lines[-1])
    assert line["value"] > 0 and any("time budget" in str(v) for e in line["multi_gpu"].values() for v in e["us_per_step_by_exchange"].values())
    line = _detail(tmp_path)
    skipped = [(k, ex) for k, e in line["multi_gpu"].items() for ex, v in e["exchange"].items() if "time budget" in str(v.get("error", ""))]
    measured = [(k, ex) for k, e in line["multi_gpu"].items() for ex, v in e["exchange"].items() if "ms_per_step" in v]
    assert measured and skipped and line["value"] > 0
    assert line["time_budget"]["used_s"] < 120.0
    seen = _ranks_seen(tmp_path)
    assert len({tuple((d["key"], d["ex"]) for d in v) for v in seen.values()}) == 1


def test_torchrun_path_is_the_same_code(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--config", "c2"]
    r, lines = _run(cmd, tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    js = [ln for ln in lines if ln.startswith("{")]
    assert len(js) == 1
    line = json.loads(js[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and "launcher" not in line
    assert sorted(_ranks_seen(tmp_path)) == [0, 1]


def test_world_size_mismatch_is_refused(tmp_path):
    r, lines = _run([sys.executable, "bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"], tmp_path,
                    {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    line = json.loads(lines[-1])
    assert line["value"] is None and "WORLD_SIZE=2" in line["error"]


def test_a_census_that_disagrees_is_never_reported_as_n_gpus(tmp_path):
    # RCCL's own rank count says 1 (every rank alone in its communicator): no protocol is valid, no value is claimed
    r, lines = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "c2"], tmp_path,
                    {"EMX_BENCH_STUB_RANKS": "1"})
    line = json.loads(lines[-1])
    assert line["value"] is None and "no exchange protocol" in line["error"]
    assert "rank census failed" in json.dumps(_detail(tmp_path)["multi_gpu"])


def test_preflight_disables_failing_and_hanging_protocols(tmp_path):
    r, lines = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "c2"], tmp_path,
                    {"EMX_BENCH_STUB_FAIL": "pull", "EMX_BENCH_STUB_HANG": "direct"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(lines[-1])
    assert set(line["preflight"]["disabled"]) == {"pull", "direct"} and line["preflight"]["ok"]["allgather"]
    line = _detail(tmp_path)
    dis = line["preflight"]["disabled"]
    assert "pull" in dis and "direct" in dis and "allgather" not in dis
    ex = line["multi_gpu"]["c2_weak_65536_per_gpu"]["exchange"]
    assert "skipped" in ex["pull"]["error"] and "skipped" in ex["direct"]["error"]
    assert "ms_per_step" in ex["allgather"] and line["multi_gpu"]["c2_weak_65536_per_gpu"]["reported"] != "pull"
    # the measurement children of the disabled protocols were never started
    started = {(d["key"], d["ex"]) for v in _ranks_seen(tmp_path).values() for d in v}
    assert ("c2", "pull") not in started and ("c2", "direct") not in started
