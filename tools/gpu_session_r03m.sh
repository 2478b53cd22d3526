#!/bin/bash
set -u
mkdir -p gpurun_out/r03m
O=gpurun_out/r03m
export TMPDIR=/tmp
# two REAL rank processes on the one GPU, the protocol that needs no collective library: launcher -> preflight -> child per rank ->
# hipIpc import -> emx_run with stores into the peer's buffers + device-side barrier -> digest comparison -> census -> the line
timeout 600 python bench.py --gpus 2 --all-on-device 0 --config c2 --exchange replay_push --steps 20 --warmup 5 > $O/bench_n2_replay_push_one_device.json 2> $O/bench_n2.err
tail -6 $O/bench_n2.err | cut -c1-300; head -c 4000 $O/bench_n2_replay_push_one_device.json; echo
timeout 600 python bench.py --gpus 2 --all-on-device 0 --config w512 --exchange replay_push --steps 6 --warmup 2 > $O/bench_n2_w512_replay_push_one_device.json 2> $O/bench_n2_w512.err
tail -3 $O/bench_n2_w512.err | cut -c1-300; python - <<'PY'
import json
for f in ("bench_n2_replay_push_one_device.json", "bench_n2_w512_replay_push_one_device.json"):
    try:
        d = json.load(open("gpurun_out/r03m/" + f))
        print(f, "value", d.get("value"), "n_gpus", d.get("n_gpus"), "err", str(d.get("error"))[:200])
        for k, v in (d.get("multi_gpu") or {}).items():
            print("  ", k, v.get("reported"), {ex: (round(e.get("ms_per_step", -1) * 1e3, 1), e.get("replicas_agree"), e.get("device_status"), e.get("rccl_ranks"), e.get("error", "")[:100]) for ex, e in v["exchange"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
