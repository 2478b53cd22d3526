#!/bin/bash
set -u
mkdir -p gpurun_out/r03n
O=gpurun_out/r03n
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_direct_ipc.py tests/test_gpu_logprob_host.py tests/test_c_abi.py -m gpu -q -p no:cacheprovider > $O/pytest_ipc.log 2>&1; tail -15 $O/pytest_ipc.log | cut -c1-400
timeout 900 python bench.py --gpus 2 --all-on-device 0 --steps 10 --warmup 3 > $O/bench_selflaunch_n2_one_device.json 2> $O/bench_n2.err
grep "preflight" $O/bench_n2.err | cut -c1-300
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03n/bench_selflaunch_n2_one_device.json"))
print("value", d.get("value"), "n_gpus", d.get("n_gpus"), "rccl_ranks", d.get("rccl_ranks"), d.get("test_mode"), "err", str(d.get("error"))[:200])
print("preflight", d.get("preflight", {}).get("seconds"), {k: v[:90] for k, v in d.get("preflight", {}).get("disabled", {}).items()})
for k, v in (d.get("multi_gpu") or {}).items():
    print("  ", k, v.get("reported"), {ex: (round(e.get("ms_per_step", -1) * 1e3, 1), e.get("replicas_agree"), e.get("error", "")[:60]) for ex, e in v["exchange"].items()})
PY
