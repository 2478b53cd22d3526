#!/bin/bash
set -u
mkdir -p gpurun_out/r03g
O=gpurun_out/r03g
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log; grep -h "us/step" $O/pytest_gpu.log
timeout 300 python bench.py --force-dist --config c2 --steps 20 --warmup 5 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; tail -2 $O/bench_forcedist.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03g/bench_forcedist.json"))
for k, v in d["multi_gpu"].items():
    print(k, v.get("reported"), {ex: (round(e.get("ms_per_step", -1) * 1e3, 2), e.get("error", "")[:80]) for ex, e in v["exchange"].items()})
print(d.get("preflight", {}).get("seconds"), d.get("preflight", {}).get("disabled"))
PY
du -sh $O
