#!/bin/bash
# round 5, session j: the whole GPU suite and the driver's bench command on the current code
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -q -x -m gpu -p no:cacheprovider ) > $O/gpu_tests_j.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary_j.txt
tail -n 8 $O/gpu_tests_j.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_n1_j.json 2> $O/bench_n1_j.err; echo "bench rc=$?" | tee -a $O/summary_j.txt
tail -n 3 $O/bench_n1_j.err
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05"
try:
    d = json.loads(open(O + "/bench_n1_j.json").read().strip().splitlines()[-1])
    print("value %.4g %s  ms/step %.5f  frac %.3f" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"]))
    for k in ("exact_mode",):
        e = d.get(k) or d.get("extras", {}).get(k)
        if e: print(k, e.get("ms_per_step"), e.get("pipeline_stage_us_per_step"))
except Exception as e:
    print("parse failed", e)
PY
