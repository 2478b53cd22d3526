#!/bin/bash
# round 5, session t: the N > 1 bench path after this round's changes: two ranks on the one device of the box (self-launched)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --time-budget 240 --all-on-device 0 ) > $O/bench_n2_one_device.json 2> $O/bench_n2_one_device.err; echo "bench n2 rc=$?" | tee -a $O/summary_t.txt
tail -n 5 $O/bench_n2_one_device.err
python - <<'PY'
import json, os
p = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05/bench_n2_one_device.json"
try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("metric", "value", "n_gpus", "ms_per_step", "scaling")})
    print(str(d.get("census") or d.get("config"))[:400])
except Exception as e:
    print("parse failed", e, open(p).read()[-500:])
PY
