#!/bin/bash
# bench.py after the split into tools/benchkit: the default line, the sharded path at world 1, two real ranks on the one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04q
O=$PWD/gpurun_out/r04q
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/bench_n1.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_n1.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"])
for k,v in d["configs"].items(): print(k, {kk:v.get(kk) for kk in ("ms_per_step","error")}, (v.get("roofline") or {}).get("frac"))
print("exact", d["exact_mode"].get("ms_per_step"), "exact_c3", json.dumps(d.get("exact_mode_c3"))[:900])
print("cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"])
PY
timeout 400 python bench.py --force-dist --config c2 --steps 200 --warmup 20 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "forcedist rc=$?" | tee -a $O/summary.txt
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_forcedist.json") if l.startswith("{")][-1])
print(json.dumps(d)[:600])
PY
timeout 500 python bench.py --gpus 2 --all-on-device 0 --exchange replay_push --steps 20 --warmup 5 > $O/bench_n2_one_device.json 2> $O/bench_n2_one_device.err; echo "n2 rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/bench_n2_one_device.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_n2_one_device.json") if l.startswith("{")][-1])
print({k:d.get(k) for k in ("value","n_gpus","test_mode","time_budget")})
for k,v in d.get("multi_gpu",{}).items(): print(k, {e:(x.get("ms_per_step"), x.get("replicas_agree"), x.get("error")) for e,x in v["exchange"].items()})
PY
