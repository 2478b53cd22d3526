#!/bin/bash
# round 6, session j: regen tests again, exact mode at C2 with the sixth finisher, the whole GPU suite, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06j
O=$PWD/gpurun_out/r06j
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_regen.py -q -m gpu -p no:cacheprovider ) > $O/regen_tests.log 2>&1; echo "regen tests rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/regen_tests.log | cut -c1-300
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/exact_c2.txt
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
for N in (65536, 32768):
    wl = bench.Workload("c2", N)
    for rep in range(3):
        for name, tune in (("persistent+regen, finishers auto", {}), ("persistent+regen, 5 finishers", {"mt_pipeline": 5}), ("persistent+regen, 7 finishers", {"mt_pipeline": 7})):
            r = bench.measure_single(wl, 400, 40, rng="mt19937", spin_s=0.05, want_kernel=False, tuning=tune)
            p = r.get("pipeline") or {}
            print("N=%6d %-34s %.2f us/step (best %.2f)  generator %.1f tokenizer %.1f finishers(sum) %.1f [%d] tok-waits-words %.1f tok-waits-consumer %.1f" % (
                N, name, r["wall_s"] * 1e6 / 400, r["wall_min_s"] * 1e6 / 400, p.get("generator_us", 0), p.get("tokenizer_us", 0),
                p.get("finishers_us_summed", 0), p.get("finisher_threads", 0), p.get("tokenizer_waited_for_words_us", 0), p.get("tokenizer_waited_for_consumer_us", 0)), flush=True)
PY
( time timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -n 6 $O/gpu_tests.log | cut -c1-250
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail_n1.json 2>/dev/null
wc -c $O/bench_n1.json | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06j"
d = json.loads(open(O + "/bench_n1.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.5f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for k, c in d["configs"].items():
    print("  %-28s %.2f us/step frac %.3f" % (k, c["ms_per_step"] * 1e3, c["frac"]))
print("  exact:", json.dumps(d["exact_mode"]))
PY
du -sh $O
