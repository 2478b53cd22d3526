#!/bin/bash
# round 6, session t: the stagger as the default: the persistent-kernel tests, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06t
O=$PWD/gpurun_out/r06t
( time timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_persist_slab.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x ) > $O/gpu_tests_persist.log 2>&1; echo "persist tests rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/gpu_tests_persist.log | cut -c1-200
for rep in 1 2; do
for st in 0 -1; do
  EMX_AB_TUNE="{\"persist_stagger\": $st}" timeout 300 python tools/ab_cfg.py 20 c2 2>&1 | grep -v amdgpu.ids | tee -a $O/stagger_default_ab.txt
  EMX_AB_TUNE="{\"persist_stagger\": $st}" timeout 300 python tools/ab_cfg.py 400 c2 2>&1 | grep -v amdgpu.ids | tee -a $O/stagger_default_ab.txt
done
done
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail_n1.json 2>/dev/null
python - <<'PY' | tee -a $O/summary.txt
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06t"
d = json.loads(open(O + "/bench_n1.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.5f frac %.4f hpl %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["halfsteps_per_launch"]))
for k, c in d["configs"].items():
    print("  %-28s %.2f us/step frac %.3f" % (k, c["ms_per_step"] * 1e3, c["frac"]))
PY
