#!/bin/bash
# round 6, session f: the persistent slab and odd-ndim tests on the final rules, the whole GPU suite, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06f
O=$PWD/gpurun_out/r06f
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_persist_slab.py -q -m gpu -p no:cacheprovider ) > $O/pslab_tests.log 2>&1; echo "persistent slab / odd tests rc=$?" | tee -a $O/summary.txt
tail -n 6 $O/pslab_tests.log | cut -c1-250
( time timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_persist_slab.py ) > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -n 6 $O/gpu_tests.log | cut -c1-250
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail_n1.json 2>/dev/null
wc -c $O/bench_n1.json | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06f"
d = json.loads(open(O + "/bench_n1.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.5f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for k, c in d["configs"].items():
    print("  %-28s %.2f us/step frac %.3f" % (k, c["ms_per_step"] * 1e3, c["frac"]))
print("  exact:", json.dumps(d["exact_mode"]))
PY
# mid-size odd ndim: before / after
python - <<'PY' 2>/dev/null | tee $O/odd_ndim.txt
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["x", "0"]
import numpy as np, bench
from emcee_amd import _lib
exec(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools", "pslab_bench.py")).read().split("print(")[0])
K = 200
print("dense Gaussian, stretch move, Philox: us/step with k_persist (persist_odd = 1) / per-half-step launches (persist_odd = 0)")
for D in (5, 17, 33, 47, 63):
    for N in (1024, 4096, 16384, 65536):
        wl = WL(N, D, "stretch")
        row = []
        for tune in ({"persist_odd": 1, "small_kernel": 0}, {"persist_odd": 0, "small_kernel": 0}):
            r = bench.measure_single(wl, K, 10, want_kernel=False, spin_s=0.05, tuning=tune)
            row.append(r["wall_s"] * 1e6 / K)
        print("%6d x %3d   %7.2f / %7.2f us/step   %.2fx" % (N, D, row[0], row[1], row[1] / row[0]), flush=True)
PY
du -sh $O
