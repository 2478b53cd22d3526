#!/bin/bash
# round 6, session g: k_plan_regen -- bit-equality (tests/test_gpu_regen.py, the full-size exact runs), then exact mode at C2 with the
# draws made again on the device against the words copied (tuning mt_regen_min_walkers), stage times of the host pipeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06g
O=$PWD/gpurun_out/r06g
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_regen.py tests/test_gpu_persist_slab.py -q -m gpu -p no:cacheprovider ) > $O/regen_tests.log 2>&1; echo "regen + pslab tests rc=$?" | tee -a $O/summary.txt
tail -n 12 $O/regen_tests.log | cut -c1-300
( time timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "exact" ) > $O/exact_tests.log 2>&1; echo "exact-mode parity tests rc=$?" | tee -a $O/summary.txt
tail -n 5 $O/exact_tests.log | cut -c1-300
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/exact_regen_ab.txt
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
for N in (65536, 32768, 131072):
    wl = bench.Workload("c2", N)
    for rep in range(2):
        for regen in (16384, 0):
            r = bench.measure_single(wl, 400, 40, rng="mt19937", spin_s=0.05, want_kernel=False, tuning={"mt_regen_min_walkers": regen})
            p = r.get("pipeline") or {}
            print("N=%6d regen_min=%5d: %.2f us/step (best %.2f)  generator %.1f tokenizer %.1f finishers(sum) %.1f tok-waits-words %.1f  [%d finishers]" % (
                N, regen, r["wall_s"] * 1e6 / 400, r["wall_min_s"] * 1e6 / 400, p.get("generator_us", 0), p.get("tokenizer_us", 0),
                p.get("finishers_us_summed", 0), p.get("tokenizer_waited_for_words_us", 0), p.get("finisher_threads", 0)), flush=True)
PY
du -sh $O
