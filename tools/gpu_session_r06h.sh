#!/bin/bash
# round 6, session h: where k_plan_regen runs -- the consumer's stream (0), the upload stream under the step before (1), k_plan_raw
# there too (2) -- exact mode at 65 536 / 32 768 / 131 072 walkers; kernel statistics of the exact-mode C2 run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06h
O=$PWD/gpurun_out/r06h
R=$PWD
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_regen.py -q -m gpu -p no:cacheprovider ) > $O/regen_tests.log 2>&1; echo "regen tests (side 0) rc=$?" | tee -a $O/summary.txt
( time EMX_TUNE=mt_regen_side=2 timeout 600 python -m pytest tests/test_gpu_regen.py -q -m gpu -p no:cacheprovider ) > $O/regen_tests_side2.log 2>&1; echo "regen tests (side 2) rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/regen_tests_side2.log | cut -c1-200
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/exact_regen_side.txt
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
for N in (65536, 32768, 131072):
    wl = bench.Workload("c2", N)
    for rep in range(2):
        for regen, side in ((16384, 0), (16384, 1), (16384, 2), (0, 0)):
            r = bench.measure_single(wl, 400, 40, rng="mt19937", spin_s=0.05, want_kernel=False, tuning={"mt_regen_min_walkers": regen, "mt_regen_side": side})
            p = r.get("pipeline") or {}
            print("N=%6d regen_min=%5d side=%d: %.2f us/step (best %.2f)  generator %.1f tokenizer %.1f finishers(sum) %.1f tok-waits-words %.1f tok-waits-consumer %.1f" % (
                N, regen, side, r["wall_s"] * 1e6 / 400, r["wall_min_s"] * 1e6 / 400, p.get("generator_us", 0), p.get("tokenizer_us", 0),
                p.get("finishers_us_summed", 0), p.get("tokenizer_waited_for_words_us", 0), p.get("tokenizer_waited_for_consumer_us", 0)), flush=True)
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_exact -o ex -f csv -- python $R/tools/exact_mode_probe.py > $O/trace_exact.log 2>&1; echo "trace rc=$?" | tee -a $O/summary.txt
cd $R
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*.db" -delete
f=$(find $O/trace_exact -name "*kernel_stats.csv" | head -1); head -8 $f
du -sh $O
