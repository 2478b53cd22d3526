#!/bin/bash
# round 6, session i: exact mode on the persistent kernel with regen steps -- tests, then C2 / 32 768 / 131 072 against the per-step path
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06i
O=$PWD/gpurun_out/r06i
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_regen.py -q -x -m gpu -p no:cacheprovider ) > $O/regen_tests.log 2>&1; echo "regen tests rc=$?" | tee -a $O/summary.txt
tail -n 12 $O/regen_tests.log | cut -c1-300
( time timeout 1200 python -m pytest tests/test_gpu_persist.py tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_gpu_mtdev.py -q -m gpu -p no:cacheprovider -k "exact or mt or qualify" ) > $O/exact_tests.log 2>&1; echo "exact-mode tests rc=$?" | tee -a $O/summary.txt
tail -n 6 $O/exact_tests.log | cut -c1-300
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/exact_persist_regen.txt
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
for N in (65536, 32768, 16384, 131072):
    wl = bench.Workload("c2", N)
    for rep in range(2):
        for name, tune in (("persistent+regen", {}), ("per-step+regen(side1)", {"persist_exact": 0, "mt_regen_side": 1}), ("per-step+regen(side0)", {"persist_exact": 0}),
                           ("round-5 path", {"mt_regen_min_walkers": 0})):
            r = bench.measure_single(wl, 400, 40, rng="mt19937", spin_s=0.05, want_kernel=False, tuning=tune)
            p = r.get("pipeline") or {}
            print("N=%6d %-24s %.2f us/step (best %.2f)  generator %.1f tokenizer %.1f finishers(sum) %.1f tok-waits-words %.1f tok-waits-consumer %.1f  hpl %.1f" % (
                N, name, r["wall_s"] * 1e6 / 400, r["wall_min_s"] * 1e6 / 400, p.get("generator_us", 0), p.get("tokenizer_us", 0),
                p.get("finishers_us_summed", 0), p.get("tokenizer_waited_for_words_us", 0), p.get("tokenizer_waited_for_consumer_us", 0), r.get("halfsteps_per_launch", 1.0)), flush=True)
PY
du -sh $O
