#!/bin/bash
# round 6, session k: the size rule between the host pipeline (with regen steps) and the device producer, exact mode, Rosenbrock ndim 32
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06k
O=$PWD/gpurun_out/r06k
export TMPDIR=/tmp
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/mtdev_sizes_r06.txt
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
for N in (131072, 262144, 524288, 1048576):
    wl = bench.Workload("c3", N)
    for rep in range(2):
        for name, tune in (("host pipeline + regen", {"mt_device": 0}), ("device producer", {"mt_device": 2}), ("host pipeline, words copied", {"mt_device": 0, "mt_regen_min_walkers": 0})):
            K = 100 if N <= 262144 else 40
            r = bench.measure_single(wl, K, 10, rng="mt19937", spin_s=0.05, want_kernel=False, tuning=tune)
            p = r.get("pipeline") or {}
            print("N=%8d %-30s %.1f us/step (best %.1f)  generator %.1f tokenizer %.1f finishers(sum) %.1f [%d]" % (
                N, name, r["wall_s"] * 1e6 / K, r["wall_min_s"] * 1e6 / K, p.get("generator_us", 0), p.get("tokenizer_us", 0),
                p.get("finishers_us_summed", 0), p.get("finisher_threads", 0)), flush=True)
PY
du -sh $O
