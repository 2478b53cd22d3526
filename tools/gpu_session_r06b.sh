#!/bin/bash
# round 6, session b: the hierarchical device-wide barrier (persist_barrier_hier) -- bit-equality first (the persistent suite, the
# stress tests), then A/B against the arrival counters on C2 / C4 / snooker-only / 32 768 x 64 at the driver's K = 20 and at K = 400;
# the live-reference test and the longer full-size exact runs; the wide dense tests after the fused propose was removed.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06b
O=$PWD/gpurun_out/r06b
R=$PWD
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_persist.py -q -x -m gpu -p no:cacheprovider ) > $O/persist_tests.log 2>&1; echo "persist tests rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/persist_tests.log
for h in 0 1 2; do
  for K in 20 400; do
    EMX_TUNE=persist_hier=$h timeout 300 python tools/ab_cfg.py $K c2 c4 c2+store 2>/dev/null | sed "s/^/hier=$h /" | tee -a $O/ab_hier.txt
  done
done
python - <<'PY' 2>&1 | tee -a $O/ab_hier.txt
# snooker alone and mid-size device-wide shapes, K = 200
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import bench
from emcee_amd import _lib
for h in (0, 1, 2):
    for key, N, moves in (("c4", 65536, "snooker"), ("c2", 32768, None), ("c2", 16384, None), ("c4", 32768, None)):
        wl = bench.Workload(key, N)
        if moves == "snooker":
            wl.moves, wl.weights = [wl.moves[1]], [1.0]
        r = bench.measure_single(wl, 200, 10, want_kernel=False, tuning={"persist_hier": h})
        print("hier=%d %s N=%d %s: %.3f us/step (events %.3f)" % (h, key, N, moves or "", r["wall_s"] * 1e6 / 200, r["gpu_ms"] * 1e3 / 200), flush=True)
PY
( time timeout 900 python -m pytest tests/test_gpu_live_reference.py tests/test_gpu_full_size.py tests/test_gpu_wide_dense.py -q -m gpu -p no:cacheprovider ) > $O/new_tests.log 2>&1; echo "live reference / full size / wide rc=$?" | tee -a $O/summary.txt
tail -n 30 $O/new_tests.log
( time timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_persist.py ) > $O/gpu_tests.log 2>&1; echo "rest of gpu suite rc=$?" | tee -a $O/summary.txt
tail -n 6 $O/gpu_tests.log
du -sh $O
