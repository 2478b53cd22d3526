#!/bin/bash
# round 6, session e: the persistent slab kernel with the skewed start and the device-wide form from 8 192 walkers on -- tests, then
# skew on / off at the sizes where it matters, then the table of tools/pslab_bench.py
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06e
O=$PWD/gpurun_out/r06e
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_persist_slab.py -q -m gpu -p no:cacheprovider ) > $O/pslab_tests.log 2>&1; echo "persistent slab tests rc=$?" | tee -a $O/summary.txt
tail -n 8 $O/pslab_tests.log | cut -c1-250
python - <<'PY' 2>/dev/null | tee $O/pslab_skew.txt
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["x", "0"]
import numpy as np, bench
from emcee_amd import _lib
exec(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools", "pslab_bench.py")).read().split("print(")[0])
K = 200
for move in ("stretch", "de"):
    for D in (128, 96):
        for N in (8192, 32768, 65536):
            wl = WL(N, D, move)
            row = []
            for tune in ({"persist_slab": 1, "persist_slab_skew": 0}, {"persist_slab": 1, "persist_slab_skew": 1}, {"persist_slab": 1, "persist_slab_skew": 3}, {"persist_slab": 0}):
                r = bench.measure_single(wl, K, 10, want_kernel=False, spin_s=0.05, tuning=tune)
                row.append(r["wall_s"] * 1e6 / K)
            print("%-28s persistent skew 0 / 1 / 3: %.2f / %.2f / %.2f us/step   per-half-step %.2f" % (wl.label, row[0], row[1], row[2], row[3]), flush=True)
PY
timeout 1500 python tools/pslab_bench.py 200 2>/dev/null | tee $O/pslab_bench.txt
du -sh $O
