#!/bin/bash
# round 6, session l: odd ndim above 64 on the slab kernel (tests + timing), a soak of the exact-mode default path, the mtdev tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06l
O=$PWD/gpurun_out/r06l
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_persist_slab.py tests/test_gpu_persist.py tests/test_gpu_mtdev.py tests/test_gpu_regen.py -q -m gpu -p no:cacheprovider ) > $O/tests.log 2>&1; echo "pslab / persist / mtdev / regen tests rc=$?" | tee -a $O/summary.txt
tail -n 8 $O/tests.log | cut -c1-300
timeout 900 python tools/soak_exact.py 20000 3 2>&1 | grep -v amdgpu.ids | tee $O/soak_exact.txt
python - <<'PY' 2>/dev/null | tee $O/odd_slab.txt
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["x", "0"]
import numpy as np, bench
from emcee_amd import _lib
exec(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools", "pslab_bench.py")).read().split("print(")[0])
K = 200
print("dense Gaussian, stretch move, Philox, odd ndim above 64: us/step with k_persist_slab (persist_odd = 1) / per-half-step launches (persist_odd = 0)")
for D in (65, 97, 127):
    for N in (1024, 4096, 16384, 32768, 65536):
        wl = WL(N, D, "stretch")
        row = []
        for tune in ({"persist_odd": 1}, {"persist_odd": 0}):
            r = bench.measure_single(wl, K, 10, want_kernel=False, spin_s=0.05, tuning=tune)
            row.append(r["wall_s"] * 1e6 / K)
        print("%6d x %3d   %7.2f / %7.2f us/step   %.2fx" % (N, D, row[0], row[1], row[1] / row[0]), flush=True)
PY
du -sh $O
