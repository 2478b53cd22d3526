#!/bin/bash
# round 6, session m: 40 half-steps per persistent launch (PersistIter shrunk) -- the persistent suites, then C2 / C4 / C2 + store at
# K = 20 and 400, odd ndim at 65 536 walkers on the slab kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06m
O=$PWD/gpurun_out/r06m
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_persist.py tests/test_gpu_persist_slab.py tests/test_gpu_regen.py tests/test_gpu_full_size.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider ) > $O/tests.log 2>&1; echo "persistent suites rc=$?" | tee -a $O/summary.txt
tail -n 8 $O/tests.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
for K in 20 400; do timeout 400 python tools/ab_cfg.py $K c2 c4 c2+store w128 2>/dev/null | tee -a $O/ab_iters40.txt; done
python - <<'PY' 2>/dev/null | tee $O/odd_slab_65536.txt
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["x", "0"]
import numpy as np, bench
from emcee_amd import _lib
exec(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools", "pslab_bench.py")).read().split("print(")[0])
for D in (97, 127, 65):
    wl = WL(65536, D, "stretch")
    row = []
    for tune in ({"persist_odd": 1}, {"persist_odd": 0}):
        r = bench.measure_single(wl, 200, 10, want_kernel=False, spin_s=0.05, tuning=tune)
        row.append(r["wall_s"] * 1e6 / 200)
    print("65536 x %3d   %7.2f / %7.2f us/step   %.2fx" % (D, row[0], row[1], row[1] / row[0]), flush=True)
PY
du -sh $O
