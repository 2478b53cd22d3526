#!/bin/bash
# round 5, session x: k_plan_fetch by ticket, off the one-XCD launch's XCD: the exact-mode tests, the A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05x
O=$PWD/gpurun_out/r05x
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_parity.py tests/test_gpu_small_run.py tests/test_gpu_sampler_api.py -q -p no:cacheprovider -x ) > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/tests.log
timeout 600 python tools/exp/fetch_avoid_ab.py > $O/fetch_avoid_ab.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep -v amdgpu.ids $O/fetch_avoid_ab.txt
