#!/bin/bash
# round 5, session g: k_plan_fetch as few workgroups beside a device-wide persistent launch, or the copy engine: exact-mode tests, the A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05g
O=$PWD/gpurun_out/r05g
export TMPDIR=/tmp
( time EMX_TUNE=fetch_copy=1 timeout 900 python -m pytest tests/test_gpu_persist.py -q -p no:cacheprovider -x -k "exact or mt or redone" ) > $O/tests_copy.log 2>&1; echo "tests (fetch_copy=1) rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/tests_copy.log
timeout 900 python tools/exp/fetch_blocks_ab.py 64,-1,0 > $O/fetch_copy_ab.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep -v amdgpu.ids $O/fetch_copy_ab.txt
