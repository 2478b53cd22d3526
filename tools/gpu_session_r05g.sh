#!/bin/bash
# round 5, session g: k_plan_fetch as few workgroups beside a device-wide persistent launch (tuning fetch_blocks): the exact-mode
# persistent tests, then the A/B of workgroup counts at 8 192 / 16 384 / 32 768 walkers with the Philox rate of each shape.
# (The runs recorded in profiles/r05/exact_mix_probe.txt (5) used the value lists 64,0,16,256 and 64,32,96,128, and a build that
# also had the copy-engine variant as -1.)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05g
O=$PWD/gpurun_out/r05g
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_parity.py -q -p no:cacheprovider -x ) > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/tests.log
timeout 900 python tools/exp/fetch_blocks_ab.py "${1:-64,0}" > $O/fetch_blocks_ab.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep -v amdgpu.ids $O/fetch_blocks_ab.txt
