#!/bin/bash
# round 5, session l: exact mode, the plan as one span of the pinned ring (order written in front of the words)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -x -p no:cacheprovider ) > $O/exact_tests_l.log 2>&1; echo "parity+full-size tests rc=$?" | tee -a $O/summary_l.txt
tail -n 4 $O/exact_tests_l.log
for cfg in "EMX_TUNE=mt_ring_raw=1" "EMX_TUNE=mt_ring_raw=0" "EMX_TUNE=mt_ring_raw=1,mt_pipeline=4" "EMX_TUNE=mt_ring_raw=1,mt_pipeline=5"; do
  echo "== $cfg" | tee -a $O/exact_c2_l.txt
  env $cfg timeout 300 python tools/exact_mode_probe.py 2>&1 | tail -n 1 | cut -c1-640 | tee -a $O/exact_c2_l.txt
done
