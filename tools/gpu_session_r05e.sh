#!/bin/bash
# round 5, session e: host pipeline with the state-word ring (generator without tempering, 64-wide scan): stage times, exact mode at C2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
for m in "" "EMX_PIPE_GEN_INRING=1"; do
  echo "== $m" >> $O/mt_pipe_host_e.txt
  ( env $m EMX_PIPE_STATS=1 timeout 120 python tools/mt_pipe_bench.py 65536 400 0 ) >> $O/mt_pipe_host_e.txt 2>&1
done
grep -E "==|workers=[46]|workers [0-9] rc" $O/mt_pipe_host_e.txt
timeout 300 python tools/exact_mode_probe.py > $O/exact_c2_e.txt 2>&1; echo "exact rc=$?" | tee -a $O/summary_e.txt
tail -n 3 $O/exact_c2_e.txt
( time timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "exact or mt or golden or fixture" ) > $O/exact_tests_e.log 2>&1; echo "exact tests rc=$?" | tee -a $O/summary_e.txt
tail -n 4 $O/exact_tests_e.log
