#!/bin/bash
# round 5, session h: host pipeline + device finish: rates only
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( EMX_PIPE_STATS=1 timeout 120 python tools/mt_pipe_bench.py 65536 400 0 ) > $O/mt_pipe_host_h.txt 2>&1
grep -E "workers=[46]|workers [0-9] rc" $O/mt_pipe_host_h.txt | cut -c1-400
timeout 300 python tools/exact_mode_probe.py > $O/exact_c2_h.txt 2>&1; echo "exact rc=$?" | tee -a $O/summary_h.txt
tail -n 2 $O/exact_c2_h.txt
EMX_TUNE=mt_device_finish=0 timeout 300 python tools/exact_mode_probe.py > $O/exact_c2_h_hostfinish.txt 2>&1
tail -n 2 $O/exact_c2_h_hostfinish.txt
