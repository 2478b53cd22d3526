#!/bin/bash
# round 5, session f: exact-mode plans finished on the device (k_plan_raw / k_plan_comp over the pinned state-word ring): parity, rate
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -x -p no:cacheprovider ) > $O/exact_tests_f.log 2>&1; echo "parity+full-size tests rc=$?" | tee -a $O/summary_f.txt
tail -n 6 $O/exact_tests_f.log
( EMX_PIPE_STATS=1 timeout 120 python tools/mt_pipe_bench.py 65536 400 0 ) > $O/mt_pipe_host_f.txt 2>&1
grep -E "workers=[46]|workers [0-9] rc" $O/mt_pipe_host_f.txt
timeout 300 python tools/exact_mode_probe.py > $O/exact_c2_f.txt 2>&1; echo "exact rc=$?" | tee -a $O/summary_f.txt
tail -n 3 $O/exact_c2_f.txt
EMX_TUNE=mt_device_finish=0 timeout 300 python tools/exact_mode_probe.py > $O/exact_c2_f_hostfinish.txt 2>&1
tail -n 2 $O/exact_c2_f_hostfinish.txt
