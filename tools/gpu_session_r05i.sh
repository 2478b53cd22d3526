#!/bin/bash
# round 5, session i: exact-mode plans finished on the device, four steps per upload: parity, rate
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_sampler_api.py -q -x -p no:cacheprovider ) > $O/exact_tests_i.log 2>&1; echo "parity+full-size+api tests rc=$?" | tee -a $O/summary_i.txt
tail -n 6 $O/exact_tests_i.log
timeout 300 python tools/exact_mode_probe.py > $O/exact_c2_i.txt 2>&1; echo "exact rc=$?" | tee -a $O/summary_i.txt
tail -n 2 $O/exact_c2_i.txt
EMX_TUNE=mt_device_finish=0 timeout 300 python tools/exact_mode_probe.py > $O/exact_c2_i_hostfinish.txt 2>&1
tail -n 2 $O/exact_c2_i_hostfinish.txt
