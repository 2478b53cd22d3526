#!/bin/bash
set -u
mkdir -p gpurun_out/r03t
O=gpurun_out/r03t
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; grep -h "us/step" $O/pytest_gpu.log
timeout 300 python tools/device_callable_bench.py > $O/device_callable.txt 2>&1; cat $O/device_callable.txt
rm -rf $O/prof_user
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_user -o p -f csv -- python tools/callback_profile.py user 400 > $O/prof_user.log 2>&1
grep "us/step" $O/prof_user.log; find $O/prof_user -name "*kernel_stats.csv" -exec head -5 {} \;
find $O/prof_user -name "*kernel_trace.csv" -delete
timeout 200 python tools/ab_cfg.py 10 w512 2>&1 | grep -v amdgpu
