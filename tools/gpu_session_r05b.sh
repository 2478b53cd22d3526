#!/bin/bash
# round 5, session b: the device-side exchange tests between processes (0 skips), the device producer's tests after the status-bit
# change, host facts of the GPU box, host -> device rates at plan-upload sizes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( lscpu; echo; nproc; cat /sys/devices/system/cpu/cpu0/cache/index3/shared_cpu_list; cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list; free -g ) > $O/host_facts.txt 2>&1
timeout 120 tools/ubench/bin/h2d_rate > $O/h2d_rate.txt 2>&1; echo "h2d rc=$?" | tee -a $O/summary_b.txt
cat $O/h2d_rate.txt
( time timeout 900 python -m pytest tests/test_gpu_sharded.py -q -x -p no:cacheprovider -k "device_side" ) > $O/device_side_tests.log 2>&1; echo "device_side tests rc=$?" | tee -a $O/summary_b.txt
tail -n 25 $O/device_side_tests.log
( time timeout 600 python -m pytest tests/test_gpu_mtdev.py tests/test_gpu_direct_ipc.py -q -x -p no:cacheprovider ) > $O/mtdev_tests.log 2>&1; echo "mtdev+ipc tests rc=$?" | tee -a $O/summary_b.txt
tail -n 8 $O/mtdev_tests.log
head -n 30 $O/host_facts.txt
