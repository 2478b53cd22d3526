#!/bin/bash
# Round 3 closing session after the persistent kernel: the suite, the driver's bench command (+ --pmc), rocprofv3 summaries of the same command
set -u
mkdir -p gpurun_out/r03z
O=gpurun_out/r03z
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -2 $O/bench_n1.err; head -c 400 $O/bench_n1.json; echo
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc --no-extras --no-cpu-baseline > $O/bench_n1_pmc.json 2> $O/bench_n1_pmc.err; tail -2 $O/bench_n1_pmc.err
python -c "
import json; d=json.load(open('$O/bench_n1_pmc.json')); r=d['roofline']; print(r['traffic'], r['halfsteps_per_launch'], r['traffic_source'][:200])"
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof/c2 -o c2 -f csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/all -o all -f csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_all.log 2>&1
find $O/prof -name "*kernel_trace.csv" -exec sh -c 'head -200 "$1" > "$1.head"; rm "$1"' _ {} \;
find $O/prof -name "c2_kernel_stats.csv" -exec head -6 {} \;
timeout 400 python bench.py --gpus 2 --steps 20 --warmup 5 --all-on-device 0 --config c2 > $O/bench_selflaunch_n2_one_device.json 2> $O/bench_n2.err; head -c 500 $O/bench_selflaunch_n2_one_device.json; echo
du -sh $O
