#!/bin/bash
# Round 3 closing session: the suite, the driver's bench command (+ --pmc), the rocprofv3 summary of the same command, N>1 control flow
set -u
mkdir -p gpurun_out/r03j
O=gpurun_out/r03j
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log; grep -h "us/step" $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -2 $O/bench_n1.err; head -c 600 $O/bench_n1.json; echo
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc --no-extras --no-cpu-baseline > $O/bench_n1_pmc.json 2> $O/bench_n1_pmc.err; tail -2 $O/bench_n1_pmc.err
python -c "
import json; d=json.load(open('$O/bench_n1_pmc.json')); print(d['roofline']['traffic'], d['roofline']['traffic_source'][:120])"
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof/c2 -o c2 -f csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/all -o all -f csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_all.log 2>&1
find $O/prof -name "*kernel_trace.csv" -exec sh -c 'head -200 "$1" > "$1.head"; rm "$1"' _ {} \;
find $O/prof -name "c2_kernel_stats.csv" -exec head -5 {} \;
timeout 300 python bench.py --force-dist --config c2 --steps 20 --warmup 5 > $O/bench_forcedist_world1.json 2> $O/bench_forcedist.err; tail -1 $O/bench_forcedist.err
timeout 400 python bench.py --gpus 2 --steps 20 --warmup 5 --all-on-device 0 --config c2 > $O/bench_selflaunch_n2_one_device.json 2> $O/bench_n2.err; head -c 700 $O/bench_selflaunch_n2_one_device.json; echo
du -sh $O
