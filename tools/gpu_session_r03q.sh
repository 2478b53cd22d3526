#!/bin/bash
# Round 3 last session (snooker / DE steps persistent too): the suite, the driver's bench command, rocprofv3 summary of the full line
set -u
mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
export TMPDIR=/tmp
timeout 200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; grep -h "passed\|failed" $O/pytest_gpu.log | tail -2
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -1 $O/bench_n1.err; head -c 300 $O/bench_n1.json; echo
timeout 110 rocprofv3 --kernel-trace --stats -d $O/prof/all -o all -f csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_all.log 2>&1
find $O/prof -name "*kernel_trace.csv" -exec rm {} \;
find $O/prof -name "all_kernel_stats.csv" -exec head -8 {} \; | cut -c1-160
