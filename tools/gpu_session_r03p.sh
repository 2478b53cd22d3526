#!/bin/bash
set -u
mkdir -p gpurun_out/r03p
O=gpurun_out/r03p
export TMPDIR=/tmp
timeout 200 python tools/ab_cfg.py 10 w128 > $O/w128.txt 2>&1
EMX_TUNE=dense_wide=1 timeout 200 python tools/ab_cfg.py 10 w128 | sed 's/^cur /wide/' >> $O/w128.txt 2>&1
cat $O/w128.txt
timeout 200 python tools/dense_crossover_probe.py > $O/crossover.txt 2>&1; tail -12 $O/crossover.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
