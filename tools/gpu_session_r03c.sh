#!/bin/bash
# Round 3, third GPU session: device-callable target, plan batches on a side stream (+ wave priority), snooker prefetch depth
set -u
mkdir -p gpurun_out/r03c
O=gpurun_out/r03c
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -s -k "device_callable or autocorr or handed_back" > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
for i in 1 2 3; do
  timeout 120 python tools/ab_cfg.py 20 c2 c3 c4
  EMX_PLAN_STREAM=1 timeout 120 python tools/ab_cfg.py 20 c2 c3 c4 | sed 's/^cur /side/'
  EMX_PLAN_STREAM=1 EMX_LIB=$PWD/emcee_amd/libemx_noprio.so timeout 120 python tools/ab_cfg.py 20 c2 c3 c4 | sed 's/^libemx_noprio.so/side+noprio     /'
done > $O/ab_side.txt 2>&1
cat $O/ab_side.txt
EMX_PLAN_STREAM=1 timeout 120 python tools/ab_cfg.py 400 c2 c3 c4 c5 > $O/ab_side_k400.txt 2>&1; cat $O/ab_side_k400.txt
timeout 200 python tools/wpb_sweep.py > $O/wpb_sweep.txt 2>&1; tail -12 $O/wpb_sweep.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
EMX_PLAN_STREAM=1 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu_side.log 2>&1; tail -6 $O/pytest_gpu_side.log
du -sh $O
