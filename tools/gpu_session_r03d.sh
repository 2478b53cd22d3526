#!/bin/bash
set -u
mkdir -p gpurun_out/r03d
O=gpurun_out/r03d
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
for t in 0 256 512 1024; do
  EMX_TUNE=ablate=$t timeout 120 python tools/ab_cfg.py 20 c2 c3 | sed "s/^cur /ablate=$t /"
done > $O/plan_ablate.txt 2>&1; cat $O/plan_ablate.txt
for i in 1 2 3; do
  timeout 120 python tools/ab_cfg.py 20 c4
  EMX_LIB=$PWD/emcee_amd/libemx_sn48.so timeout 120 python tools/ab_cfg.py 20 c4
done > $O/ab_snooker.txt 2>&1; cat $O/ab_snooker.txt
timeout 300 python tools/device_callable_bench.py > $O/device_callable.txt 2>&1; cat $O/device_callable.txt
du -sh $O
