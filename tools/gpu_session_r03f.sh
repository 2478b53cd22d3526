#!/bin/bash
set -u
mkdir -p gpurun_out/r03f
O=gpurun_out/r03f
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -s -k "device_callable or users_hip" > $O/pytest_cb.log 2>&1; tail -8 $O/pytest_cb.log
timeout 300 python tools/device_callable_bench.py > $O/device_callable.txt 2>&1; cat $O/device_callable.txt
for i in 1 2 3; do
  timeout 120 python tools/ab_cfg.py 20 c4
  EMX_LIB=$PWD/emcee_amd/libemx_hotds.so timeout 120 python tools/ab_cfg.py 20 c4
done > $O/ab_hotds.txt 2>&1; cat $O/ab_hotds.txt
timeout 200 python tools/ab_cfg.py 10 w512 w128 > $O/ab_wide.txt 2>&1; cat $O/ab_wide.txt
du -sh $O
