#!/bin/bash
set -u
mkdir -p gpurun_out/r03k
O=gpurun_out/r03k
export TMPDIR=/tmp
for k in -1 6 8 3; do
  echo "mt_pipeline=$k"
  EMX_TUNE=mt_pipeline=$k timeout 200 python tools/exact_mode_probe.py 2>&1 | grep ms_per_step
done > $O/exact_finishers.txt 2>&1; cat $O/exact_finishers.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "sharded or direct_ipc or device_callable" > $O/pytest_part.log 2>&1; tail -3 $O/pytest_part.log
