#!/bin/bash
set -u
mkdir -p gpurun_out/r03l
O=gpurun_out/r03l
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 200 python tools/exact_mode_probe.py 2>&1 | grep ms_per_step
