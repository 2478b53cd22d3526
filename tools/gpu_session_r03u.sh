#!/bin/bash
set -u
mkdir -p gpurun_out/r03u
O=gpurun_out/r03u
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
