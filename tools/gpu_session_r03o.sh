#!/bin/bash
set -u
mkdir -p gpurun_out/r03o
O=gpurun_out/r03o
timeout 600 python -m pytest tests/test_gpu_direct_ipc.py -m gpu -q -p no:cacheprovider -k sampler_api > $O/pytest.log 2>&1; grep -h "SAMPLER_REPLAY\|passed\|failed" $O/pytest.log | cut -c1-300
