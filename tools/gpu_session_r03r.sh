#!/bin/bash
set -u
mkdir -p gpurun_out/r03r
O=gpurun_out/r03r
for i in 1 2 3; do
  timeout 200 python tools/ab_cfg.py 20 c2 c3 c3+store
  EMX_LIB=$PWD/emcee_amd/libemx_ntplan.so timeout 200 python tools/ab_cfg.py 20 c2 c3 c3+store
done > $O/ab_ntplan.txt 2>&1; grep -v amdgpu.ids $O/ab_ntplan.txt
