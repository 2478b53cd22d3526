#!/bin/bash
set -u
mkdir -p gpurun_out/r03q
O=gpurun_out/r03q
for i in 1 2 3; do
  timeout 200 python tools/ab_cfg.py 20 c2+store c3+store c5+store
  EMX_LIB=$PWD/emcee_amd/libemx_nt0.so timeout 200 python tools/ab_cfg.py 20 c2+store c3+store c5+store
done > $O/ab_nt.txt 2>&1; grep -v amdgpu.ids $O/ab_nt.txt
