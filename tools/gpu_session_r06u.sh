#!/bin/bash
# round 6, session u8: k_persist_mix with the next half-step's own rows asked for behind the MFMA phase (c4, and c4 with the chain stored)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06u
O=$PWD/gpurun_out/r06u
for rep in 1 2 3; do
for lib in m0 m1; do
  EMX_LIB=$PWD/emcee_amd/libemx_$lib.so timeout 300 python tools/ab_cfg.py 20 c4 c4+store 2>&1 | grep -v amdgpu.ids | tee -a $O/mix_rows_late_ab.txt
done
done
