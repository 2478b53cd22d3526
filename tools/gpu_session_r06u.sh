#!/bin/bash
# round 6, session u2: the blocks of the barrier's words 128 B apart (one page, as shipped), 4 KB apart, 64 KB apart
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06u
O=$PWD/gpurun_out/r06u
for rep in 1 2; do
for lib in base bs1k bs16k; do
  EMX_LIB=$PWD/emcee_amd/libemx_$lib.so timeout 300 python tools/ab_cfg.py 20 c2 c4 c2+store 2>&1 | grep -v amdgpu.ids | tee -a $O/bar_stride_ab.txt
done
done
