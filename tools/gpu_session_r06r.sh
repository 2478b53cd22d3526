#!/bin/bash
# round 6, session r6: k_persist_mix (c4) with the staggered partner loads, longer waits
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06r
O=$PWD/gpurun_out/r06r
for rep in 1 2 3; do
for st in 0 1028 1030 1032 524 528 536 1036; do
  EMX_AB_TUNE="{\"persist_stagger\": $st}" timeout 300 python tools/ab_cfg.py 20 c4 2>&1 | grep -v amdgpu.ids | tee -a $O/stagger_ab6.txt
done
done
