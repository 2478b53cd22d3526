#!/bin/bash
# round 6, session r: the persistent kernels with some waves issuing their partner loads a little later (persist_stagger): A/B
# (value = how * 256 + count of 64-clock sleeps per unit; how 0: waves 4-7, 1: odd waves, 2: waves 2,3,6,7, 4: wave & 3 units)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06r
O=$PWD/gpurun_out/r06r
for rep in 1 2; do
for st in 0 516 520 8 4 1026 1028 260; do
  EMX_AB_TUNE="{\"persist_stagger\": $st}" timeout 300 python tools/ab_cfg.py 20 c4 c2 c2+store 2>&1 | grep -v amdgpu.ids | tee -a $O/stagger_ab5.txt
done
done
