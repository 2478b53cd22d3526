#!/bin/bash
# round 6, session y2: k_halfstep_slab (65 536 x 128, the per-half-step launches) with the waves of SIMDs 2 and 3 asking for their rows a
# little later (slab_stagger), with and without the sibling skew it has
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06y
O=$PWD/gpurun_out/r06y
for rep in 1 2; do
  for sk in 1 0; do
  for st in 0 2 4 8 12; do
    EMX_AB_TUNE="{\"slab_skew\": $sk, \"slab_stagger\": $st}" timeout 300 python tools/ab_cfg.py 20 w128 2>&1 | grep -v amdgpu.ids | tee -a $O/slab_stagger_ab.txt
  done
  done
done
