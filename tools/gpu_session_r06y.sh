#!/bin/bash
# round 6, session y: k_persist_slab at 65 536 x 128 (persist_slab = 2) with the SIMD-pair stagger, with and without its sibling skew,
# against the per-half-step slab launches
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06y
O=$PWD/gpurun_out/r06y
for rep in 1 2; do
  EMX_AB_TUNE='{}' timeout 300 python tools/ab_cfg.py 20 w128 2>&1 | grep -v amdgpu.ids | tee -a $O/pslab_stagger_ab.txt
  for sk in 1 0; do
  for st in 0 516 520 528 1028; do
    EMX_AB_TUNE="{\"persist_slab\": 2, \"persist_slab_skew\": $sk, \"persist_stagger\": $st}" timeout 300 python tools/ab_cfg.py 20 w128 2>&1 | grep -v amdgpu.ids | tee -a $O/pslab_stagger_ab.txt
  done
  done
done
