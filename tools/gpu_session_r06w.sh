#!/bin/bash
# round 6, session w: what the workgroups wait for at the barrier (arrival / release wall-clock stamps, instrumented build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06w
O=$PWD/gpurun_out/r06w
for st in -1 0; do
  EMX_AB_TUNE="{\"persist_stagger\": $st}" timeout 300 python tools/exp/barrier_skew.py 65536 64 2>&1 | grep -v amdgpu.ids | tee -a $O/barrier_skew.txt
done
