#!/bin/bash
# round 6, session s: the phase clock of k_persist with and without the stagger, for an early wave (0) and a late one (2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06s
O=$PWD/gpurun_out/r06s
for st in 0 516 1028; do
for lib in stamps stamps2; do
  EMX_STAMPS_LIB=$PWD/emcee_amd/libemx_$lib.so EMX_AB_TUNE="{\"persist_stagger\": $st}" timeout 300 python tools/persist_phase_clock.py 65536 64 0 2>&1 | grep -v amdgpu.ids | tee -a $O/persist_phase_stagger.txt
done
done
