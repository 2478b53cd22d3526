#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
rm -f $O/p2p_ab_variants4.txt
for v in A B D; do
  L=$PWD/emcee_amd/libemx_$v.so
  EMX_LIB=$L timeout 300 python tools/exp/p2p_ab.py 800 3 1 2>&1 | grep -v amdgpu.ids | tee -a $O/p2p_ab_variants4.txt
done
for v in As Bs Ds; do
EMX_STAMPS_LIB=$PWD/emcee_amd/libemx_$v.so timeout 300 python tools/persist_phase_clock.py 65536 64 0 1 2>&1 | grep -v amdgpu.ids | tee $O/persist_phase_c2_p2p_$v.txt
done
