#!/bin/bash
# round 5: k_persist_p2p variants with a sleep between polls -- A/B + phase clocks
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
rm -f $O/p2p_ab_variants3.txt
for v in A B C D E; do
  L=$PWD/emcee_amd/libemx_$v.so
  EMX_LIB=$L timeout 300 python tools/exp/p2p_ab.py 800 3 1 2>&1 | grep -v amdgpu.ids | tee -a $O/p2p_ab_variants3.txt
done
EMX_STAMPS_LIB=$PWD/emcee_amd/libemx_Bs.so timeout 300 python tools/persist_phase_clock.py 65536 64 0 1 2>&1 | grep -v amdgpu.ids | tee $O/persist_phase_c2_p2p_B.txt
EMX_STAMPS_LIB=$PWD/emcee_amd/libemx_Ds.so timeout 300 python tools/persist_phase_clock.py 65536 64 0 1 2>&1 | grep -v amdgpu.ids | tee $O/persist_phase_c2_p2p_D.txt
