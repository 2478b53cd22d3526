#!/bin/bash
# round 5: k_persist_p2p variants with the tile words 256 bytes apart -- A/B + phase clock
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
rm -f $O/p2p_ab_variants2.txt
for v in "" p01 p10 p00; do
  L=$PWD/emcee_amd/libemx${v:+_$v}.so
  EMX_LIB=$L timeout 300 python tools/exp/p2p_ab.py 800 3 1 2>&1 | grep -v amdgpu.ids | tee -a $O/p2p_ab_variants2.txt
done
timeout 300 python tools/persist_phase_clock.py 65536 64 0 1 2>&1 | grep -v amdgpu.ids | tee $O/persist_phase_c2_p2p_11b.txt
