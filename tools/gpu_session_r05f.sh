#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
rm -f $O/skew_ab.txt
for v in K0 K24 K40 K56 K0 K40; do
  L=$PWD/emcee_amd/libemx_$v.so
  EMX_LIB=$L timeout 300 python tools/exp/p2p_ab.py 800 3 1 2>&1 | grep -v amdgpu.ids | tee -a $O/skew_ab.txt
done
