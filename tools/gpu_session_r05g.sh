#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
AB_KEY=persist_flat timeout 600 python tools/exp/p2p_ab.py 1600 4 0 2>&1 | grep -v amdgpu.ids | tee $O/flat_barrier_ab.txt
( time timeout 600 python -m pytest tests/test_gpu_persist.py -q -x -p no:cacheprovider -k "without_a_barrier or gives_the_bits" ) > $O/p2p_tests_g.log 2>&1; echo "tests rc=$?"
tail -n 4 $O/p2p_tests_g.log
