#!/bin/bash
# round 6, session u9: k_persist with its plan entries two half-steps ahead (asked for in front of the barrier) against one ahead (in the
# middle of the half-step)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06u
O=$PWD/gpurun_out/r06u
( EMX_LIB=$PWD/emcee_amd/libemx_a1.so timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu -p no:cacheprovider -x ) > $O/gpu_tests_persist_a1.log 2>&1; echo "persist tests (a1) rc=$?" | tee -a $O/summary_u9.txt
tail -n 2 $O/gpu_tests_persist_a1.log | cut -c1-200
for rep in 1 2 3; do
for lib in a0 a1; do
  EMX_LIB=$PWD/emcee_amd/libemx_$lib.so timeout 300 python tools/ab_cfg.py 20 c2 c2+store 2>&1 | grep -v amdgpu.ids | tee -a $O/entries_ahead_ab2.txt
done
done
