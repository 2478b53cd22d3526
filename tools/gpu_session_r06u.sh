#!/bin/bash
# round 6, session u3: the barrier without the second election (`go` counts the XCDs' last arrivers) against the three-trip one
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06u
O=$PWD/gpurun_out/r06u
( timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu -p no:cacheprovider -x ) > $O/gpu_tests_persist_two.log 2>&1; echo "persist tests (product build = two trips) rc=$?" | tee -a $O/summary_u3.txt
tail -n 2 $O/gpu_tests_persist_two.log | cut -c1-200
for rep in 1 2; do
for lib in three two; do
  EMX_LIB=$PWD/emcee_amd/libemx_$lib.so timeout 300 python tools/ab_cfg.py 20 c2 c4 c2+store 2>&1 | grep -v amdgpu.ids | tee -a $O/bar_two_trip_ab.txt
  EMX_LIB=$PWD/emcee_amd/libemx_$lib.so timeout 300 python tools/ab_cfg.py 400 c2 2>&1 | grep -v amdgpu.ids | tee -a $O/bar_two_trip_ab.txt
done
done
