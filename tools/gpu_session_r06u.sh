#!/bin/bash
# round 6, session u7: launches that store chain rows on k_persist<..., ROWS_LATE> (tuning persist_rows_late), with and without the stagger
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06u
O=$PWD/gpurun_out/r06u
( timeout 600 python -m pytest tests/test_gpu_persist.py tests/test_gpu_sampler_api.py -q -m gpu -p no:cacheprovider -x ) > $O/gpu_tests_persist_late.log 2>&1; echo "persist + sampler api tests rc=$?" | tee -a $O/summary_u7.txt
tail -n 2 $O/gpu_tests_persist_late.log | cut -c1-200
for rep in 1 2; do
for t in '{"persist_rows_late": 0}' '{"persist_rows_late": 1}' '{"persist_rows_late": 1, "persist_stagger": 516}' '{"persist_rows_late": 1, "persist_stagger": 8}' '{"persist_rows_late": 1, "persist_stagger": 1028}'; do
  EMX_AB_TUNE="$t" timeout 300 python tools/ab_cfg.py 20 c2+store 2>&1 | grep -v amdgpu.ids | tee -a $O/rows_late_ab.txt
done
EMX_AB_TUNE='{}' timeout 300 python tools/ab_cfg.py 20 c2 2>&1 | grep -v amdgpu.ids | tee -a $O/rows_late_ab.txt
done
