#!/bin/bash
# round 6, session u4: a copy of the barrier's `go` word per XCD (4 KB apart; the elected last arriver writes all eight, a workgroup polls
# its XCD's) against the one word everybody polls
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06u
O=$PWD/gpurun_out/r06u
( EMX_LIB=$PWD/emcee_amd/libemx_rep8.so timeout 600 python -m pytest tests/test_gpu_persist.py -q -m gpu -p no:cacheprovider -x ) > $O/gpu_tests_persist_rep8.log 2>&1; echo "persist tests (rep8) rc=$?" | tee -a $O/summary_u4.txt
tail -n 2 $O/gpu_tests_persist_rep8.log | cut -c1-200
for rep in 1 2 3; do
for lib in base rep8; do
  EMX_LIB=$PWD/emcee_amd/libemx_$lib.so timeout 300 python tools/ab_cfg.py 20 c2 c4 c2+store 2>&1 | grep -v amdgpu.ids | tee -a $O/bar_replicas_ab.txt
done
done
