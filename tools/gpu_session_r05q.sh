#!/bin/bash
# round 5, session q: the slab kernel with its image staged in one batch of loads (csrc/emx_slab.hip): its tests, the A/B of its
# skew settings, the mid-size bench entry
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05q
O=$PWD/gpurun_out/r05q
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_wide_dense.py -q -p no:cacheprovider -x ) > $O/slab_tests.log 2>&1; echo "slab tests rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/slab_tests.log
timeout 600 python tools/exp/slab_ab.py slab_skew 3 0,1,2 > $O/slab_ab.txt 2>&1; echo "slab ab rc=$?" | tee -a $O/summary.txt
grep -v amdgpu.ids $O/slab_ab.txt
timeout 300 python tools/ab_cfg.py 20 w128 c2 > $O/ab_cfg.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
grep -v amdgpu.ids $O/ab_cfg.txt
