#!/bin/bash
# round 5, session u: the wide dense path with the propose pass made inside the role-split log-prob kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_wide_dense.py tests/test_gpu_full_size.py -q -x -p no:cacheprovider ) > $O/tests_u.log 2>&1; echo "tests u rc=$?" | tee -a $O/summary_u.txt
tail -n 12 $O/tests_u.log
for f in 1 0 1 0; do EMX_TUNE=wide_fuse=$f timeout 200 python tools/ab_cfg.py 20 w512 2>&1 | grep w512 | sed "s/^/wide_fuse=$f /" | tee -a $O/wide_fuse_ab.txt; done
