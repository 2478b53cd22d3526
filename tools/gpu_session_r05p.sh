#!/bin/bash
# round 5, session p2: two device producers in one process
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_mtdev.py -q -x -p no:cacheprovider ) > $O/tests_p2.log 2>&1; echo "tests p2 rc=$?" | tee -a $O/summary_p.txt
tail -n 25 $O/tests_p2.log
