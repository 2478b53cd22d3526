#!/bin/bash
# round 5, session p: exact-mode launches that give up are redone
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_persist.py -q -x -p no:cacheprovider -k "redone or exact_mode" ) > $O/tests_p.log 2>&1; echo "tests p rc=$?" | tee -a $O/summary_p.txt
tail -n 25 $O/tests_p.log
