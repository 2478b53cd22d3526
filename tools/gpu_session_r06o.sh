#!/bin/bash
# round 6, session o: the GPU suite with durations (which tests made it 414 s?), after the side-stream key was removed
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06o
O=$PWD/gpurun_out/r06o
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=40 ) > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -n 60 $O/gpu_tests.log | cut -c1-200
