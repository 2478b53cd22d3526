#!/bin/bash
# round 4: full GPU suite + the driver's bench line at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04i
O=$PWD/gpurun_out/r04i
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt
tail -n 6 $O/gpu_tests.log
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 3000 $O/bench.log
tail -n 5 $O/bench.err
