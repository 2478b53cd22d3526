#!/bin/bash
# round 5, session o: long equality runs of this round's paths, the soak run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
timeout 900 python tools/exp/long_equal_check.py > $O/long_equal_check.txt 2>&1; echo "long equal rc=$?" | tee -a $O/summary_o.txt
grep -v amdgpu.ids $O/long_equal_check.txt
timeout 600 python tools/soak.py > $O/soak.json 2> $O/soak.err; echo "soak rc=$?" | tee -a $O/summary_o.txt
tail -c 1500 $O/soak.json
