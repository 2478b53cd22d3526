#!/bin/bash
# round 5, session g: device finish: kernel statistics of the exact-mode run at C2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_exact_g -o x -- python $GRAFT_REPO_ROOT/tools/exact_mode_probe.py > $O/prof_exact_g.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/prof_exact_g -name "*kernel_stats.csv" | head -1); head -n 12 "$f" | cut -c1-200
grep ms_per_step $O/prof_exact_g.log | cut -c1-300
