#!/bin/bash
# round 5, session d: slab skew threshold sweep; host pipeline after the generator bursts; exact mode at C2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
timeout 300 python tools/exp/slab_ab.py slab_skew 3 0,1,2,3,4 > $O/slab_skew_sweep.txt 2>&1; echo "slab sweep rc=$?" | tee -a $O/summary_d.txt
cat $O/slab_skew_sweep.txt
( EMX_PIPE_STATS=1 timeout 120 python tools/mt_pipe_bench.py 65536 400 0 ) > $O/mt_pipe_host_d.txt 2>&1
grep -E "workers=6|workers [0-9] rc" $O/mt_pipe_host_d.txt
timeout 300 python tools/exact_mode_probe.py > $O/exact_c2_d.txt 2>&1; echo "exact rc=$?" | tee -a $O/summary_d.txt
tail -n 3 $O/exact_c2_d.txt
