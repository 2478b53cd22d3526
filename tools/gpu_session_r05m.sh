#!/bin/bash
# round 5, session m: the snooker move in one pass (one round of group reductions, |q - z| in closed form): whole GPU suite, C4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -q -x -m gpu -p no:cacheprovider ) > $O/gpu_tests_m.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary_m.txt
tail -n 6 $O/gpu_tests_m.log
timeout 300 python tools/ab_cfg.py 20 c4 > $O/c4_m.txt 2>&1; tail -n 3 $O/c4_m.txt
timeout 300 python tools/exp/mix_probe.py > $O/mix_probe_m.txt 2>&1; tail -n 12 $O/mix_probe_m.txt
