#!/bin/bash
# round 5, session n2: generator / tokenizer spin window against the default, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
for rep in 1 2 3; do
for cfg in "X=1" "EMX_PIPE_GT_SPIN_US=200" "EMX_PIPE_GT_SPIN_US=500"; do
  echo "== $cfg" | tee -a $O/exact_c2_n2.txt
  env $cfg timeout 120 python tools/exact_mode_probe.py 2>&1 | tail -n 1 | cut -c1-200 | tee -a $O/exact_c2_n2.txt
done
done
