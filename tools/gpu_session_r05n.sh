#!/bin/bash
# round 5, session n: generator / tokenizer spin window, copier thread
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
for cfg in "X=1" "EMX_PIPE_GT_SPIN_US=60" "EMX_PIPE_GT_SPIN_US=200" "EMX_PIPE_COPIER=1 EMX_PIPE_GT_SPIN_US=200" "EMX_PIPE_COPIER=1 EMX_PIPE_GT_SPIN_US=60" "EMX_PIPE_COPIER=1 EMX_PIPE_GT_SPIN_US=200 EMX_TUNE=mt_pipeline=4" "EMX_PIPE_COPIER=1"; do
  echo "== $cfg" | tee -a $O/exact_c2_n.txt
  env $cfg timeout 120 python tools/exact_mode_probe.py 2>&1 | tail -n 1 | cut -c1-640 | tee -a $O/exact_c2_n.txt
done
