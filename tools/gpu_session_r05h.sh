#!/bin/bash
# round 5, session h: the pipeline's inner loops alone on this host; exact mode at C2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
for n in 65536 16384; do tools/ubench/bin/mt_scan_bench $n; done > $O/mt_scan_bench.txt 2>&1
cat $O/mt_scan_bench.txt
for cfg in "EMX_TUNE=mt_pipeline=6" "EMX_TUNE=mt_pipeline=4" "EMX_TUNE=mt_pipeline=5"; do
  echo "== $cfg" | tee -a $O/exact_c2_h3.txt
  env $cfg timeout 300 python tools/exact_mode_probe.py 2>&1 | tail -n 1 | cut -c1-600 | tee -a $O/exact_c2_h3.txt
done
