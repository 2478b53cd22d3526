#!/bin/bash
# round 5, session w: where the exact-mode mixture at 4 096 walkers spends its step: kernel statistics of an exact and a Philox run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05w
O=$PWD/gpurun_out/r05w
R=$PWD
export TMPDIR=/tmp
cd /tmp
for rng in mt philox; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$rng -o t -f csv -- python $R/tools/exp/exact_mix_one.py 4096 $rng 2000 > $O/run_$rng.log 2>&1; echo "$rng rc=$?" | tee -a $O/summary.txt
  grep "us/step" $O/run_$rng.log
  head -8 $O/trace_$rng/*kernel_stats.csv | cut -c1-170
done
timeout 120 python $R/tools/exp/exact_mix_one.py 4096 mt 2000 | grep us/step
timeout 120 python $R/tools/exp/exact_mix_one.py 4096 philox 2000 | grep us/step
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
