#!/bin/bash
# (ran on the build that had the tuning key plan_side -- commit history: "Plan kernel for batches of lean stretch steps alone ..."; the key is gone: profiles/r05/plan_side_ab.txt)
# round 5, session y: Philox plans of the next batch on a side stream next to the resident persistent launch (tuning plan_side):
# the Philox persistent tests, the A/B at the headline (alternating fresh processes), kernel statistics with it on
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05y
O=$PWD/gpurun_out/r05y
R=$PWD
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_gpu_sampler_api.py -q -p no:cacheprovider -x ) > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/tests.log
for r in 1 2 3; do
  for v in 1 0; do
    echo "plan_side=$v" >> $O/ab.txt
    EMX_TUNE=plan_side=$v timeout 200 python tools/ab_cfg.py 20 c2 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  done
done
cat $O/ab.txt
cd /tmp
EMX_TUNE=plan_side=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_c2 -o c2 -f csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/trace_c2.log 2>&1; echo "trace rc=$?" | tee -a $O/summary.txt
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
head -5 $O/trace_c2/*kernel_stats.csv | cut -c1-170
