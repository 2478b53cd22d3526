#!/bin/bash
set -u
mkdir -p gpurun_out/r03e
O=gpurun_out/r03e
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -s -k "device_callable or users_hip" > $O/pytest_cb.log 2>&1; tail -8 $O/pytest_cb.log
timeout 300 python tools/device_callable_bench.py > $O/device_callable.txt 2>&1; cat $O/device_callable.txt
# where the plan kernel's time goes: its own average duration under the timing switches (the half-steps are garbage there)
for t in 0 256 512; do
  rm -rf $O/prof_t$t
  EMX_TUNE=ablate=$t timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_t$t -o p -f csv -- python tools/ab_cfg.py 20 c3 > $O/prof_t$t.log 2>&1
  echo "ablate=$t"; find $O/prof_t$t -name "*kernel_stats.csv" -exec grep -h "k_native_plan_batch" {} \;
  find $O/prof_t$t -name "*kernel_trace.csv" -delete
done
for i in 1 2; do timeout 120 python tools/ab_cfg.py 20 c2 c3 c4 c5; done > $O/ab_final.txt 2>&1; cat $O/ab_final.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
du -sh $O
