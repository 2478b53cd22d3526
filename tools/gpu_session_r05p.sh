#!/bin/bash
# round 5, session p: the plan kernel after plan_log / closed-form permutation inverse / narrow bounded draw (emx_planlog.hpp,
# emx_rng.hpp): the GPU suite, k_native_plan_batch's duration at C2 (rocprofv3 kernel statistics of the driver's command), step times
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05p
O=$PWD/gpurun_out/r05p
R=$PWD
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt
tail -n 5 $O/gpu_tests.log
timeout 300 python tools/ab_cfg.py 20 c2 c3 c4 > $O/ab_cfg.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary.txt
cat $O/ab_cfg.txt | grep -v amdgpu.ids
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_c2 -o c2 -f csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/trace_c2.log 2>&1; echo "trace c2 rc=$?" | tee -a $O/summary.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_c3 -o c3 -f csv -- python $R/tools/ab_cfg.py 20 c3 > $O/trace_c3.log 2>&1; echo "trace c3 rc=$?" | tee -a $O/summary.txt
cd $R
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
head -4 $O/trace_c2/*kernel_stats.csv $O/trace_c3/*kernel_stats.csv | cut -c1-160
