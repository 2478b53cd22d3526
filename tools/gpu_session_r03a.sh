#!/bin/bash
# Round 3, first GPU session: parity suite, the driver's bench command, A/B of this round's host/plan changes, resident state,
# rocprof kernel stats, and the self-launched N=2 control flow on the one GPU.
set -u
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; tail -2 $O/bench_k20.err; head -c 1500 $O/bench_k20.json; echo
for i in 1 2 3; do
  timeout 120 python tools/ab_cfg.py 20 c2 c3
  EMX_LIB=$PWD/emcee_amd/libemx_mad0.so timeout 120 python tools/ab_cfg.py 20 c2 c3
  EMX_SPIN_SYNC=0 timeout 120 python tools/ab_cfg.py 20 c2 c3
done > $O/ab_k20.txt 2>&1
cat $O/ab_k20.txt
timeout 120 python tools/ab_cfg.py 400 c2 c3 c4 c5 >> $O/ab_k400.txt 2>&1; cat $O/ab_k400.txt
timeout 200 python tools/api_profile_c2.py > $O/api_profile_c2.txt 2>&1; head -8 $O/api_profile_c2.txt
rm -rf $O/prof
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof/all -o all -f csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_all.log 2>&1
tail -2 $O/prof_all.log
find $O/prof -name "*kernel_trace.csv" -size +2M -delete
find $O/prof -name "*_kernel_stats.csv" -exec head -30 {} \;
timeout 400 python bench.py --gpus 2 --steps 20 --warmup 5 --all-on-device 0 --config c2 > $O/bench_n2_one_device.json 2> $O/bench_n2_one_device.err
tail -5 $O/bench_n2_one_device.err; head -c 2500 $O/bench_n2_one_device.json; echo
du -sh $O
