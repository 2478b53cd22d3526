#!/bin/bash
# Round 3, second GPU session: full parity suite (replay exchange, lean plans), A/B of the lean plans, sharded paths at world 1, PMC passes
set -u
mkdir -p gpurun_out/r03b
O=gpurun_out/r03b
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
for i in 1 2; do
  timeout 120 python tools/ab_cfg.py 20 c2 c3 c4
done > $O/ab_k20.txt 2>&1
cat $O/ab_k20.txt
timeout 300 python bench.py --force-dist --config c2 --steps 20 --warmup 5 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; tail -3 $O/bench_forcedist.err; head -c 3000 $O/bench_forcedist.json; echo
timeout 300 python bench.py --force-dist --config w512 --steps 10 --warmup 2 > $O/bench_forcedist_w512.json 2> $O/bench_forcedist_w512.err; tail -3 $O/bench_forcedist_w512.err; head -c 3000 $O/bench_forcedist_w512.json; echo
bash tools/pmc_hbm.sh $O 2>&1 | tail -12
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof/c2 -o c2 -f csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_c2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof/c3 -o c3 -f csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --config c3 > $O/prof_c3.log 2>&1
find $O/prof -name "*kernel_trace.csv" -size +2M -delete
find $O/prof -name "*_kernel_stats.csv" -exec head -8 {} \;
du -sh $O
