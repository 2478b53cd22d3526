#!/bin/bash
# round 5, closing session on one MI355X box: the GPU suite, smoke, the driver's bench command, its rocprofv3 kernel statistics
# (headline alone and every configuration of the line), the in-place PMC traffic pass.  Summaries go to gpurun_out/r05z/ and are
# copied into profiles/r05/ (tracked) by hand afterwards.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05z
O=$PWD/gpurun_out/r05z
R=$PWD
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt
tail -n 5 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc --no-extras --no-cpu-baseline > $O/bench_n1_pmc.json 2> $O/bench_n1_pmc.err; echo "bench pmc rc=$?" | tee -a $O/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_c2 -o c2 -f csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/trace_c2.log 2>&1; echo "trace c2 rc=$?" | tee -a $O/summary.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_c3 -o c3 -f csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --config c3 --no-cpu-baseline > $O/trace_c3.log 2>&1; echo "trace c3 rc=$?" | tee -a $O/summary.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_all -o all -f csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/trace_all.log 2>&1; echo "trace all rc=$?" | tee -a $O/summary.txt
cd $R
find $O -name "*kernel_trace.csv" -size +2M -exec sh -c 'head -300 "$1" > "$1.head"; rm "$1"' _ {} \;
find $O -name "*.db" -delete
find $O -name "*kernel_stats.csv" | head
du -sh $O
