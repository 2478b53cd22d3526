#!/bin/bash
# One GPU-box session: parity tests, contract bench, rocprofv3 kernel trace + PMC passes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r02}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
timeout 300 python bench.py > gpurun_out/bench_${R}.json 2> gpurun_out/bench_${R}.err
cat gpurun_out/bench_${R}.json; tail -3 gpurun_out/bench_${R}.err
timeout 300 python bench.py --force-dist --config c2 --steps 200 --warmup 20 > gpurun_out/bench_${R}_forcedist.json 2> gpurun_out/bench_${R}_forcedist.err
cat gpurun_out/bench_${R}_forcedist.json; tail -3 gpurun_out/bench_${R}_forcedist.err
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
# kernel trace + stats of the same bench command
rm -rf gpurun_out/prof_${R}
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${R}/trace -o c2 -f csv -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/prof_${R}_trace.log 2>&1
tail -2 gpurun_out/prof_${R}_trace.log
# every configuration of the bench line (C3, C4 DE+snooker, C5, C2 stored, exact mode, quality run) in one kernel-stats table
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${R}/trace_all -o all -f csv -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/prof_${R}_trace_all.log 2>&1
tail -2 gpurun_out/prof_${R}_trace_all.log
# HBM traffic: separate PMC passes (MI355X_MICROARCH.md HBM section)
timeout 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_${R}/pmc_fetch -o c2 -f csv -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/prof_${R}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_${R}/pmc_write -o c2 -f csv -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/prof_${R}_write.log 2>&1
find gpurun_out/prof_${R} -name "*.csv" | head -20
bash tools/pmc_session.sh > gpurun_out/pmc_${R}.txt 2>&1; tail -20 gpurun_out/pmc_${R}.txt
# keep only small summaries
find gpurun_out/prof_${R} -name "*kernel_trace.csv" -size +4M -exec sh -c 'head -400 "$1" > "$1.head"; rm "$1"' _ {} \;
find gpurun_out/prof_${R} -name "*counter_collection.csv" -size +4M -exec sh -c 'head -2000 "$1" > "$1.head"; rm "$1"' _ {} \;
du -sh gpurun_out/prof_${R}
