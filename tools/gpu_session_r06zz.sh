#!/bin/bash
# round 6, closing session (final library, after ROWS_LATE): the GPU suite, smoke, the driver's bench command, its rocprofv3 kernel statistics, the in-place PMC pass
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06zz
O=$PWD/gpurun_out/r06zz
R=$PWD
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/gpu_tests.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail_n1.json 2>/dev/null
wc -c $O/bench_n1.json | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06zz"
d = json.loads(open(O + "/bench_n1.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.5f frac %.4f hpl %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["halfsteps_per_launch"]))
for k, c in d["configs"].items():
    print("  %-28s %.2f us/step frac %.3f" % (k, c["ms_per_step"] * 1e3, c["frac"]))
print("  exact:", json.dumps(d["exact_mode"]))
PY
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc --no-extras --no-cpu-baseline > $O/bench_n1_pmc.json 2> $O/bench_n1_pmc.err; echo "bench pmc rc=$?" | tee -a $O/summary.txt
python -c "
import json; d=json.loads(open('$O/bench_n1_pmc.json').read().strip().splitlines()[-1]); print('pmc traffic per launch', d['roofline'].get('traffic'), 'frac_traffic', d['roofline'].get('frac_traffic'), 'hpl', d['roofline'].get('halfsteps_per_launch'))" | tee -a $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail_n1_pmc.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_c2 -o c2 -f csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/trace_c2.log 2>&1; echo "trace c2 rc=$?" | tee -a $O/summary.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_all -o all -f csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/trace_all.log 2>&1; echo "trace all rc=$?" | tee -a $O/summary.txt
cd $R
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*.db" -delete
du -sh $O
